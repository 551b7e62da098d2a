// Standalone check of sqr8 (cross sum / doubling / diagonal chains) and of ecsimd_hip_square against host
// big-int arithmetic on carry-heavy operands.  This is the program that showed the HIP squaring to be exact
// and the REFERENCE's square() to drop a carry.  Build: see the first lines of main().
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include "../../ecsimd_amd/csrc/field.cuh"
#include "../../include/ecsimd_hip.h"
using namespace ecsimd_hip;

template <int STAGE> __global__ void k(const uint64_t* a, uint64_t* out, size_t n) {
  size_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  fe x = fe_load(a, i);
  fe2 t;
  if constexpr (STAGE == 3) { t = sqr8(x); }
  else {
    uint32_t c[16]; c[0] = 0; c[15] = 0; uint64_t acc = 0; uint32_t ex = 0;
#pragma unroll
    for (int kk = 1; kk < 14; ++kk) {
      bool first = true;
#pragma unroll
      for (int ii = 0; ii < 8; ++ii) {
        const int j = kk - ii;
        if (j <= ii || j > 7) continue;
        if (STAGE == 0) { if (first) { if (kk <= 2) { if (kk == 1) acc = mul_wide(x.w[ii], x.w[j]); else mac_nocarry(acc, x.w[ii], x.w[j]); } else mac_first(acc, ex, x.w[ii], x.w[j]); } else mac(acc, ex, x.w[ii], x.w[j]); }
        else { // STAGE 1/2: always generic mac with explicit zeroing
          if (first) ex = 0;
          mac(acc, ex, x.w[ii], x.w[j]);
        }
        first = false;
      }
      c[kk] = (uint32_t)acc;
      if (STAGE == 0 && (kk == 1 || kk == 2)) { acc >>= 32; ex = 0; }
      else acc = (acc >> 32) | ((uint64_t)ex << 32);
    }
    c[14] = (uint32_t)acc;
    if (STAGE == 2) {
      c[15] = c[14] >> 31;
#pragma unroll
      for (int ii = 14; ii > 0; --ii) c[ii] = __builtin_amdgcn_alignbit(c[ii], c[ii - 1], 31);
    }
#pragma unroll
    for (int ii = 0; ii < 16; ++ii) t.w[ii] = c[ii];
  }
  fe2_store(out, i, t);
}

typedef unsigned __int128 u128;
static void host_ref(const uint64_t* a, int stage, uint32_t* out) {  // words
  uint32_t w[8]; for (int i = 0; i < 4; ++i) { w[2*i] = (uint32_t)a[i]; w[2*i+1] = (uint32_t)(a[i] >> 32); }
  // accumulate into 17 x 64-bit columns then carry propagate
  uint64_t col_lo[18] = {0}, col_hi[18] = {0};
  auto add = [&](int pos, u128 v) { // add 128-bit v at word position pos
    for (int k = 0; k < 4 && pos + k < 18; ++k) { u128 s = (u128)col_lo[pos + k] + (uint32_t)(v >> (32 * k)); col_lo[pos + k] = (uint64_t)s; }
  };
  int mult = (stage == 0 || stage == 1) ? 1 : 2;
  for (int i = 0; i < 8; ++i) for (int j = i + 1; j < 8; ++j) { u128 p = (u128)w[i] * w[j] * mult; add(i + j, p); }
  if (stage == 3) for (int i = 0; i < 8; ++i) add(2 * i, (u128)w[i] * w[i]);
  uint64_t carry = 0;
  for (int k = 0; k < 16; ++k) { u128 s = (u128)col_lo[k] + carry; out[k] = (uint32_t)s; carry = (uint64_t)(s >> 32); }
}

int main() {
  const size_t n = (size_t)1 << 20;
  std::mt19937_64 rng(1);
  const uint32_t pat[6] = {0, 0xffffffffu, 0x80000000u, 0x7fffffffu, 1, 0xfffffffeu};
  std::vector<uint64_t> a(4 * n);
  for (size_t i = 0; i < n; ++i) for (int l = 0; l < 4; ++l) {
    uint32_t lo = (rng() % 4 == 0) ? (uint32_t)rng() : pat[rng() % 6], hi = (rng() % 4 == 0) ? (uint32_t)rng() : pat[rng() % 6];
    a[4 * i + l] = lo | ((uint64_t)hi << 32);
  }
  uint64_t *da, *dout; hipMalloc(&da, a.size() * 8); hipMalloc(&dout, 8 * n * 8);
  hipMemcpy(da, a.data(), a.size() * 8, hipMemcpyHostToDevice);
  std::vector<uint64_t> out(8 * n);
  ecsimd_hip_ctx* ctx = nullptr; int rc = ecsimd_hip_init(0, &ctx); printf("init rc=%d\n", rc);
  for (int stage = 0; stage < 5; ++stage) {
    if (stage == 0) hipLaunchKernelGGL(k<0>, dim3(n / 256), dim3(256), 0, 0, da, dout, n);
    if (stage == 1) hipLaunchKernelGGL(k<1>, dim3(n / 256), dim3(256), 0, 0, da, dout, n);
    if (stage == 2) hipLaunchKernelGGL(k<2>, dim3(n / 256), dim3(256), 0, 0, da, dout, n);
    if (stage == 3) hipLaunchKernelGGL(k<3>, dim3(n / 256), dim3(256), 0, 0, da, dout, n);
    if (stage == 4) { rc = ecsimd_hip_square(ctx, da, dout, n); ecsimd_hip_sync(ctx); printf("square rc=%d\n", rc); }
    hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost);
    size_t bad = 0; int shown = 0;
    for (size_t i = 0; i < n; ++i) {
      uint32_t ref[16]; host_ref(&a[4 * i], stage == 4 ? 3 : stage, ref);
      bool ok = true;
      for (int k = 0; k < 16; ++k) { uint32_t g = (uint32_t)(out[8 * i + k / 2] >> (32 * (k & 1))); if (g != ref[k]) ok = false; }
      if (!ok) { ++bad; if (shown++ < 3) { printf("  stage %d row %zu a=", stage, i); for (int l = 3; l >= 0; --l) printf("%016llx", (unsigned long long)a[4*i+l]); printf("\n   got="); for (int k = 15; k >= 0; --k) printf("%08x ", (uint32_t)(out[8*i+k/2] >> (32*(k&1)))); printf("\n   exp="); for (int k = 15; k >= 0; --k) printf("%08x ", ref[k]); printf("\n"); } }
    }
    printf("stage %d (%s): %zu bad of %zu\n", stage, stage == 0 ? "cross, product code" : stage == 1 ? "cross, generic mac" : stage == 2 ? "cross generic + doubling" : stage == 3 ? "full sqr8" : "PRODUCT ecsimd_hip_square", bad, n);
  }
  return 0;
}
