// hbm_gather_calib.hip -- calibrates rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ* on gfx950 for the access patterns of this
// library.  /opt/skills/guides/MI355X_MICROARCH.md establishes "FETCH_SIZE = half the bytes" only for
// wide coalesced streaming reads; the window-table kernels (k_base_windowed_g, k_varwin_mult_*) read random 64-byte
// entries, 4 x global_load_dwordx4 per lane.  Every kernel here requests a KNOWN number of bytes, each byte exactly once:
//
//   calib_stream16      one 16-byte load per lane, coalesced (the guide's own case: the control)
//   calib_stream32      two 16-byte loads per lane from a 32-byte element (fe_load: the ladder's scalars and points)
//   calib_gather64      each lane reads ONE random 64-byte-aligned 64-byte entry (4 x 16 B) of a 1 GiB table; the slots are a
//                       permutation of the lanes (slot = lane * odd constant mod 2^24), so every entry is read exactly once and
//                       the two halves of a 128-byte line are read by lanes far apart in time
//   calib_gather64_lo   the same over a 2 GiB table using only the FIRST half of every 128-byte line
//   calib_gather128     each lane reads one random 128-byte line (8 x 16 B) of a 2 GiB table
//   calib_gather64_ic   calib_gather64 over a 32 MiB table, 32 rounds: the table stays in the Infinity Cache / L2
//   calib_own512        lane i reads ONE of the eight 64-byte entries of ITS OWN 512-byte block (the per-element window
//                       tables of k_varwin_mult_*: neighbouring lanes read from neighbouring blocks), 2^22 lanes over 2 GiB
//   calib_own512_reread (r4) lane i reads ALL EIGHT entries of its own 512-byte block, 63 times over (the constant-time window loop of
//                       k_varwin_mult_odd<true>: one table per lane, re-read in every window), at 3 workgroups per CU like that
//                       kernel (dynamic LDS caps the occupancy): the blocks of the resident lanes -- 256 CUs x 768 lanes x 512 B =
//                       100 MB -- exceed the L2s (4 MiB per XCD) and fit the 256 MB Infinity Cache.  If this pattern sustains more
//                       than HBM's 8 TB/s, the re-reads are served by the Infinity Cache -- and the counters count them all the same.
//
// Run plainly it prints requested bytes and GB/s per kernel (HIP events); under `rocprofv3 --pmc FETCH_SIZE` (and, in
// separate passes, WRITE_SIZE, TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum) the counters per kernel divided by the
// requested bytes are the calibration factors tools/summarize_profiles.py applies per access pattern.
//
// build: hipcc --offload-arch=gfx950 -O3 -o hbm_gather_calib hbm_gather_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

__global__ void __launch_bounds__(256) calib_stream16(const uint4* __restrict__ p, uint32_t* sink, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = fold(p[i]);
  if (s == 0x12345678u) sink[0] = s;
}
__global__ void __launch_bounds__(256) calib_stream32(const uint4* __restrict__ p, uint32_t* sink, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = fold(p[2 * i]) ^ fold(p[2 * i + 1]);
  if (s == 0x12345678u) sink[0] = s;
}
// slot mask m + 1 = number of slots (a power of two); st = 16-byte units between slots (4: every 64 B, 8: every 128 B);
// slot = lane * odd constant + salt mod 2^k: a permutation of the slots.  Distinct symbols so that rocprofv3 reports them apart.
__global__ void __launch_bounds__(256) calib_gather64(const uint4* t, uint32_t* s, size_t n, uint32_t m, uint32_t st, uint32_t salt) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  const uint32_t slot = ((uint32_t)i * 2654435761u + salt) & m; const uint4* e = t + (size_t)slot * st;
  const uint32_t v = fold(e[0]) ^ fold(e[1]) ^ fold(e[2]) ^ fold(e[3]); if (v == 0x12345678u) s[0] = v; }
__global__ void __launch_bounds__(256) calib_gather64_lo(const uint4* t, uint32_t* s, size_t n, uint32_t m, uint32_t st, uint32_t salt) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  const uint32_t slot = ((uint32_t)i * 2654435761u + salt) & m; const uint4* e = t + (size_t)slot * st;
  const uint32_t v = fold(e[0]) ^ fold(e[1]) ^ fold(e[2]) ^ fold(e[3]); if (v == 0x12345678u) s[0] = v; }
__global__ void __launch_bounds__(256) calib_gather128(const uint4* t, uint32_t* s, size_t n, uint32_t m, uint32_t st, uint32_t salt) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  const uint32_t slot = ((uint32_t)i * 2654435761u + salt) & m; const uint4* e = t + (size_t)slot * st;
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) v ^= fold(e[k]);
  if (v == 0x12345678u) s[0] = v; }
__global__ void __launch_bounds__(256) calib_gather64_ic(const uint4* t, uint32_t* s, size_t n, uint32_t m, uint32_t st, uint32_t salt) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  const uint32_t slot = ((uint32_t)i * 2654435761u + salt) & m; const uint4* e = t + (size_t)slot * st;
  const uint32_t v = fold(e[0]) ^ fold(e[1]) ^ fold(e[2]) ^ fold(e[3]); if (v == 0x12345678u) s[0] = v; }

__global__ void __launch_bounds__(256) calib_own512(const uint4* t, uint32_t* s, size_t n, uint32_t salt) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  const uint32_t d = (((uint32_t)i * 2654435761u + salt) >> 13) & 7u;                 // a pseudo-random digit per lane
  const uint4* e = t + i * 32 + d * 4;
  const uint32_t v = fold(e[0]) ^ fold(e[1]) ^ fold(e[2]) ^ fold(e[3]); if (v == 0x12345678u) s[0] = v; }

// 63 rounds x 8 entries x 64 B per lane; asm volatile loads: the compiler must not keep the block in registers
__global__ void __launch_bounds__(256) calib_own512_reread(const uint4* t, uint32_t* s, size_t n, int rounds) {
  extern __shared__ uint32_t occupancy_cap[];
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  const uint4* e = t + i * 32;
  uint32_t v = 0;
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
      uint4 a0, a1, a2, a3, a4, a5, a6, a7;
      asm volatile("global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:16\n\tglobal_load_dwordx4 %2, %8, off offset:32\n\t"
                   "global_load_dwordx4 %3, %8, off offset:48\n\tglobal_load_dwordx4 %4, %8, off offset:64\n\tglobal_load_dwordx4 %5, %8, off offset:80\n\t"
                   "global_load_dwordx4 %6, %8, off offset:96\n\tglobal_load_dwordx4 %7, %8, off offset:112\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(e + k) : "memory");
      v ^= fold(a0) ^ fold(a1) ^ fold(a2) ^ fold(a3) ^ fold(a4) ^ fold(a5) ^ fold(a6) ^ fold(a7);
    }
  }
  if (v == 0x12345678u) { s[0] = v; occupancy_cap[threadIdx.x] = v; }
}

int main() {
  const size_t n = (size_t)1 << 24;                 // lanes per launch
  const size_t bytes = (size_t)2 << 30;             // 2 GiB: 8x the Infinity Cache
  uint4* buf = nullptr; uint32_t* sink = nullptr;
  CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&sink, 256));
  CHECK(hipMemset(buf, 0x5a, bytes)); CHECK(hipMemset(sink, 0, 256));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const dim3 grid((unsigned)(n / 256)), block(256);
  auto report = [&](const char* name, double requested, int launches) {
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"kernel\": \"%s\", \"launches\": %d, \"requested_bytes_per_launch\": %.0f, \"ms_per_launch\": %.4f, \"requested_GBps\": %.1f}\n",
           name, launches, requested, ms / launches, requested / (ms / launches * 1e-3) / 1e9);
  };
  const int reps = 3;
  // every timed group is preceded by a pass over the OTHER half of the buffer so that nothing useful is left in the caches
  auto flush = [&]() { hipLaunchKernelGGL(calib_stream16, dim3((unsigned)((bytes / 16) / 256)), block, 0, 0, buf, sink, bytes / 16); };
  flush(); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(calib_stream16, dim3((unsigned)((bytes / 16) / 256)), block, 0, 0, buf, sink, bytes / 16);   // 2 GiB per launch
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); report("calib_stream16", (double)bytes, reps);
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(calib_stream32, dim3((unsigned)((bytes / 32) / 256)), block, 0, 0, buf, sink, bytes / 32);   // 2 GiB per launch
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); report("calib_stream32", (double)bytes, reps);
  // gather64: 2^24 lanes x 64 B = the first 1 GiB, every slot once
  flush(); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(calib_gather64, grid, block, 0, 0, buf, sink, n, (uint32_t)(n - 1), 4u, 0u);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); report("calib_gather64", (double)n * 64, 1);
  flush(); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(calib_gather64_lo, grid, block, 0, 0, buf, sink, n, (uint32_t)(n - 1), 8u, 0u);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); report("calib_gather64_lo", (double)n * 64, 1);
  flush(); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(calib_gather128, grid, block, 0, 0, buf, sink, n, (uint32_t)(n - 1), 8u, 0u);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); report("calib_gather128", (double)n * 128, 1);
  // Infinity-Cache resident: 32 MiB table = 2^19 slots, 2^24 lanes = 32 reads of every entry per launch
  flush(); CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL(calib_gather64_ic, grid, block, 0, 0, buf, sink, n, (uint32_t)((1u << 19) - 1), 4u, 0u);      // warms the caches
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(calib_gather64_ic, grid, block, 0, 0, buf, sink, n, (uint32_t)((1u << 19) - 1), 4u, 7u);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); report("calib_gather64_ic", (double)n * 64, 1);
  flush(); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(calib_own512, dim3((unsigned)((n / 4) / 256)), block, 0, 0, buf, sink, n / 4, 3u);
  CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); report("calib_own512", (double)(n / 4) * 64, 1);
  // the constant-time window loop's pattern: 2^22 lanes, 63 rounds over each lane's own 512-byte block, 3 workgroups per CU
  {
    const size_t lanes = n / 4; const int rounds = 63;
    const size_t lds = 52 * 1024;                                   // 160 KB per CU / 52 KB = 3 workgroups of 4 waves
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(calib_own512_reread), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    flush(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(calib_own512_reread, dim3((unsigned)(lanes / 256)), block, lds, 0, buf, sink, lanes, rounds);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); report("calib_own512_reread", (double)lanes * 512 * rounds, 1);
  }
  CHECK(hipFree(buf)); CHECK(hipFree(sink));
  return 0;
}
