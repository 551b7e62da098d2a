// Instruction-issue-rate microbenchmark for gfx950 (MI355X).
// Measures wave64 issue cost (cycles per wave-instruction per SIMD) of the integer / fp64
// VALU instructions the 256-bit field arithmetic is built from.  The result fixes the
// roofline denominator ("measure it with a dependency-free v_mad_u64_u32 stream").
//
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

static constexpr int ITERS = 2048;   // loop trips
static constexpr int UNROLL = 16;    // independent instructions per trip (x REP)
static constexpr int REP = 4;

// Each kernel keeps 16 independent accumulators so the stream is dependency-free at depth 16.
#define KERNEL_BEGIN(name) \
  __global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) { \
    uint32_t a[16]; uint32_t b = seed * 2654435761u + threadIdx.x; uint32_t c = seed ^ 0x9e3779b9u; \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = b * (i + 1) + c; \
    for (int it = 0; it < ITERS; ++it) { _Pragma("unroll") for (int r = 0; r < REP; ++r) { _Pragma("unroll") for (int i = 0; i < 16; ++i) {
#define KERNEL_END \
    } } } uint32_t s = 0; _Pragma("unroll") for (int i = 0; i < 16; ++i) s ^= a[i]; \
    if (s == 0x12345678u) out[threadIdx.x] = s; }

KERNEL_BEGIN(k_add_u32)      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); KERNEL_END
KERNEL_BEGIN(k_add_co)       asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc"); KERNEL_END
KERNEL_BEGIN(k_addc_co)      asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"); KERNEL_END
KERNEL_BEGIN(k_add3)         asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); KERNEL_END
KERNEL_BEGIN(k_mul_lo)       asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); KERNEL_END
KERNEL_BEGIN(k_mul_hi)       asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); KERNEL_END
KERNEL_BEGIN(k_mad_u32_u24)  asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); KERNEL_END
KERNEL_BEGIN(k_mul_hi_u24)   asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b)); KERNEL_END
KERNEL_BEGIN(k_cndmask)      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : ); KERNEL_END
KERNEL_BEGIN(k_alignbit)     asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(b)); KERNEL_END
KERNEL_BEGIN(k_xor)          asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); KERNEL_END
KERNEL_BEGIN(k_mov)          asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b)); KERNEL_END
KERNEL_BEGIN(k_fma_f32)      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); KERNEL_END
KERNEL_BEGIN(k_pk_fma_f32_h) if (i < 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(uint64_t*)&a[2*i]) : "v"(*(uint64_t*)&a[0])); KERNEL_END

// 64-bit-register instructions: 8 independent 64-bit accumulators (pairs of the 16 words), two passes
#define KERNEL64_BEGIN(name) \
  __global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) { \
    uint64_t a[16]; uint32_t b = seed * 2654435761u + threadIdx.x; uint32_t c = seed ^ 0x9e3779b9u; \
    uint64_t b64 = ((uint64_t)b << 32) | c; \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = (uint64_t)(b * (i + 1) + c) * 0x100000001ull; \
    for (int it = 0; it < ITERS; ++it) { _Pragma("unroll") for (int r = 0; r < REP; ++r) { _Pragma("unroll") for (int i = 0; i < 16; ++i) {
#define KERNEL64_END \
    } } } uint64_t s = 0; _Pragma("unroll") for (int i = 0; i < 16; ++i) s ^= a[i]; \
    if (s == 0x12345678u) out[threadIdx.x] = (uint32_t)s; }

KERNEL64_BEGIN(k_mad_u64_u32)     asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"); KERNEL64_END
KERNEL64_BEGIN(k_mad_u64_u32_s)   asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "s20", "s21"); KERNEL64_END
KERNEL64_BEGIN(k_mad_i64_i32)     asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"); KERNEL64_END
KERNEL64_BEGIN(k_lshl_add_u64)    asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(b64)); KERNEL64_END
KERNEL64_BEGIN(k_fma_f64)         asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b64)); KERNEL64_END
KERNEL64_BEGIN(k_mul_f64)         asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b64)); KERNEL64_END
KERNEL64_BEGIN(k_add_f64)         asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b64)); KERNEL64_END
KERNEL64_BEGIN(k_lshlrev_b64)     asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a[i]) : ); KERNEL64_END
KERNEL64_BEGIN(k_lshrrev_b64)     asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(a[i]) : ); KERNEL64_END
KERNEL64_BEGIN(k_pk_add_u32pair)  asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(*(uint32_t*)&a[i]) : "v"(b)); KERNEL64_END
// mixed: 1 mad + 1 addc (Comba inner step) -- counts as 2 instructions
KERNEL64_BEGIN(k_comba_step)      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(a[i]), "+v"(c) : "v"(b), "v"(b) : "vcc"); KERNEL64_END


KERNEL_BEGIN(k_cndmask_e64s)  asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b) : ); KERNEL_END
KERNEL_BEGIN(k_cmp_cndmask)   asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"); KERNEL_END
KERNEL_BEGIN(k_cndmask_vccset) if (i == 0 && r == 0) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a[0]), "v"(b) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : ); KERNEL_END
KERNEL_BEGIN(k_bfi)           asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c)); KERNEL_END
KERNEL_BEGIN(k_and_or)        asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); KERNEL_END
KERNEL_BEGIN(k_and)           asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); KERNEL_END
KERNEL_BEGIN(k_sub_co)        asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc"); KERNEL_END
KERNEL_BEGIN(k_subb_co)       asm volatile("v_subb_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"); KERNEL_END
KERNEL_BEGIN(k_addc_e64s)     asm volatile("v_addc_co_u32_e64 %0, s[20:21], %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b) : "s20", "s21"); KERNEL_END
KERNEL_BEGIN(k_lshl_add_u32)  asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b)); KERNEL_END
KERNEL_BEGIN(k_lshlrev_b32)   asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]) : ); KERNEL_END
KERNEL_BEGIN(k_xad)           asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); KERNEL_END
KERNEL_BEGIN(k_sub_u32)       asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); KERNEL_END
KERNEL_BEGIN(k_cmp_only)      asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc"); KERNEL_END
KERNEL_BEGIN(k_pk_add_u16)    asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b)); KERNEL_END
// dependent chains: all 16 slots hit the same accumulator(s)
KERNEL64_BEGIN(k_mad_dep1)    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[0]) : "v"(b), "v"(c) : "vcc"); KERNEL64_END
KERNEL64_BEGIN(k_mad_dep2)    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i&1]) : "v"(b), "v"(c) : "vcc"); KERNEL64_END
KERNEL64_BEGIN(k_mad_dep4)    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i&3]) : "v"(b), "v"(c) : "vcc"); KERNEL64_END
KERNEL64_BEGIN(k_comba_dep1)  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(a[0]), "+v"(c) : "v"(b), "v"(b) : "vcc"); KERNEL64_END
KERNEL_BEGIN(k_addc_dep1)     asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[0]) : "v"(b) : "vcc"); KERNEL_END
KERNEL_BEGIN(k_addc_chain8)   asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i&7]) : "v"(b) : "vcc"); KERNEL_END
KERNEL_BEGIN(k_add_dep1)      asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(b)); KERNEL_END
KERNEL64_BEGIN(k_fma64_dep1)  asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[0]) : "v"(b64)); KERNEL64_END

// register-move forms for the Comba column shift, and the cost of the s_nop the compiler puts between
// dependent inline-asm statements (gfx950 dst-forwarding hazard workaround)
KERNEL64_BEGIN(k_pk_mov)      asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]" : "+v"(a[i]) : "v"(b64)); KERNEL64_END
KERNEL64_BEGIN(k_mov_b64)     asm volatile("v_mov_b64 %0, %1" : "+v"(a[i]) : "v"(b64)); KERNEL64_END
KERNEL64_BEGIN(k_comba4_sep)  if (i < 4) { uint32_t ex = c;
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(a[i]), "+v"(ex) : "v"(b), "v"(c) : "vcc");
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(a[i]), "+v"(ex) : "v"(b), "v"(c) : "vcc");
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(a[i]), "+v"(ex) : "v"(b), "v"(c) : "vcc");
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(a[i]), "+v"(ex) : "v"(b), "v"(c) : "vcc");
    a[i] ^= ex; } KERNEL64_END
KERNEL64_BEGIN(k_comba4_one)  if (i < 4) { uint32_t ex = c;
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
                 "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
                 "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
                 "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(a[i]), "+v"(ex) : "v"(b), "v"(c) : "vcc");
    a[i] ^= ex; } KERNEL64_END


// ---- round 2: forms for the conditional corrections of the modular linear ops.  Each statement below is a GROUP of 8
// VALU instructions (plus scalar set-up) executed for i < 2, so a trip still counts 16 VALU instructions per REP.
#define G8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define OPS8 "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
#define SRC8 "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15])
// 8 selects by an SGPR-pair mask (what cond_sub_p does today)
KERNEL_BEGIN(k_g_cndmask_e64) if (i < 2) asm volatile("s_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x33333333\n\t"
    "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\tv_cndmask_b32_e64 %1, %1, %9, s[20:21]\n\tv_cndmask_b32_e64 %2, %2, %10, s[20:21]\n\tv_cndmask_b32_e64 %3, %3, %11, s[20:21]\n\t"
    "v_cndmask_b32_e64 %4, %4, %12, s[20:21]\n\tv_cndmask_b32_e64 %5, %5, %13, s[20:21]\n\tv_cndmask_b32_e64 %6, %6, %14, s[20:21]\n\tv_cndmask_b32_e64 %7, %7, %15, s[20:21]"
    : OPS8 : SRC8 : "s20", "s21"); KERNEL_END
// the mask in VCC written by a scalar instruction, 8 VOP2 selects
KERNEL_BEGIN(k_g_cndmask_vcc_salu) if (i < 2) asm volatile("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x33333333\n\t"
    "v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %9, vcc\n\tv_cndmask_b32 %2, %2, %10, vcc\n\tv_cndmask_b32 %3, %3, %11, vcc\n\t"
    "v_cndmask_b32 %4, %4, %12, vcc\n\tv_cndmask_b32 %5, %5, %13, vcc\n\tv_cndmask_b32 %6, %6, %14, vcc\n\tv_cndmask_b32 %7, %7, %15, vcc"
    : OPS8 : SRC8 : "vcc"); KERNEL_END
// the mask in VCC written by ONE vector compare (a 9th VALU instruction, not counted), 8 VOP2 selects
KERNEL_BEGIN(k_g_cndmask_vcc_valu) if (i < 2) asm volatile("v_cmp_lt_u32 vcc, %0, %8\n\t"
    "v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %9, vcc\n\tv_cndmask_b32 %2, %2, %10, vcc\n\tv_cndmask_b32 %3, %3, %11, vcc\n\t"
    "v_cndmask_b32 %4, %4, %12, vcc\n\tv_cndmask_b32 %5, %5, %13, vcc\n\tv_cndmask_b32 %6, %6, %14, vcc\n\tv_cndmask_b32 %7, %7, %15, vcc"
    : OPS8 : SRC8 : "vcc"); KERNEL_END
// vector compare, scalar and-not into VCC, 8 VOP2 selects (cond_sub_p with the mask kept in VCC)
KERNEL_BEGIN(k_g_cndmask_vcc_andn2) if (i < 2) asm volatile("s_mov_b32 s20, 0x0f0f0f0f\n\ts_mov_b32 s21, 0x00ff00ff\n\tv_cmp_lt_u32 vcc, %0, %8\n\ts_andn2_b64 vcc, vcc, s[20:21]\n\t"
    "v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %9, vcc\n\tv_cndmask_b32 %2, %2, %10, vcc\n\tv_cndmask_b32 %3, %3, %11, vcc\n\t"
    "v_cndmask_b32 %4, %4, %12, vcc\n\tv_cndmask_b32 %5, %5, %13, vcc\n\tv_cndmask_b32 %6, %6, %14, vcc\n\tv_cndmask_b32 %7, %7, %15, vcc"
    : OPS8 : SRC8 : "vcc", "scc", "s20", "s21"); KERNEL_END
// EXEC-masked moves: s_and_saveexec, 8 v_mov, restore
KERNEL_BEGIN(k_g_mov_exec) if (i < 2) asm volatile("s_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x33333333\n\ts_and_saveexec_b64 s[22:23], s[20:21]\n\t"
    "v_mov_b32 %0, %8\n\tv_mov_b32 %1, %9\n\tv_mov_b32 %2, %10\n\tv_mov_b32 %3, %11\n\tv_mov_b32 %4, %12\n\tv_mov_b32 %5, %13\n\tv_mov_b32 %6, %14\n\tv_mov_b32 %7, %15\n\t"
    "s_mov_b64 exec, s[22:23]"
    : OPS8 : SRC8 : "scc", "s20", "s21", "s22", "s23"); KERNEL_END
// EXEC-masked in-place borrow chain with inline constants (subtract p under the mask)
KERNEL_BEGIN(k_g_subb_exec) if (i < 2) asm volatile("s_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x33333333\n\ts_and_saveexec_b64 s[22:23], s[20:21]\n\t"
    "v_sub_co_u32 %0, vcc, %0, -1\n\tv_subb_co_u32 %1, vcc, %1, -1, vcc\n\tv_subb_co_u32 %2, vcc, %2, -1, vcc\n\tv_subb_co_u32 %3, vcc, %3, 0, vcc\n\t"
    "v_subb_co_u32 %4, vcc, %4, 0, vcc\n\tv_subb_co_u32 %5, vcc, %5, 0, vcc\n\tv_subb_co_u32 %6, vcc, %6, 1, vcc\n\tv_subb_co_u32 %7, vcc, %7, -1, vcc\n\t"
    "s_mov_b64 exec, s[22:23]"
    : OPS8 : : "vcc", "scc", "s20", "s21", "s22", "s23"); KERNEL_END
// the same chain without the EXEC games (reference for the previous line)
KERNEL_BEGIN(k_g_subb_plain) if (i < 2) asm volatile(
    "v_sub_co_u32 %0, vcc, %0, -1\n\tv_subb_co_u32 %1, vcc, %1, -1, vcc\n\tv_subb_co_u32 %2, vcc, %2, -1, vcc\n\tv_subb_co_u32 %3, vcc, %3, 0, vcc\n\t"
    "v_subb_co_u32 %4, vcc, %4, 0, vcc\n\tv_subb_co_u32 %5, vcc, %5, 0, vcc\n\tv_subb_co_u32 %6, vcc, %6, 1, vcc\n\tv_subb_co_u32 %7, vcc, %7, -1, vcc"
    : OPS8 : : "vcc"); KERNEL_END
// v_swap_b32 (a conditional swap under EXEC would be 8 of these per field element)
KERNEL_BEGIN(k_g_swap) if (i < 2) asm volatile(
    "v_swap_b32 %0, %8\n\tv_swap_b32 %1, %9\n\tv_swap_b32 %2, %10\n\tv_swap_b32 %3, %11\n\tv_swap_b32 %4, %12\n\tv_swap_b32 %5, %13\n\tv_swap_b32 %6, %14\n\tv_swap_b32 %7, %15"
    : OPS8, "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])); KERNEL_END
// carry-less add under EXEC
KERNEL_BEGIN(k_g_add_exec) if (i < 2) asm volatile("s_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x33333333\n\ts_and_saveexec_b64 s[22:23], s[20:21]\n\t"
    "v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %9\n\tv_add_u32 %2, %2, %10\n\tv_add_u32 %3, %3, %11\n\tv_add_u32 %4, %4, %12\n\tv_add_u32 %5, %5, %13\n\tv_add_u32 %6, %6, %14\n\tv_add_u32 %7, %7, %15\n\t"
    "s_mov_b64 exec, s[22:23]"
    : OPS8 : SRC8 : "scc", "s20", "s21", "s22", "s23"); KERNEL_END
// v_cmp_eq with an SGPR-pair destination + scalar test + never-taken branch (the rare-case guard)
KERNEL_BEGIN(k_g_guard) if (i < 2) asm volatile(
    "v_sub_co_u32 %0, vcc, %0, -1\n\tv_subb_co_u32 %1, vcc, %1, -1, vcc\n\tv_subb_co_u32 %2, vcc, %2, -1, vcc\n\tv_subb_co_u32 %3, vcc, %3, 0, vcc\n\t"
    "v_subb_co_u32 %4, vcc, %4, 0, vcc\n\tv_subb_co_u32 %5, vcc, %5, 0, vcc\n\tv_subb_co_u32 %6, vcc, %6, 1, vcc\n\t"
    "v_cmp_eq_u32_e64 s[20:21], %7, -1\n\ts_cmp_lg_u64 s[20:21], 0\n\ts_cbranch_scc0 1f\n\tv_add_u32 %7, %7, 1\n\t1:"
    : OPS8 : : "vcc", "scc", "s20", "s21"); KERNEL_END


// ---- round 4: the instructions a reduced-radix (9 x 29-bit signed limbs) field arithmetic is built from
KERNEL_BEGIN(k4_ashrrev_i32)  asm volatile("v_ashrrev_i32 %0, 29, %0" : "+v"(a[i]) : ); KERNEL_END
KERNEL_BEGIN(k4_lshrrev_b32)  asm volatile("v_lshrrev_b32 %0, 29, %0" : "+v"(a[i]) : ); KERNEL_END
KERNEL_BEGIN(k4_bfe_i32)      asm volatile("v_bfe_i32 %0, %0, 0, 29" : "+v"(a[i]) : ); KERNEL_END
KERNEL_BEGIN(k4_and_lit)      asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(a[i]) : ); KERNEL_END
KERNEL_BEGIN(k4_lshl_or)      asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b)); KERNEL_END
KERNEL_BEGIN(k4_add_lshl)     asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(a[i]) : "v"(b)); KERNEL_END
KERNEL_BEGIN(k4_sub_lit)      asm volatile("v_subrev_u32 %0, 0x10000000, %0" : "+v"(a[i]) : ); KERNEL_END
KERNEL64_BEGIN(k4_ashrrev_i64)   asm volatile("v_ashrrev_i64 %0, 29, %0" : "+v"(a[i]) : ); KERNEL64_END
KERNEL64_BEGIN(k4_mad_i64_sconst) asm volatile("s_mov_b32 s20, 0x200\n\tv_mad_i64_i32 %0, s[22:23], %1, s20, %0" : "+v"(a[i]) : "v"(b) : "s20", "s22", "s23"); KERNEL64_END
// a normalisation pass element: c = x >> 29 (arithmetic), y = (x & m) + c'
KERNEL_BEGIN(k4_norm_limb)    if (i < 5) { uint32_t t;
    asm volatile("v_ashrrev_i32 %1, 29, %0\n\tv_and_b32 %0, 0x1fffffff, %0\n\tv_add_u32 %0, %0, %2" : "+v"(a[i]), "=&v"(t) : "v"(b)); a[i + 8] ^= t; } KERNEL_END
// alternating full-rate and dual-rate instructions: do they cost the sum?
KERNEL64_BEGIN(k4_mad_and_mix) if (i < 8) { asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_and_b32 %3, 0x1fffffff, %3" : "+v"(a[i]), "+v"(*(uint32_t*)&a[i + 8]) : "v"(b), "v"(c) : "vcc"); } KERNEL64_END

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; int insts_per_slot; };

int main(int argc, char** argv) {
  int dev = 0; CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, dev));
  int cus = prop.multiProcessorCount;
  printf("device: %s  CUs=%d  clockRate=%d kHz  arch=%s\n", prop.name, cus, prop.clockRate, prop.gcnArchName);
  uint32_t* out; CHECK(hipMalloc(&out, 4096));
  std::vector<Entry> es = {
    {"v_add_u32", k_add_u32, 1}, {"v_add_co_u32", k_add_co, 1}, {"v_addc_co_u32", k_addc_co, 1},
    {"v_add3_u32", k_add3, 1}, {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mul_hi_u32", k_mul_hi, 1},
    {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u24, 1},
    {"v_cndmask_b32", k_cndmask, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_xor_b32", k_xor, 1}, {"v_mov_b32", k_mov, 1},
    {"v_fma_f32", k_fma_f32, 1},
    {"v_mad_u64_u32(vcc)", k_mad_u64_u32, 1}, {"v_mad_u64_u32(sgpr)", k_mad_u64_u32_s, 1}, {"v_mad_i64_i32", k_mad_i64_i32, 1},
    {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_fma_f64", k_fma_f64, 1}, {"v_mul_f64", k_mul_f64, 1}, {"v_add_f64", k_add_f64, 1},
    {"v_lshlrev_b64", k_lshlrev_b64, 1}, {"v_lshrrev_b64", k_lshrrev_b64, 1},
    {"mad_u64_u32+addc", k_comba_step, 2},
    {"v_cndmask(e64,sgpr)", k_cndmask_e64s, 1}, {"v_cmp+v_cndmask", k_cmp_cndmask, 2}, {"v_cndmask(vcc set once)", k_cndmask_vccset, 1},
    {"v_bfi_b32", k_bfi, 1}, {"v_and_or_b32", k_and_or, 1}, {"v_and_b32", k_and, 1}, {"v_sub_co_u32", k_sub_co, 1}, {"v_subb_co_u32", k_subb_co, 1},
    {"v_addc_co(e64,sgpr)", k_addc_e64s, 1}, {"v_lshl_add_u32", k_lshl_add_u32, 1}, {"v_lshlrev_b32", k_lshlrev_b32, 1}, {"v_xad_u32", k_xad, 1},
    {"v_sub_u32", k_sub_u32, 1}, {"v_cmp_lt_u32", k_cmp_only, 1}, {"v_pk_add_u16", k_pk_add_u16, 1},
    {"mad_u64 dep-chain x1", k_mad_dep1, 1}, {"mad_u64 dep-chain x2", k_mad_dep2, 1}, {"mad_u64 dep-chain x4", k_mad_dep4, 1},
    {"mad+addc dep x1", k_comba_dep1, 2}, {"addc dep x1", k_addc_dep1, 1}, {"addc chain(8 regs)", k_addc_chain8, 1}, {"v_add_u32 dep x1", k_add_dep1, 1}, {"v_fma_f64 dep x1", k_fma64_dep1, 1},
    {"v_pk_mov_b32", k_pk_mov, 1}, {"v_mov_b64", k_mov_b64, 1},
    {"4x(mad+addc) 4 asm /4", k_comba4_sep, 2}, {"4x(mad+addc) 1 asm /4", k_comba4_one, 2},
    {"8 cndmask e64 sgpr", k_g_cndmask_e64, 1}, {"8 cndmask vcc<-salu", k_g_cndmask_vcc_salu, 1}, {"cmp + 8 cndmask vcc", k_g_cndmask_vcc_valu, 1},
    {"cmp,andn2,8 cndmask", k_g_cndmask_vcc_andn2, 1}, {"8 v_mov under exec", k_g_mov_exec, 1}, {"8 subb under exec", k_g_subb_exec, 1},
    {"8 subb plain", k_g_subb_plain, 1}, {"8 v_swap_b32", k_g_swap, 1}, {"8 v_add_u32 exec", k_g_add_exec, 1}, {"7 subb+cmp+guard", k_g_guard, 1},
    // round 4 (names start with "r4 ": `valu_rates r4` runs only these).  Slots that execute for i < K only are scaled below.
    {"r4 v_ashrrev_i32", k4_ashrrev_i32, 1}, {"r4 v_lshrrev_b32", k4_lshrrev_b32, 1}, {"r4 v_bfe_i32", k4_bfe_i32, 1}, {"r4 v_and_b32 literal", k4_and_lit, 1},
    {"r4 v_lshl_or_b32", k4_lshl_or, 1}, {"r4 v_add_lshl_u32", k4_add_lshl, 1}, {"r4 v_subrev_u32 literal", k4_sub_lit, 1}, {"r4 v_ashrrev_i64", k4_ashrrev_i64, 1},
    {"r4 v_mad_i64_i32 sgpr", k4_mad_i64_sconst, 1},
    {"r4 norm limb ashr+and+add /3", k4_norm_limb, 1},
    {"r4 mad_i64 + v_and pairs /2", k4_mad_and_mix, 1},
  };
  const char* filter = argc > 1 ? argv[1] : nullptr;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int wps_list[] = {1, 2, 4, 8};   // waves per SIMD
  printf("%-22s", "instruction");
  for (int w : wps_list) printf("  w/SIMD=%d: Tinst/s  cyc@2.4GHz", w);
  printf("\n");
  for (auto& e : es) {
    if (filter && !strstr(e.name, filter)) continue;
    printf("%-30s", e.name);
    for (int wps : wps_list) {
      // blocks of 256 threads = 4 waves = one wave per SIMD of a CU; wps blocks per CU
      int grid = cus * wps;
      hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, out, 1u);   // warm-up
      CHECK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, out, (uint32_t)rep + 2);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      double wave_insts = (double)grid * 4 /*waves*/ * ITERS * REP * UNROLL * e.insts_per_slot;
      double lane_insts_per_s = wave_insts * 64 / (best * 1e-3);
      // cycles per wave-instruction per SIMD at 2.4 GHz: each SIMD runs wps waves
      double insts_per_simd = (double)wps * ITERS * REP * UNROLL * e.insts_per_slot;
      double cyc = (best * 1e-3) * 2.4e9 / insts_per_simd;
      printf("  %9.2f T/s %7.2f cyc    ", lane_insts_per_s * 1e-12, cyc);
    }
    printf("\n");
  }
  return 0;
}
