// tm_platform.h -- everything the kernel sources ask of the machine, in ONE place.
//
// The product is gfx950 only: the second half of this file (the #else branch) is what ships -- DPP moves, ballots, s_setprio /
// s_sleep, agent-scope atomics, the reciprocal-based division sequences, SGPR / VGPR pins.  The first half is the same vocabulary
// for tests/emul (TM_EMULATE): the `-m "not gpu"` tier compiles the kernel SOURCE with a host compiler and runs it lane by lane
// (lanes as fibers or host threads, hip_emul.h) so that indexing, ordering and hand-off protocols are checked against the oracle
// before GPU time is spent.  It is not a second backend: nothing of it is in libturbometrics_hip.so, and there is no CPU path.
// Kernel bodies (tm_device_math.h, tm_kernels.h, tm_ssim_kernels.h) carry no #ifdef of their own.
#pragma once

#ifdef TM_EMULATE
// =====================================================================================================================
// test tier: host stand-ins (tests/emul/hip_emul.h declares the wave-level helpers tm_wave_sum6, tm_shfl_xor ... )
// =====================================================================================================================
#include "hip_emul.h"
#include <stdint.h>

namespace tmdev {
struct tm_f2 { float x, y; };
static inline tm_f2 operator*(tm_f2 a, tm_f2 b) { return {a.x * b.x, a.y * b.y}; }
static inline tm_f2 operator-(tm_f2 a, tm_f2 b) { return {a.x - b.x, a.y - b.y}; }
static inline tm_f2 operator+(tm_f2 a, tm_f2 b) { return {a.x + b.x, a.y + b.y}; }
static inline tm_f2 operator-(tm_f2 a) { return {-a.x, -a.y}; }
static inline tm_f2 f2_fma(tm_f2 a, tm_f2 b, tm_f2 c) { return {fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
static inline float tm_fract_pos(float s) { return s - floorf(s); }
} // namespace tmdev

// the emulator runs its lanes one after the other or as fibers: "any lane" is always taken (every use computes the same bits on
// either side of the branch), "all lanes" is this lane's own answer (the poll loops it guards re-check per lane)
#define TM_WAVE_ANY(c) true
#define TM_WAVE_ALL(c) (c)
#define TM_NO_IF_CONVERSION() ((void)0)
#define TM_WAVES_PER_SIMD(n)
#define TM_LDS_BARRIER() __syncthreads()
#define TM_GLOBAL_AS
#define TM_PIN_SGPR(v) ((void)0)
#define TM_KEEP_IN_VGPR(v) ((void)0)
#define TM_SETPRIO(n) ((void)0)
#define TM_SLEEP(n) ((void)0)
#define TM_EF_SPIN_PAUSE() ::tm_emul_yield()
#define TM_EF_LDS_FENCE() __atomic_thread_fence(__ATOMIC_SEQ_CST)

struct alignas(16) tm_f4 { float x, y, z, w; };
struct alignas(8) tm_g2 { float x, y; };
struct alignas(8) tm_u2 { unsigned x, y; };
static inline unsigned tm_mul24(unsigned a, unsigned b) { return a * b; }
static inline float tm_swap1(float v) { return tm_shfl_xor(v, 1); }
// {value of the even lane, value of the odd lane} of this lane's pair; `odd`: this lane is the odd one
static inline void tm_pair_values(float v, bool odd, float &even_v, float &odd_v)
{
    const float other = tm_swap1(v);
    even_v = odd ? other : v;
    odd_v = odd ? v : other;
}
// IEEE quotients (the device forms below are the compiler's own division sequence minus the range fix-ups: same bits)
static inline float tm_div_inrange(float n, float d) { return n / d; }
static inline float tm_rcp_inrange(float d) { return 1.0f / d; }
// tagged 64-bit hand-off words between workgroups
static inline void tm_ll_store(unsigned long long *p, float v, unsigned tag) { *(volatile unsigned long long *)p = ((unsigned long long)tag << 32) | __float_as_uint(v); }
static inline unsigned long long tm_ll_load(const unsigned long long *p) { return *(const volatile unsigned long long *)p; }

#else
// =====================================================================================================================
// gfx950 (the product)
// =====================================================================================================================
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tmdev {
// two-lane f32 vectors: v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 work on two floats per lane in one VALU instruction
typedef float tm_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ tm_f2 f2_fma(tm_f2 a, tm_f2 b, tm_f2 c) { return __builtin_elementwise_fma(a, b, c); }
// s - floor(s) for s >= 0 (exact): v_fract_f32
__device__ __forceinline__ float tm_fract_pos(float s) { return __builtin_amdgcn_fractf(s); }
} // namespace tmdev

// "does any / every lane of the wave ..." -- wave-uniform branches
#define TM_WAVE_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)
#define TM_WAVE_ALL(c) (__builtin_amdgcn_ballot_w64(c) == ~0ull)
#define TM_NO_IF_CONVERSION() asm volatile("; rare path") /* keeps the compiler from turning the uniform branch into selects */
#define TM_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n))) /* holds the register allocation to 512 / n VGPRs */
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding GLOBAL store of the wave
// (s_waitcnt vmcnt(0)); the ingest kernel only ever exchanges data through LDS, and waiting ~2 us for store acknowledgements at
// each of its barriers was most of a workgroup's lifetime.
#define TM_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define TM_GLOBAL_AS __attribute__((address_space(1)))
// Pin a wave-uniform 64-bit value into an SGPR pair: the empty asm is opaque to LLVM, which otherwise re-associates
// base + row * pitch + lane into a per-lane 64-bit address for every load of a window (2 VGPRs and a v_lshl_add_u64 each) instead
// of selecting the scalar-base + 32-bit-lane-offset form of global_load / global_store.
#define TM_PIN_SGPR(v) asm("" : "+s"(v))
// "this value is needed HERE": a load still pending at a loop header would make the compiler wait for everything outstanding --
// the previous iteration's stores included -- at the top of every iteration
#define TM_KEEP_IN_VGPR(v) asm volatile("" : "+v"(v))
#define TM_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#define TM_SLEEP(n) __builtin_amdgcn_s_sleep(n)
#define TM_EF_SPIN_PAUSE() __builtin_amdgcn_s_sleep(1)
#define TM_EF_LDS_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")

typedef float tm_f4 __attribute__((ext_vector_type(4))); // plain vector: assignable through address_space(1)
typedef float tm_g2 __attribute__((ext_vector_type(2)));
typedef unsigned tm_u2 __attribute__((ext_vector_type(2)));

// wave-level sum helpers; return true on the lane that ends up holding the total
__device__ __forceinline__ bool tm_wave_sum6(double (&a)[6])
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) a[k] += __shfl_down(a[k], off, 64);
    }
    return (threadIdx.x & 63) == 0;
}
__device__ __forceinline__ float tm_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ unsigned tm_shfl_xor_u32(unsigned v, int mask) { return (unsigned)__shfl_xor((int)v, mask, 64); }
__device__ __forceinline__ bool tm_wave_sum_u32x3(unsigned (&v)[3])
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] += __shfl_down(v[k], off, 64);
    }
    return ((threadIdx.x + threadIdx.y * blockDim.x) & 63) == 0;
}

// lane-dependent row * pitch products: v_mul_u32_u24 (full rate) instead of the 64-bit / 32-bit integer multiplies (quarter rate)
// that size_t arithmetic compiles to.  Both factors are below 2^24 and the product below 2^32: rows <= 16 384, pitches of the
// engine's own planes <= 2^16 floats, and tm_engine_set_frame_* refuses surfaces of 4 GB and more.
__device__ __forceinline__ unsigned tm_mul24(unsigned a, unsigned b) { return __umul24(a, b); }
// lane ^ 1 through DPP quad_perm [1, 0, 3, 2]: one full-rate VALU move, no LDS crossbar
__device__ __forceinline__ float tm_swap1(float v)
{
    return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0xB1, 0xF, 0xF, true));
}
// {value of the even lane, value of the odd lane} of this lane's pair: two DPP quad broadcasts
__device__ __forceinline__ void tm_pair_values(float v, bool, float &even_v, float &odd_v)
{
    even_v = __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0xA0, 0xF, 0xF, true)); // quad_perm [0, 0, 2, 2]
    odd_v = __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0xF5, 0xF, 0xF, true));  // quad_perm [1, 1, 3, 3]
}
// n / d and 1 / d for operands far from the ends of the exponent range: the sequence the compiler emits for an IEEE division --
// reciprocal, one Newton step on it, quotient, two residual corrections -- without v_div_scale / v_div_fixup, which only act on
// operands that need rescaling or are special: same operations on the same values, hence the same correctly rounded quotient, 8
// instead of 12 instructions (numerator 1: the first product is exact and disappears).
__device__ __forceinline__ float tm_div_inrange(float n, float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float r1 = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
    const float q0 = n * r1;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, n), r1, q0);
    return __builtin_fmaf(__builtin_fmaf(-d, q1, n), r1, q1);
}
__device__ __forceinline__ float tm_rcp_inrange(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float r1 = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
    const float q1 = __builtin_fmaf(__builtin_fmaf(-d, r1, 1.0f), r1, r1);
    return __builtin_fmaf(__builtin_fmaf(-d, q1, 1.0f), r1, q1);
}
// tagged 64-bit hand-off words between workgroups: relaxed agent-scope atomics (the tag IS the flag: a word is either the old
// {tag, value} or the new one, never a mix)
__device__ __forceinline__ void tm_ll_store(unsigned long long *p, float v, unsigned tag)
{
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long tm_ll_load(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#endif
