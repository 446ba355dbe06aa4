// tests/emul/hip_emul.h -- TEST INFRASTRUCTURE ONLY.
// A minimal host-side stand-in for the HIP device environment so that the kernel SOURCE in
// turbo-metrics_amd/csrc/tm_kernels.h can be executed lane by lane on the CPU inside the
// `-m "not gpu"` test tier (index/ordering logic is checked against the oracle before GPU time is
// spent).  It is compiled only into tests/emul/libtm_emul.so; the product library never sees it.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
// a wavefront executes in lockstep on the device; here its 64 lanes are 64 host threads and this is the
// rendezvous that stands in for that lockstep wherever lanes exchange data through LDS
void tm_emul_wave_barrier();
#define __builtin_amdgcn_wave_barrier() tm_emul_wave_barrier()
#define __builtin_amdgcn_readfirstlane(x) (x)
// role-waves of the split column pass share no data; the emulator runs them one after the other, so the
// drift-limiting barrier has nothing to do here
#define __builtin_amdgcn_s_barrier() ((void)0)
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
void tm_emul_syncthreads();
void tm_emul_yield(); // a lane polling memory that another wave of its workgroup writes: lets the other lanes run
#define __syncthreads() tm_emul_syncthreads()

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint3_ { unsigned x, y, z; };
extern thread_local uint3_ threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float a, float b) { return {a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }
static inline float __uint_as_float(unsigned v) { float f; memcpy(&f, &v, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned v; memcpy(&v, &f, 4); return v; }
static inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
static inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
using std::min;
using std::max;
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

// lanes of one wave run back to back (x fastest), so a running total per wave is enough
bool tm_wave_sum6(double (&a)[6]);
bool tm_wave_sum_u32x3(unsigned (&v)[3]);
float tm_shfl_xor(float v, int mask); // lockstep wave emulation only
unsigned tm_shfl_xor_u32(unsigned v, int mask);
struct alignas(8) uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned a, unsigned b) { return {a, b}; }
