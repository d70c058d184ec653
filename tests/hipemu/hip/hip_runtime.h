// tests/hipemu/hip/hip_runtime.h -- TEST-ONLY stand-in for <hip/hip_runtime.h>.
//
// Lets the product's .hip sources be compiled with g++ and executed on the CPU so that kernel
// LOGIC (indexing, union-find, scans, wave collectives) can be debugged in the build container,
// which has no GPU.  Every GPU thread is a ucontext fiber; blocks run one after another;
// __syncthreads() and the wave collectives (__shfl*, __ballot, mfma) are rendezvous points.
// It models none of the hardware's memory system or timing and is NEVER loaded by the product:
// lecturemath_amd/_lib.py only opens liblecturemath_hip.so.  tests/ builds liblecturemath_emu.so
// from the same sources through tests/hipemu/Makefile and hands its path to the binding explicitly.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define LM_HIP_EMULATED 1

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct int2 { int x, y; };
static inline int2 make_int2(int a, int b) { return int2{a, b}; }
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
static inline int4 make_int4(int a, int b, int c, int d) { return int4{a, b, c, d}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
#define hipErrorOutOfMemory 2
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __restrict__ __restrict
#define __launch_bounds__(...)

namespace hipemu {
struct Fiber;
extern Fiber* g_cur;
struct ThreadCtx { dim3 tid, bid, bdim, gdim; };
extern ThreadCtx g_ctx;
void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t shmem);
void syncthreads();
// wave rendezvous: every live lane of the calling lane's wave deposits `v`; returns the 64 deposited
// values (dead lanes: 0) and the mask of participating lanes.
const unsigned long long* wave_gather(unsigned long long v, unsigned long long* active_mask);
int lane_id();
}  // namespace hipemu

#define threadIdx (hipemu::g_ctx.tid)
#define blockIdx (hipemu::g_ctx.bid)
#define blockDim (hipemu::g_ctx.bdim)
#define gridDim (hipemu::g_ctx.gdim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch([&]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block), (size_t)(shmem))

static inline void __syncthreads() { hipemu::syncthreads(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// ---- memory API
static inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = 0) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
#define hipStreamNonBlocking 1
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)(uintptr_t)1; return hipSuccess; }   // launches are synchronous
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = 0) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// ---- bit intrinsics
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline unsigned __brev(unsigned v) { unsigned r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }

// ---- atomics (single OS thread: plain read-modify-write)
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicXor(T* p, T v) { T o = *p; *p = o ^ v; return o; }
template <class T> static inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

// ---- wave collectives (all live lanes of the wave must reach the same call)
static inline unsigned long long __ballot(int pred)
{
    unsigned long long act;
    const unsigned long long* v = hipemu::wave_gather(pred ? 1ull : 0ull, &act);
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) if (((act >> i) & 1) && v[i]) m |= 1ull << i;
    return m;
}
static inline unsigned long long __activemask_emu()
{
    unsigned long long act;
    hipemu::wave_gather(0, &act);
    return act;
}
template <class T> static inline T __shfl_generic(T val, int src)
{
    static_assert(sizeof(T) <= 8, "shuffle payload too large");
    unsigned long long raw = 0, act;
    memcpy(&raw, &val, sizeof(T));
    const unsigned long long* v = hipemu::wave_gather(raw, &act);
    unsigned long long got = v[src & 63];
    T out;
    memcpy(&out, &got, sizeof(T));
    return out;
}
template <class T> static inline T __shfl(T val, int src, int width = 64)
{
    int l = hipemu::lane_id();
    int base = l & ~(width - 1);
    return __shfl_generic(val, base + (src & (width - 1)));
}
template <class T> static inline T __shfl_up(T val, unsigned d, int width = 64)
{
    int l = hipemu::lane_id();
    int base = l & ~(width - 1);
    int s = l - (int)d;
    return __shfl_generic(val, s < base ? l : s);
}
template <class T> static inline T __shfl_down(T val, unsigned d, int width = 64)
{
    int l = hipemu::lane_id();
    int base = l & ~(width - 1);
    int s = l + (int)d;
    return __shfl_generic(val, s >= base + width ? l : s);
}
template <class T> static inline T __shfl_xor(T val, int m, int width = 64)
{
    int l = hipemu::lane_id();
    int base = l & ~(width - 1);
    int s = l ^ m;
    return __shfl_generic(val, (s >= base + width || s < base) ? l : s);
}

// ---- math



// ---- f32 MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32), k-ordered fmaf chain as on gfx950
typedef float lm_f32x16 __attribute__((ext_vector_type(16)));
typedef float lm_f32x4 __attribute__((ext_vector_type(4)));
lm_f32x16 hipemu_mfma_32x32x2f32(float a, float b, lm_f32x16 c);
lm_f32x4 hipemu_mfma_16x16x4f32(float a, float b, lm_f32x4 c);
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu_mfma_32x32x2f32(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu_mfma_16x16x4f32(a, b, c)
