// lm_common.h -- shared declarations of the MI355X-native LectureMath hot path (gfx950 only).
//
// Data layout in HBM for one batch of B binary frames of W x H pixels (rows of all frames are
// contiguous, R = B*H rows):
//   bits     u64 [R][WW]    bit-packed foreground (bit x&63 of word x>>6; WW = ceil(W/64)), 1 bit / px
//   starts   u64 [R][WW]    bit set where a horizontal run of foreground begins
//   prefix   u16 [R][WW]    number of run starts in the row left of the word
//   rowoff   u32 [R]        band * SLOT + number of runs of the band above the row (run ids are band-structured:
//                           gid = band * SLOT + local index, monotone in raster order)
//   parent   i32 [B][CAP]   union-find forest over runs (frame-relative run ids, root = smallest id)
//   rootbits u64 [B][CAP/64] root flags; wordprefix u32 [B][CAP/64] their exclusive popcount scan
//   final    i32 [B][CAP]   1-based scipy-ordered label of every run
// A "run" is a maximal horizontal segment of foreground pixels; runs are numbered in raster order,
// so the smallest run id of a component starts at the component's first pixel in raster order and
// numbering roots in id order reproduces scipy.ndimage.label's numbering (labeler.py:126).
// CAP = nbands * SLOT with SLOT = 64 * ceil(W/2) (worst-case runs of a 64-row band), so there is no overflow path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef LM_HIP_EMULATED
#define LM_HIP_EMULATED 0
#endif

typedef float lm_f32x16 __attribute__((ext_vector_type(16)));
typedef float lm_f32x4 __attribute__((ext_vector_type(4)));

#define LM_OK 0
#define LM_ERR_ARG 1
#define LM_ERR_HIP 2
#define LM_ERR_CAPACITY 3
#define LM_ERR_STATE 4

#define LM_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            lm_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return LM_ERR_HIP;                                                             \
        }                                                                                  \
    } while (0)

void lm_set_error(const char* fmt, ...);

struct LmGeom {
    int W, H;
    int WW;        // 64-bit words per row
    int cap;       // per-frame run-id space / label capacity: nbands * slot
};

// Device workspace for the per-frame CC path (batch of up to max_batch frames).
struct LmCtx {
    LmGeom g;
    int max_batch;
    int device;
    // run structures
    uint64_t* bits;
    uint64_t* starts;
    uint16_t* prefix;
    uint32_t* rowoff;        // [R] band * slot + runs of the band above the row
    uint32_t* rowcnt;        // [R] runs of the row (lm_k_pack_rows -> lm_k_band)
    int32_t* band_runs;      // [B][nbands] runs per band
    uint32_t* band_base;     // [B][nbands] labels (roots) in the bands above
    uint32_t* band_roots;    // [B][nbands] roots per band
    uint8_t* band_fallback;  // [B][nbands] 1 = forest too large for LDS, unions done in L2
    unsigned* mid_sync;      // [2 queues][1 + 2 * B] ticket + per-frame arrival counters of lm_k_middle
    int nbands, slot;        // bands of band_rows rows; slot = id space per band (worst-case runs, multiple of 64)
    int band_rows;           // 32 up to 2048 px wide, 16 above (lm_cc_kernels.hip)
    int32_t* parent;         // [B][cap]
    int32_t* final_label;    // [B][cap]
    int32_t* n_labels;       // [B]
    unsigned long long* rootbits;   // [B][ceil(cap/64)] bit i set <=> run i is a root
    uint32_t* wordprefix;    // [B][ceil(cap/64)] roots before word j
    // per-label statistics (CC_AgeBoundaries order), [B][cap] each
    int32_t* st_min_y;
    int32_t* st_max_y;
    int32_t* st_min_x;
    int32_t* st_max_x;
    int32_t* st_count;
    // kept-CC selection staging, [B][cap]
    int32_t* kept_label;     // label-1 of the k-th kept CC of the frame
    uint32_t* kept_cropoff;  // frame-relative crop word offset
    int32_t* frame_kept;     // [B]
    uint32_t* frame_cropwords;  // [B]
    int last_batch;          // frames in the most recent lm_label_batch
    int last_fused;          // 1: that batch came through lm_k_pack_rows_logits (threshold fused into the row packing)
    int stats_fresh;         // 1: the statistics arrays of that batch hold their initial values (written by the numbering kernel)
    // host-pointer convenience path (drop-in entry points): staging buffers
    uint8_t* stage_u8;
    int32_t* stage_i32;
    float* stage_f32;
    size_t stage_px;
    // optional live timing of the labelling launch sequence (bench.py roofline): hipEvent pairs per call
    int profiling;
    void* prof;     // LmProfile*, owned
    // second queue of lm_label_batch (parts of a batch side by side), created on first use
    void* aux_stream;
    void* ev_fork;
    void* ev_join;
};

#define LM_LABEL_FUSED_MIDDLE_DEFAULT 0     // 1: lm_k_middle (one launch, per-frame rendezvous) instead of seam / flatten / apply launches; env LM_LABEL_FUSED_MIDDLE overrides
// parts a batch is labelled in, side by side on two queues (env LM_LABEL_PARTS overrides: 1..8).  Two parts shorten the launch
// sequence where it shares the GPU with other work (0.39 of the HBM peak in the pipeline's timed region instead of 0.30-0.32) by
// filling the gaps of its latency-bound middle with the other half of the batch -- but that is where the OTHER kernels of the
// pipeline (matching, step 03) would have run: the whole pipeline is 14 % faster with one part (68 k against 59 k frames/s), so
// one part is the default.  Alone the two forms take the same time (331 / 337 us per 64 frames).
#define LM_LABEL_PARTS 1
#define LM_LABEL_PART_MIN 8     // ... as long as every part has at least this many frames

// ---------------------------------------------------------------- device helpers
#if LM_HIP_EMULATED
#define LM_DEV static inline
#else
#define LM_DEV __device__ __forceinline__
#endif

LM_DEV int lm_lane() { return (int)(threadIdx.x & 63); }

// hides a register's value from the optimiser at this point (stops loop-invariant code motion of what is derived from it)
#if LM_HIP_EMULATED
#define LM_OPAQUE(x) ((void)0)
#else
#define LM_OPAQUE(x) asm volatile("" : "+v"(x))
#endif

// All lanes of the wave have executed what precedes (data exchange through LDS inside ONE wave).  The hardware runs a wave in
// lockstep and serves its LDS instructions in order, so this is a compiler fence there; the CPU emulator runs the lanes one after
// another and needs a real rendezvous (a wave collective).
#if LM_HIP_EMULATED
#define LM_WAVE_SYNC() ((void)__ballot(1))
#else
// wave_barrier alone is declared memory-free to the optimiser: the release / acquire fences at wavefront scope are what forbid
// moving the LDS loads of one side above the LDS stores of the other (they emit no instruction beyond the LDS counter wait)
#define LM_WAVE_SYNC()                                           \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)
#endif

// a line the instruction scheduler may not move anything across
#if LM_HIP_EMULATED
#define LM_SCHED_BARRIER() ((void)0)
#else
#define LM_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif

// a value that is the same in every lane of the wave, said so to the compiler (scalar registers, scalar address arithmetic)
#if LM_HIP_EMULATED
#define LM_UNIFORM(x) (x)
#else
#define LM_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

template <class T> LM_DEV T lm_wave_incl_scan(T v)
{
    int lane = lm_lane();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

template <class T> LM_DEV T lm_wave_sum(T v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// Exclusive scan across a block of BLOCK threads (BLOCK multiple of 64, <= 1024); every thread calls.
template <int BLOCK> LM_DEV unsigned lm_block_excl_scan(unsigned v, unsigned* total)
{
    __shared__ unsigned wsum[BLOCK / 64];
    __shared__ unsigned wtot;
    const int lane = lm_lane(), wid = (int)(threadIdx.x >> 6);
    unsigned incl = lm_wave_incl_scan(v);
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    if (wid == 0) {
        unsigned t = (lane < BLOCK / 64) ? wsum[lane] : 0u;
        unsigned ti = lm_wave_incl_scan(t);
        if (lane < BLOCK / 64) wsum[lane] = ti - t;
        if (lane == BLOCK / 64 - 1) wtot = ti;
    }
    __syncthreads();
    unsigned res = wsum[wid] + incl - v;
    *total = wtot;
    __syncthreads();
    return res;
}

// Workgroup barrier that only orders LDS traffic: waits for this wave's outstanding LDS operations (lgkmcnt), not for its
// global loads / stores (vmcnt) as __syncthreads() does.  For phases that communicate through LDS alone while global loads
// (prefetch) and write-through stores stay in flight.
LM_DEV void lm_lds_barrier()
{
#if LM_HIP_EMULATED
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// lm_block_excl_scan with LDS-only barriers (the scan itself communicates through LDS only).
template <int BLOCK> LM_DEV unsigned lm_block_excl_scan_lds(unsigned v, unsigned* total)
{
    __shared__ unsigned wsum[BLOCK / 64];
    __shared__ unsigned wtot;
    const int lane = lm_lane(), wid = (int)(threadIdx.x >> 6);
    unsigned incl = lm_wave_incl_scan(v);
    if (lane == 63) wsum[wid] = incl;
    lm_lds_barrier();
    if (wid == 0) {
        unsigned t = (lane < BLOCK / 64) ? wsum[lane] : 0u;
        unsigned ti = lm_wave_incl_scan(t);
        if (lane < BLOCK / 64) wsum[lane] = ti - t;
        if (lane == BLOCK / 64 - 1) wtot = ti;
    }
    lm_lds_barrier();
    unsigned res = wsum[wid] + incl - v;
    *total = wtot;
    lm_lds_barrier();
    return res;
}

// four consecutive int32 from a 4-byte aligned address as ONE load instruction (global_load_dwordx4; gfx950 takes dword-aligned
// vector loads): a lane that needs the labels of runs id .. id + 3 issues one request instead of four
typedef int lm_i32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
LM_DEV void lm_load4(const int32_t* p, int (&out)[4])
{
    const lm_i32x4_a4 v = *(const lm_i32x4_a4*)p;
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
}

LM_DEV uint64_t lm_lowmask_incl(int p) { return (p >= 63) ? ~0ull : ((1ull << (p + 1)) - 1ull); }
LM_DEV uint64_t lm_lowmask_excl(int p) { return (p <= 0) ? 0ull : ((p >= 64) ? ~0ull : ((1ull << p) - 1ull)); }
