// lm_api.hip -- C-ABI entry points (include/lecturemath_amd.h) over the kernels of this directory.
// Unity translation unit: the kernel files are included so launches see the definitions.
#include "lm_common.h"
#include "lm_stream.h"

#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lecturemath_amd.h"

#include "lm_cc_kernels.hip"
#include "lm_match_kernels.hip"
#include "lm_match_batch.hip"
#include "lm_group.hip"
#include "lm_fcn.hip"
#include "lm_fcn2.hip"

// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void lm_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* lm_last_error(void) { return g_err; }
extern "C" int lm_abi_version(void) { return 1; }
extern "C" int lm_is_device_build(void) { return LM_HIP_EMULATED ? 0 : 1; }

// LM_DEBUG_BAND_PHASES=2 makes lm_k_band leave every band's unions to lm_k_band_union_global (results unchanged); default 3
static int lm_debug_band_phases()
{
    static int v = -1;
    if (v < 0) { const char* e = getenv("LM_DEBUG_BAND_PHASES"); v = e ? atoi(e) : 3; if (v < 2 || v > 3) v = 3; }
    return v;
}

// LM_DEBUG_BAND_STAMPS=<file>: lm_k_band leaves wall-clock stamps per phase and workgroup (8 x u64 each); the last launch's are
// written to <file> by lm_label_counts (tools/band_phases.py reads them).  Tuning aid only.
static unsigned long long* g_band_stamps = nullptr;
static size_t g_band_stamps_n = 0;
static unsigned long long* lm_debug_band_stamps(int nbands, int n_frames)
{
    static const char* path = getenv("LM_DEBUG_BAND_STAMPS");
    if (!path) return nullptr;
    const size_t need = (size_t)nbands * n_frames * 8;
    if (need > g_band_stamps_n) {
        if (g_band_stamps) (void)hipFree(g_band_stamps);
        g_band_stamps = nullptr;
        if (hipMalloc((void**)&g_band_stamps, need * sizeof(unsigned long long)) != hipSuccess) return nullptr;
        g_band_stamps_n = need;
    }
    return g_band_stamps;
}

static inline unsigned lm_blocks(long long work_items, int block, int max_blocks = 8192)
{
    long long b = (work_items + block - 1) / block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (unsigned)b;
}

template <class T> static int lm_alloc(T** p, size_t count)
{
    LM_HIP(hipMalloc((void**)p, count * sizeof(T) + 64));
    return LM_OK;
}

// ------------------------------------------------------------------------------------------------
// live profiling (event pairs around the labelling launch sequence)
// ------------------------------------------------------------------------------------------------
#include <mutex>
#include <vector>
struct LmProfile {
    std::vector<hipEvent_t> ev;     // pairs: start, stop
    std::vector<int> frames;
    size_t used = 0;                // pairs in use since the last read
};

static void lm_profile_free(LmCtx* c)
{
    LmProfile* p = (LmProfile*)c->prof;
    if (!p) return;
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    delete p;
    c->prof = nullptr;
}

extern "C" int lm_ctx_set_profiling(LmCtx* c, int enable)
{
    if (!c) { lm_set_error("lm_ctx_set_profiling: null ctx"); return LM_ERR_ARG; }
    c->profiling = enable ? 1 : 0;
    if (enable && !c->prof) c->prof = new LmProfile();
    return LM_OK;
}

extern "C" int lm_ctx_profile_read(LmCtx* c, double* total_ms, int64_t* calls, int64_t* frames)
{
    if (!c || !total_ms || !calls || !frames) { lm_set_error("lm_ctx_profile_read: bad arguments"); return LM_ERR_ARG; }
    *total_ms = 0.0; *calls = 0; *frames = 0;
    LmProfile* p = (LmProfile*)c->prof;
    if (!p) return LM_OK;
    for (size_t i = 0; i < p->used; i++) {
        float ms = 0.f;
        LM_HIP(hipEventSynchronize(p->ev[2 * i + 1]));
        LM_HIP(hipEventElapsedTime(&ms, p->ev[2 * i], p->ev[2 * i + 1]));
        *total_ms += ms;
        *frames += p->frames[i];
    }
    *calls = (int64_t)p->used;
    p->used = 0;
    return LM_OK;
}

static int lm_profile_mark(LmCtx* c, hipStream_t st, bool start, int n_frames)
{
    LmProfile* p = (LmProfile*)c->prof;
    if (!c->profiling || !p) return LM_OK;
    if (start) {
        if (p->ev.size() < 2 * (p->used + 1)) {
            hipEvent_t a, b;
            LM_HIP(hipEventCreate(&a));
            LM_HIP(hipEventCreate(&b));
            p->ev.push_back(a);
            p->ev.push_back(b);
            p->frames.push_back(0);
        }
        p->frames[p->used] = n_frames;
        LM_HIP(hipEventRecord(p->ev[2 * p->used], st));
    } else {
        LM_HIP(hipEventRecord(p->ev[2 * p->used + 1], st));
        p->used++;
    }
    return LM_OK;
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
extern "C" void lm_ctx_destroy(LmCtx* c)
{
    if (!c) return;
    lm_profile_free(c);
    void* ptrs[] = {c->bits, c->starts, c->prefix, c->rowoff, c->rowcnt, c->band_runs, c->band_base, c->band_roots, c->band_fallback, c->parent, c->final_label,
                    c->n_labels, c->rootbits, c->wordprefix, c->mid_sync, c->st_min_y, c->st_max_y, c->st_min_x, c->st_max_x, c->st_count, c->kept_label,
                    c->kept_cropoff, c->frame_kept, c->frame_cropwords, c->stage_u8, c->stage_i32, c->stage_f32};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (c->aux_stream) { (void)hipStreamDestroy((hipStream_t)c->aux_stream); (void)hipEventDestroy((hipEvent_t)c->ev_fork); (void)hipEventDestroy((hipEvent_t)c->ev_join); }
    delete c;
}

extern "C" LmCtx* lm_ctx_create(int width, int height, int max_batch)
{
    if (width <= 0 || height <= 0 || max_batch <= 0 || max_batch > 1024 || width > 32767 || height > 32767) {
        lm_set_error("lm_ctx_create: bad arguments (width=%d height=%d max_batch=%d; batch <= 1024, sides <= 32767)",
                     width, height, max_batch);
        return nullptr;
    }
    LmCtx* c = new LmCtx();
    memset(c, 0, sizeof(*c));
    c->g.W = width;
    c->g.H = height;
    c->g.WW = (width + 63) / 64;
    c->band_rows = (c->g.WW > 32) ? 16 : LM_BAND_ROWS_MAX;
    if (const char* e = getenv("LM_BAND_ROWS")) { const int v = atoi(e); if (v == 8 || v == 16 || v == 32 || v == 64) c->band_rows = v; }      // tuning experiments
    c->nbands = (height + c->band_rows - 1) / c->band_rows;
    c->slot = ((c->band_rows * ((width + 1) / 2)) + 63) & ~63;       // worst-case runs of one band, multiple of 64
    c->g.cap = c->nbands * c->slot;
    c->max_batch = max_batch;
    (void)hipGetDevice(&c->device);
    const size_t R = (size_t)max_batch * height, RW = R * c->g.WW, BC = (size_t)max_batch * c->g.cap;
    int rc = LM_OK;
    rc |= lm_alloc(&c->bits, RW);
    rc |= lm_alloc(&c->starts, RW);
    rc |= lm_alloc(&c->prefix, RW);
    rc |= lm_alloc(&c->rowoff, R);
    rc |= lm_alloc(&c->rowcnt, R);
    rc |= lm_alloc(&c->band_runs, (size_t)max_batch * c->nbands);
    rc |= lm_alloc(&c->band_base, (size_t)max_batch * c->nbands);
    rc |= lm_alloc(&c->band_roots, (size_t)max_batch * c->nbands);
    rc |= lm_alloc(&c->band_fallback, (size_t)max_batch * c->nbands);
    rc |= lm_alloc(&c->parent, BC);
    rc |= lm_alloc(&c->final_label, BC);     // lm_k_stats' lm_load4 may read up to 12 bytes past B * cap: inside lm_alloc's 64 bytes of slack
    rc |= lm_alloc(&c->n_labels, (size_t)max_batch);
    rc |= lm_alloc(&c->rootbits, (size_t)max_batch * (c->g.cap / 64));
    rc |= lm_alloc(&c->wordprefix, (size_t)max_batch * (c->g.cap / 64));
    rc |= lm_alloc(&c->st_min_y, BC);
    rc |= lm_alloc(&c->st_max_y, BC);
    rc |= lm_alloc(&c->st_min_x, BC);
    rc |= lm_alloc(&c->st_max_x, BC);
    rc |= lm_alloc(&c->st_count, BC);
    rc |= lm_alloc(&c->kept_label, BC);
    rc |= lm_alloc(&c->kept_cropoff, BC);
    rc |= lm_alloc(&c->frame_kept, (size_t)max_batch);
    rc |= lm_alloc(&c->frame_cropwords, (size_t)max_batch);
    rc |= lm_alloc(&c->mid_sync, 2 * (1 + 2 * (size_t)max_batch));
    if (rc != LM_OK) {
        lm_ctx_destroy(c);
        return nullptr;
    }
    return c;
}

// ------------------------------------------------------------------------------------------------
// threshold
// ------------------------------------------------------------------------------------------------
// x* of lm_k_thr_edge per threshold value, found once per process (0: not yet, 1: usable, 2: keep the formula kernel)
static float g_thr_edge[256];
static int g_thr_state[256];
static std::mutex g_thr_mutex;

static int lm_threshold_edge(int thr, hipStream_t st, float* edge, int* state)
{
    std::lock_guard<std::mutex> lock(g_thr_mutex);
    if (!g_thr_state[thr]) {
        float* d = nullptr;
        float h[2] = {0.0f, 0.0f};
        LM_HIP(hipMalloc(&d, 2 * sizeof(float)));
        hipLaunchKernelGGL(lm_k_thr_edge, dim3(1), dim3(64), 0, st, thr, d);
        const hipError_t e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
        const hipError_t e2 = hipStreamSynchronize(st);
        (void)hipFree(d);
        LM_HIP(e);
        LM_HIP(e2);
        g_thr_edge[thr] = h[0];
        g_thr_state[thr] = h[1] == 1.0f ? 1 : 2;
    }
    *edge = g_thr_edge[thr];
    *state = g_thr_state[thr];
    return LM_OK;
}

extern "C" int lm_threshold(const float* d_logits, uint8_t* d_out, int64_t n, int thr, int invert, void* stream)
{
    if (!d_logits || !d_out || n < 0) { lm_set_error("lm_threshold: bad arguments"); return LM_ERR_ARG; }
    if (n == 0) return LM_OK;
    const bool aligned = ((((uintptr_t)d_logits) & 15) == 0) && ((((uintptr_t)d_out) & 15) == 0);
    float edge = 0.0f;
    // the comparison form needs a threshold the sigmoid can straddle and 16-byte aligned buffers; its first use per threshold
    // value synchronises the stream once
    if (aligned && thr >= 1 && thr <= 255 && !getenv("LM_THRESHOLD_FORMULA")) {
        int state = 0;
        const int rc = lm_threshold_edge(thr, (hipStream_t)stream, &edge, &state);
        if (rc) return rc;
        if (state == 1) {
            hipLaunchKernelGGL(lm_k_threshold_cmp, dim3(lm_blocks((n + 4095) / 4096, 1, 1 << 20)), dim3(256), 0, (hipStream_t)stream, d_logits, d_out,
                               (long long)n, edge, invert ? 0xffu : 0u);
            LM_HIP(hipGetLastError());
            return LM_OK;
        }
    }
    hipLaunchKernelGGL(lm_k_threshold_invert, dim3(lm_blocks((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       d_logits, d_out, (long long)n, thr, invert ? 0xffu : 0u);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

extern "C" int lm_threshold_invert(const float* d_logits, uint8_t* d_out, int64_t n, int thr, void* stream)
{
    return lm_threshold(d_logits, d_out, n, thr, 1, stream);
}

// ------------------------------------------------------------------------------------------------
// labelling
// ------------------------------------------------------------------------------------------------
// The launch sequence of frames [f0, f0 + n) of a batch on stream `st`.  All the per-frame tables of the context are indexed by
// the frame's position in the batch and hold frame-relative ids, so a part of a batch is the same launches on shifted pointers.
// The frames come as {0, non-zero} bytes (d_binary) or -- logits != null -- as fp32 logits thresholded on the fly ((x >= edge ? 255 : 0) ^
// flip; d_binary, when given as well, receives that frame).
struct LmLabelSrc {
    const uint8_t* d_binary;
    const float* logits; float edge; unsigned flip; uint8_t* d_binary_out;
};

// runs of a band's union-find forest kept in LDS: 4,096 (16 KB) for bands of up to 32 rows, 8,192 for 64-row bands
static inline int lm_band_lds_runs(int band_rows) { return band_rows > 32 ? 2 * LM_BAND_LDS : LM_BAND_LDS; }

// phase: 0 = the whole sequence, 1 = the row packing only, 2 = everything behind it
static int lm_label_launch(LmCtx* c, const LmLabelSrc& src, int f0, int n, int32_t* d_labels, hipStream_t st, int phase = 0)
{
    const uint8_t* d_binary = src.d_binary;
    const LmGeom g = c->g;
    const int nbands = c->nbands, slot = c->slot;
    const int capw = g.cap / 64;
    const int band_lds = lm_band_lds_runs(c->band_rows);
    const size_t band_smem = (size_t)c->band_rows * g.WW * 18 + (size_t)band_lds * 4 + (65 + 64) * 4 + 64;
    const unsigned long long magic_ww = ((1ull << 40) / (unsigned)g.WW) + 1;
    const long long R = (long long)n * g.H;
    const size_t px = (size_t)g.W * g.H, r0 = (size_t)f0 * g.H, w0 = r0 * g.WW, b0 = (size_t)f0 * nbands, c0 = (size_t)f0 * g.cap, cw0 = (size_t)f0 * capw;
    if (phase == 2) goto rest;
    if (src.logits)
        hipLaunchKernelGGL(lm_k_pack_rows_logits, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, src.logits + f0 * px, src.edge, src.flip,
                           src.d_binary_out ? src.d_binary_out + f0 * px : nullptr, c->bits + w0, c->starts + w0, c->prefix + w0, c->rowcnt + r0, g.W, g.WW, R);
    else
        hipLaunchKernelGGL(lm_k_pack_rows, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, d_binary + f0 * px, c->bits + w0, c->starts + w0, c->prefix + w0,
                           c->rowcnt + r0, g.W, g.WW, R);
    if (phase == 1) { LM_HIP(hipGetLastError()); return LM_OK; }
rest:
    static const int band_threads = [] { const char* e = getenv("LM_BAND_THREADS"); const int v = e ? atoi(e) : 512; return (v == 128 || v == 256 || v == 512) ? v : 512; }();
    hipLaunchKernelGGL(lm_k_band, dim3(nbands, n), dim3(band_threads), band_smem, st, c->bits + w0, c->starts + w0, c->prefix + w0, c->rowcnt + r0,
                       c->rowoff + r0, c->band_runs + b0, c->parent + c0, c->band_fallback + b0, g.H, g.WW, slot, g.cap, lm_debug_band_phases(), magic_ww,
                       c->band_rows, f0 == 0 ? lm_debug_band_stamps(nbands, n) : nullptr, band_lds);
    const LmStatInit si = {c->st_min_y + c0, c->st_max_y + c0, c->st_min_x + c0, c->st_max_x + c0, c->st_count + c0, g.W, g.H};
    static const int fused_middle = [] { const char* e = getenv("LM_LABEL_FUSED_MIDDLE"); return e ? atoi(e) : LM_LABEL_FUSED_MIDDLE_DEFAULT; }();
#if !LM_HIP_EMULATED
    if (fused_middle) {
        // seam unions, flatten + flags, numbering in one launch with two per-frame rendezvous (lm_k_middle).  Two counter sets: part 0 (on
        // `st`) uses set 0, the other part (on the aux queue) set 1; each launch zeroes its set first and passes base values of 0
        // (lm_label_batch caps the parts at two while this switch is on)
        unsigned* sync = c->mid_sync + (size_t)(f0 ? 1 : 0) * (1 + 2 * (size_t)c->max_batch);
        LM_HIP(hipMemsetAsync(sync, 0, (1 + 2 * (size_t)n) * sizeof(unsigned), st));
        hipLaunchKernelGGL(lm_k_middle, dim3((unsigned)nbands * n), dim3(256), 0, st, c->bits + w0, c->starts + w0, c->prefix + w0, c->rowoff + r0,
                           c->band_fallback + b0, c->parent + c0, c->band_runs + b0, c->rootbits + cw0, c->wordprefix + cw0, c->band_roots + b0,
                           c->band_base + b0, c->n_labels + f0, c->final_label + c0, g.WW, g.H, g.cap, c->band_rows, slot, capw, nbands, sync, 0u, 0u, si);
    } else
#endif
    {
    hipLaunchKernelGGL(lm_k_seam_union, dim3(nbands, n), dim3(256), 0, st, c->bits + w0, c->starts + w0, c->prefix + w0, c->rowoff + r0,
                       c->band_fallback + b0, c->parent + c0, g.WW, g.H, g.cap, c->band_rows);
    hipLaunchKernelGGL(lm_k_flatten_flag, dim3(nbands, n), dim3(256), 0, st, c->parent + c0, c->band_runs + b0, c->rootbits + cw0, c->wordprefix + cw0,
                       c->band_roots + b0, slot, g.cap, capw);
    hipLaunchKernelGGL(lm_k_apply_labels, dim3(nbands, n), dim3(256), 0, st, c->parent + c0, c->band_runs + b0, c->rootbits + cw0, c->wordprefix + cw0,
                       c->band_roots + b0, c->band_base + b0, c->n_labels + f0, c->final_label + c0, slot, g.cap, capw, si);
    }
    if (d_labels) {
        const unsigned Q = (unsigned)(g.W + 3) / 4;
        const unsigned long long magic_q = ((1ull << 40) / Q) + 1;       // lm_fastdiv: exact for H * Q < 2^24
        const long long quads = (long long)g.H * Q;
        // one chunk of 64 * LM_WL_Q quads per wave (measured: work distribution moves this kernel by < 5 %, it runs at the
        // mixed read/write bandwidth of the device)
        const unsigned gx = (unsigned)((quads + 256 * LM_WL_Q - 1) / (256 * LM_WL_Q));
        hipLaunchKernelGGL(lm_k_write_labels, dim3(gx, n), dim3(256), 0, st, c->bits + w0, c->starts + w0, c->prefix + w0, c->rowoff + r0,
                           c->final_label + c0, d_labels + f0 * px, g.W, g.H, g.WW, g.cap, magic_q, 0);
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// Parts of a batch are labelled side by side: the middle of the sequence (band forests, seams, numbering) is a chain of
// latency-bound kernels on ~1/8 of the bytes, its two ends (row packing, the label image) stream at HBM rate -- with the parts on
// two queues the ends of one part run under the middle of the other.  The second queue belongs to the context and is joined back
// into the caller's stream before the call returns, so the caller's ordering is that of a single launch sequence.
static int lm_label_batch_src(LmCtx* c, const LmLabelSrc& src, int n_frames, int32_t* d_labels, void* stream)
{
    if (!c || (!src.d_binary && !src.logits) || n_frames <= 0 || n_frames > c->max_batch) {
        lm_set_error("lm_label_batch: bad arguments (n_frames=%d, max_batch=%d)", n_frames, c ? c->max_batch : -1);
        return LM_ERR_ARG;
    }
    const LmGeom g = c->g;
    hipStream_t st = (hipStream_t)stream;
    if ((long long)g.H * ((g.W + 3) / 4) >= (1ll << 24) && d_labels) { lm_set_error("lm_label_batch: frame too large for the label writer (H*W/4 must be < 2^24)"); return LM_ERR_ARG; }
#if !LM_HIP_EMULATED
    {
        // per call: the attribute belongs to the (function, device) pair and the call is cheap; a process-wide "already configured"
        // flag would skip it on a second device or race between caller threads
        const size_t band_smem = (size_t)c->band_rows * g.WW * 18 + (size_t)lm_band_lds_runs(c->band_rows) * 4 + (65 + 64) * 4 + 64;
        LM_HIP(hipFuncSetAttribute((const void*)lm_k_band, hipFuncAttributeMaxDynamicSharedMemorySize, (int)band_smem));
    }
#endif
    if (lm_profile_mark(c, st, true, n_frames)) return LM_ERR_HIP;
    static const int want_parts = [] { const char* e = getenv("LM_LABEL_PARTS"); const int v = e ? atoi(e) : LM_LABEL_PARTS; return (v >= 1 && v <= 8) ? v : LM_LABEL_PARTS; }();
    int parts = want_parts;
    while (parts > 1 && n_frames / parts < LM_LABEL_PART_MIN) parts--;
    // the experimental fused middle (LM_LABEL_FUSED_MIDDLE=1) has two rendezvous counter sets, one per queue: a third part would share
    // a set with a part running concurrently on the other queue (its memset zeroing counters workgroups spin on)
    if (parts > 2 && getenv("LM_LABEL_FUSED_MIDDLE") && atoi(getenv("LM_LABEL_FUSED_MIDDLE"))) parts = 2;
    if (parts > 1 && !c->aux_stream) {
        LM_HIP(hipStreamCreateWithFlags((hipStream_t*)&c->aux_stream, hipStreamNonBlocking));
        LM_HIP(hipEventCreateWithFlags((hipEvent_t*)&c->ev_fork, hipEventDisableTiming));
        LM_HIP(hipEventCreateWithFlags((hipEvent_t*)&c->ev_join, hipEventDisableTiming));
    }
    if (parts <= 1) {
        const int rc = lm_label_launch(c, src, 0, n_frames, d_labels, st);
        if (rc) return rc;
    } else {
        hipStream_t aux = (hipStream_t)c->aux_stream;
        // Two parts are STAGGERED: the second part starts when the first has packed its rows, so its packing (bandwidth bound) runs under
        // the first part's band / seam / numbering kernels (latency bound) and the first part's label writer under the second's middle.
        // Started together the halves ran in step -- both packings, both middles, both writers -- and nothing overlapped.
        static const int stagger = [] { const char* e = getenv("LM_LABEL_STAGGER"); return e ? atoi(e) : 0; }();
        int rc = LM_OK;
        const int n0 = n_frames / parts + (n_frames % parts ? 1 : 0);
        const bool stag = stagger && parts == 2;
        if (stag) rc = lm_label_launch(c, src, 0, n0, d_labels, st, 1);
        if (rc) return rc;
        LM_HIP(hipEventRecord((hipEvent_t)c->ev_fork, st));
        LM_HIP(hipStreamWaitEvent(aux, (hipEvent_t)c->ev_fork, 0));
        // from here on the second queue holds work: whatever fails below, it is joined back into the caller's stream before the
        // call returns (both queues touch the context's tables and d_binary)
        if (stag) {
            rc = lm_label_launch(c, src, n0, n_frames - n0, d_labels, aux, 0);
            if (rc == LM_OK) rc = lm_label_launch(c, src, 0, n0, d_labels, st, 2);
        } else {
            for (int k = 0, f0 = 0; k < parts && rc == LM_OK; k++) {
                const int n = n_frames / parts + (k < n_frames % parts ? 1 : 0);
                rc = lm_label_launch(c, src, f0, n, d_labels, (k & 1) ? aux : st);
                f0 += n;
            }
        }
        const hipError_t e1 = hipEventRecord((hipEvent_t)c->ev_join, aux);
        const hipError_t e2 = (e1 == hipSuccess) ? hipStreamWaitEvent(st, (hipEvent_t)c->ev_join, 0) : e1;
        if (e2 != hipSuccess) {
            (void)hipStreamSynchronize(aux);        // could not order the queues with an event: drain the second one instead
            if (rc == LM_OK) { lm_set_error("lm_label_batch: joining the second queue failed: %s", hipGetErrorString(e2)); rc = LM_ERR_HIP; }
        }
        if (rc) return rc;
    }
    if (lm_profile_mark(c, st, false, n_frames)) return LM_ERR_HIP;
    c->last_batch = n_frames;
    c->last_fused = src.logits ? 1 : 0;
    c->stats_fresh = 1;
    return LM_OK;
}

extern "C" int lm_label_batch(LmCtx* c, const uint8_t* d_binary, int n_frames, int32_t* d_labels, void* stream)
{
    LmLabelSrc src = {d_binary, nullptr, 0.0f, 0u, nullptr};
    return lm_label_batch_src(c, src, n_frames, d_labels, stream);
}

// Labelling straight from fp32 logits: binary = ((uint8)(sigmoid(x) * 255) >= thr ? 255 : 0), inverted when `invert` (the step-01 worker's
// 255 - binary), 4-connected labels of its non-zero pixels -- lm_threshold + lm_label_batch in one pass over the logits (4 B/px read
// instead of 4 B/px read + 1 B/px written + 1 B/px read).  d_binary may be null; when given it receives the {0, 255} frames.
// Falls back to the two calls (d_binary required then) when the row width is not a multiple of 4 pixels, the buffers are not 16-byte
// aligned or the threshold has no comparison form (lm_threshold_edge).
extern "C" int lm_label_batch_logits(LmCtx* c, const float* d_logits, int n_frames, int thr, int invert, uint8_t* d_binary, int32_t* d_labels, void* stream)
{
    if (!c || !d_logits || n_frames <= 0 || n_frames > c->max_batch) {
        lm_set_error("lm_label_batch_logits: bad arguments (n_frames=%d, max_batch=%d)", n_frames, c ? c->max_batch : -1);
        return LM_ERR_ARG;
    }
    const LmGeom g = c->g;
    float edge = 0.0f;
    int state = 0;
    const bool shape_ok = (g.W & 3) == 0 && ((((uintptr_t)d_logits) & 15) == 0) && (!d_binary || ((((uintptr_t)d_binary) & 3) == 0));
    if (shape_ok && thr >= 1 && thr <= 255 && !getenv("LM_THRESHOLD_FORMULA") && !getenv("LM_LABEL_UNFUSED")) {
        const int rc = lm_threshold_edge(thr, (hipStream_t)stream, &edge, &state);
        if (rc) return rc;
    }
    if (state == 1) {
        LmLabelSrc src = {nullptr, d_logits, edge, invert ? 0xffu : 0u, d_binary};
        return lm_label_batch_src(c, src, n_frames, d_labels, stream);
    }
    if (!d_binary) { lm_set_error("lm_label_batch_logits: this frame shape / threshold needs the two-pass path, which needs d_binary"); return LM_ERR_ARG; }
    const int rc = lm_threshold(d_logits, d_binary, (int64_t)n_frames * g.W * g.H, thr, invert, stream);
    if (rc) return rc;
    return lm_label_batch(c, d_binary, n_frames, d_labels, stream);
}

extern "C" int lm_label_was_fused(LmCtx* c) { return c ? c->last_fused : 0; }

extern "C" int lm_label_counts(LmCtx* c, int32_t* h_counts, void* stream)
{
    if (!c || !h_counts || c->last_batch <= 0) { lm_set_error("lm_label_counts: no labelled batch"); return LM_ERR_STATE; }
    LM_HIP(hipMemcpyAsync(h_counts, c->n_labels, (size_t)c->last_batch * sizeof(int32_t), hipMemcpyDeviceToHost,
                          (hipStream_t)stream));
    LM_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (g_band_stamps && getenv("LM_DEBUG_BAND_STAMPS")) {
        const size_t n = (size_t)c->nbands * c->last_batch * 8;
        unsigned long long* h = (unsigned long long*)malloc(n * sizeof(unsigned long long));
        if (h && hipMemcpy(h, g_band_stamps, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
            FILE* f = fopen(getenv("LM_DEBUG_BAND_STAMPS"), "wb");
            if (f) { fwrite(h, sizeof(unsigned long long), n, f); fclose(f); }
        }
        free(h);
    }
    return LM_OK;
}

extern "C" int lm_cc_stats_batch(LmCtx* c, void* stream)
{
    if (!c || c->last_batch <= 0) { lm_set_error("lm_cc_stats_batch: no labelled batch"); return LM_ERR_STATE; }
    const LmGeom g = c->g;
    hipStream_t st = (hipStream_t)stream;
    const int B = c->last_batch;
    // the labelling launch leaves the arrays initialised (lm_k_apply_labels); a second call for the same batch starts over
    if (!c->stats_fresh)
        hipLaunchKernelGGL(lm_k_stats_init, dim3(32, B), dim3(256), 0, st, c->st_min_y, c->st_max_y, c->st_min_x, c->st_max_x,
                           c->st_count, c->n_labels, g.W, g.H, g.cap);
    c->stats_fresh = 0;
    hipLaunchKernelGGL(lm_k_stats, dim3((g.WW + LM_ST_WORDS - 1) / LM_ST_WORDS, (g.H + LM_ST_ROWS - 1) / LM_ST_ROWS, B), dim3(256), 0,
                       st, c->bits, c->starts, c->prefix, c->rowoff, c->final_label, c->st_min_y, c->st_max_y, c->st_min_x,
                       c->st_max_x, c->st_count, g.WW, g.H, g.cap);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

extern "C" int lm_cc_stats_read(LmCtx* c, int frame, int n, int32_t* h_mins_y, int32_t* h_maxs_y, int32_t* h_mins_x,
                                int32_t* h_maxs_x, int32_t* h_counts, void* stream)
{
    if (!c || frame < 0 || frame >= c->last_batch || n < 0 || n > c->g.cap) {
        lm_set_error("lm_cc_stats_read: bad arguments");
        return LM_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t off = (size_t)frame * c->g.cap, nb = (size_t)n * sizeof(int32_t);
    if (n > 0) {
        if (h_mins_y) LM_HIP(hipMemcpyAsync(h_mins_y, c->st_min_y + off, nb, hipMemcpyDeviceToHost, st));
        if (h_maxs_y) LM_HIP(hipMemcpyAsync(h_maxs_y, c->st_max_y + off, nb, hipMemcpyDeviceToHost, st));
        if (h_mins_x) LM_HIP(hipMemcpyAsync(h_mins_x, c->st_min_x + off, nb, hipMemcpyDeviceToHost, st));
        if (h_maxs_x) LM_HIP(hipMemcpyAsync(h_maxs_x, c->st_max_x + off, nb, hipMemcpyDeviceToHost, st));
        if (h_counts) LM_HIP(hipMemcpyAsync(h_counts, c->st_count + off, nb, hipMemcpyDeviceToHost, st));
    }
    LM_HIP(hipStreamSynchronize(st));
    return LM_OK;
}

// ------------------------------------------------------------------------------------------------
// host-pointer convenience + the reference's own export
// ------------------------------------------------------------------------------------------------
static LmCtx* g_host_ctx = nullptr;

static LmCtx* lm_host_ctx(int width, int height)
{
    if (g_host_ctx && (g_host_ctx->g.W != width || g_host_ctx->g.H != height)) {
        lm_ctx_destroy(g_host_ctx);
        g_host_ctx = nullptr;
    }
    if (!g_host_ctx) g_host_ctx = lm_ctx_create(width, height, 1);
    return g_host_ctx;
}

static int lm_stage(LmCtx* c, size_t px)
{
    if (c->stage_px >= px) return LM_OK;
    if (c->stage_u8) (void)hipFree(c->stage_u8);
    if (c->stage_i32) (void)hipFree(c->stage_i32);
    if (c->stage_f32) (void)hipFree(c->stage_f32);
    c->stage_u8 = nullptr; c->stage_i32 = nullptr; c->stage_f32 = nullptr; c->stage_px = 0;
    if (lm_alloc(&c->stage_u8, px) || lm_alloc(&c->stage_i32, px) || lm_alloc(&c->stage_f32, px)) return LM_ERR_HIP;
    c->stage_px = px;
    return LM_OK;
}

extern "C" int lm_label_host(const uint8_t* h_img, int width, int height, int32_t* h_labels)
{
    if (!h_img || !h_labels || width <= 0 || height <= 0) { lm_set_error("lm_label_host: bad arguments"); return -LM_ERR_ARG; }
    LmCtx* c = lm_host_ctx(width, height);
    if (!c) return -LM_ERR_HIP;
    const size_t px = (size_t)width * height;
    if (lm_stage(c, px)) return -LM_ERR_HIP;
    if (hipMemcpy(c->stage_u8, h_img, px, hipMemcpyHostToDevice) != hipSuccess) { lm_set_error("lm_label_host: H2D failed"); return -LM_ERR_HIP; }
    int rc = lm_label_batch(c, c->stage_u8, 1, c->stage_i32, nullptr);
    if (rc) return -rc;
    int32_t n = 0;
    rc = lm_label_counts(c, &n, nullptr);
    if (rc) return -rc;
    if (hipMemcpy(h_labels, c->stage_i32, px * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) { lm_set_error("lm_label_host: D2H failed"); return -LM_ERR_HIP; }
    return n;
}

// accessmath_lib.c:357-413.  Always returns 0 like the reference (callers ignore the value, labeler.py:166);
// a failure is reported through lm_last_error() and leaves the outputs in their initial state.
extern "C" int CC_AgeBoundaries(int* labels, float* ages, int width, int height, int count_labels, int* out_mins_y,
                                int* out_maxs_y, int* out_mins_x, int* out_maxs_x, int* out_counts, float* output_age)
{
    if (count_labels <= 0 || width <= 0 || height <= 0) return 0;
    LmCtx* c = lm_host_ctx(width, height);
    const size_t px = (size_t)width * height;
    if (!c || lm_stage(c, px)) return 0;
    const int n = count_labels;
    int32_t* d_out = nullptr;     // 9 arrays of n
    if (hipMalloc((void**)&d_out, (size_t)n * 9 * sizeof(int32_t)) != hipSuccess) { lm_set_error("CC_AgeBoundaries: hipMalloc failed"); return 0; }
    int32_t *mny = d_out, *mxy = d_out + n, *mnx = d_out + 2 * (size_t)n, *mxx = d_out + 3 * (size_t)n, *cnt = d_out + 4 * (size_t)n,
            *ageb = d_out + 5 * (size_t)n, *last_px = d_out + 7 * (size_t)n, *last_neg = d_out + 8 * (size_t)n;
    float* agef = (float*)(d_out + 6 * (size_t)n);
    bool ok = hipMemcpy(c->stage_i32, labels, px * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
    if (ok && ages) ok = hipMemcpy(c->stage_f32, ages, px * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        const float* d_ages = ages ? c->stage_f32 : (const float*)nullptr;
        hipLaunchKernelGGL(lm_k_ab_init, dim3(lm_blocks(n, 256)), dim3(256), 0, (hipStream_t)0, mny, mxy, mnx, mxx, cnt, ageb, last_px, last_neg,
                           width, height, n);
        hipLaunchKernelGGL(lm_k_ab_scan, dim3(lm_blocks((long long)px, 256)), dim3(256), 0, (hipStream_t)0, c->stage_i32, d_ages, width, height, n,
                           mny, mxy, mnx, mxx, cnt, last_px, last_neg);
        hipLaunchKernelGGL(lm_k_ab_age, dim3(lm_blocks((long long)px, 256)), dim3(256), 0, (hipStream_t)0, c->stage_i32, d_ages, width, height, n,
                           last_neg, ageb);
        hipLaunchKernelGGL(lm_k_ab_finish, dim3(lm_blocks(n, 256)), dim3(256), 0, (hipStream_t)0, ageb, last_px, last_neg, d_ages, agef, n);
        const size_t nb = (size_t)n * sizeof(int32_t);
        ok = hipMemcpy(out_mins_y, mny, nb, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(out_maxs_y, mxy, nb, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(out_mins_x, mnx, nb, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(out_maxs_x, mxx, nb, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(out_counts, cnt, nb, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(output_age, agef, nb, hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (!ok) lm_set_error("CC_AgeBoundaries: device copy/launch failed");
    (void)hipFree(d_out);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// stream
// ------------------------------------------------------------------------------------------------
extern "C" void lm_stream_destroy(LmStream* s)
{
    if (!s) return;
    void* ptrs[] = {s->cc, s->assign, s->frame_cc_off, s->crop, s->chash, s->active_cc, s->active_box, s->active_last, s->active, s->counters, s->best,
                    s->batch_cc_base, s->batch_word_base};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (s->mb) {
        LmMatchBatch* m = s->mb;
        void* mp[] = {m->nt_cnt, m->nt_src, m->tlast, m->tcur[0], m->tcur[1], m->ftile_all, m->nt_foff, m->nt_list, m->cls, m->troot, m->rootpos, m->ftile, m->s_prefix, m->tcount[0], m->tcount[1], m->toff[0], m->toff[1], m->pairs[0], m->pairs[1], m->pair_u[0],
                      m->pair_u[1], m->sidx, m->s_list, m->s_box, m->newpos, m->n_src, m->ttab, m->tkey, m->twin, m->big[0], m->big[1], m->n_big};
        for (void* p : mp)
            if (p) (void)hipFree(p);
        delete m;
    }
    if (s->twin_stream) (void)hipStreamDestroy((hipStream_t)s->twin_stream);
    for (int i = 0; i < s->n_run_events; i++) (void)hipEventDestroy(s->run_events[i]);
    free(s->run_events);
    if (s->rd_scratch) (void)hipFree(s->rd_scratch);
    if (s->garena) (void)hipFree(s->garena);
    if (s->gpin) (void)hipHostFree(s->gpin);
    delete s;
}

extern "C" int lm_stream_set_min_pixels(LmStream* s, int min_pixels)
{
    if (!s || min_pixels < 0) { lm_set_error("lm_stream_set_min_pixels: bad arguments"); return LM_ERR_ARG; }
    s->min_pixels = min_pixels;
    return LM_OK;
}

extern "C" int lm_stream_reset(LmStream* s, void* stream)
{
    if (!s) { lm_set_error("lm_stream_reset: null stream"); return LM_ERR_ARG; }
    LM_HIP(hipMemsetAsync(s->counters, 0, sizeof(LmCounters), (hipStream_t)stream));
    LM_HIP(hipMemsetAsync(s->best, 0xff, (size_t)s->ctx->g.cap * sizeof(unsigned long long), (hipStream_t)stream));
    LM_HIP(hipMemsetAsync(s->frame_cc_off, 0, sizeof(long long), (hipStream_t)stream));
    LM_HIP(hipMemsetAsync(s->chash, 0, (size_t)s->cap_cc * sizeof(uint32_t), (hipStream_t)stream));     // lm_k_emit adds into it
    // the twin table is empty between batches (lm_k_mb_twin_final clears what its batch inserted); a batch cut short by an error leaves entries behind
    if (s->mb) LM_HIP(hipMemsetAsync(s->mb->ttab, 0xff, (size_t)LM_MB_TTAB * sizeof(unsigned long long), (hipStream_t)stream));
    s->frames_pushed = 0;
    s->frames_matched = 0;
    s->tempo_B = 0;
    return LM_OK;
}

extern "C" LmStream* lm_stream_create(LmCtx* ctx, int max_frames, int64_t max_ccs, int64_t max_crop_words, int max_uniques,
                                      double min_recall, double min_precision, int max_gap, int min_pixels)
{
    if (!ctx || max_frames <= 0 || max_ccs <= 0 || max_crop_words <= 0 || max_uniques <= 0) {
        lm_set_error("lm_stream_create: bad arguments");
        return nullptr;
    }
    LmStream* s = new LmStream();
    memset(s, 0, sizeof(*s));
    s->ctx = ctx;
    s->cap_frames = max_frames;
    s->cap_cc = max_ccs;
    s->cap_words = (unsigned long long)max_crop_words;
    s->cap_uniq = max_uniques;
    s->min_recall = min_recall;
    s->min_precision = min_precision;
    s->max_gap = max_gap;
    s->min_pixels = min_pixels;
    int rc = LM_OK;
    rc |= lm_alloc(&s->cc, (size_t)max_ccs);
    rc |= lm_alloc(&s->assign, (size_t)max_ccs);
    rc |= lm_alloc(&s->frame_cc_off, (size_t)max_frames + 1);
    rc |= lm_alloc(&s->crop, (size_t)max_crop_words);
    rc |= lm_alloc(&s->chash, (size_t)max_ccs);
    rc |= lm_alloc(&s->active_cc, (size_t)max_uniques);
    rc |= lm_alloc(&s->active_box, (size_t)max_uniques);
    rc |= lm_alloc(&s->active_last, (size_t)max_uniques);
    rc |= lm_alloc(&s->active, (size_t)max_uniques);
    rc |= lm_alloc(&s->best, (size_t)ctx->g.cap);
    rc |= lm_alloc(&s->counters, (size_t)1);
    rc |= lm_alloc(&s->batch_cc_base, (size_t)ctx->max_batch);
    rc |= lm_alloc(&s->batch_word_base, (size_t)ctx->max_batch);
    {
        LmMatchBatch* m = new LmMatchBatch();
        memset(m, 0, sizeof(*m));
        s->mb = m;
        m->cap_tiles = (int)(max_ccs / LM_MB_TILE) + max_frames + 2;
        long long cp = 16 * (long long)max_ccs;
        if (cp < (1 << 16)) cp = 1 << 16;
        if (cp > (1 << 23)) cp = 1 << 23;
        m->cap_pairs = (uint32_t)cp;
        rc |= lm_alloc(&m->ftile, (size_t)max_frames + 2);
        rc |= lm_alloc(&m->s_prefix, (size_t)max_frames + 2);
        rc |= lm_alloc(&m->ftile_all, (size_t)max_frames + 2);
        rc |= lm_alloc(&m->nt_foff, (size_t)max_frames + 2);
        rc |= lm_alloc(&m->nt_list, (size_t)max_ccs);
        rc |= lm_alloc(&m->cls, (size_t)max_ccs);
        rc |= lm_alloc(&m->troot, (size_t)max_ccs);
        rc |= lm_alloc(&m->rootpos, (size_t)max_ccs);
        rc |= lm_alloc(&m->nt_cnt, (size_t)LM_MB_MAX_FRAMES);
        rc |= lm_alloc(&m->nt_src, (size_t)max_ccs);
        rc |= lm_alloc(&m->tlast, (size_t)max_ccs);
        for (int k = 0; k < 2; k++) {
            rc |= lm_alloc(&m->tcount[k], (size_t)m->cap_tiles + 1);
            rc |= lm_alloc(&m->toff[k], (size_t)m->cap_tiles + 1);
            rc |= lm_alloc(&m->tcur[k], ((size_t)m->cap_tiles + 1) * LM_MB_JY);
            rc |= lm_alloc(&m->pairs[k], (size_t)m->cap_pairs);
            rc |= lm_alloc(&m->pair_u[k], (size_t)m->cap_pairs);
        }
        rc |= lm_alloc(&m->sidx, (size_t)max_ccs);
        rc |= lm_alloc(&m->s_list, (size_t)max_ccs);
        rc |= lm_alloc(&m->s_box, (size_t)max_ccs);
        rc |= lm_alloc(&m->newpos, (size_t)max_ccs);
        rc |= lm_alloc(&m->n_src, (size_t)1);
        rc |= lm_alloc(&m->ttab, (size_t)LM_MB_TTAB);
        rc |= lm_alloc(&m->tkey, (size_t)max_ccs);
        rc |= lm_alloc(&m->twin, (size_t)max_ccs);
        m->cap_big = 1u << 16;
        rc |= lm_alloc(&m->big[0], (size_t)m->cap_big);
        rc |= lm_alloc(&m->big[1], (size_t)m->cap_big);
        rc |= lm_alloc(&m->n_big, (size_t)2);
#if !LM_HIP_EMULATED
        if (rc == LM_OK &&
            hipFuncSetAttribute((const void*)lm_k_mb_resolve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LM_MB_RESOLVE_SMEM) != hipSuccess)
            rc = LM_ERR_HIP;
#endif
        const char* e = getenv("LM_MATCH_PER_FRAME");
        s->match_per_frame = (e && atoi(e) == 1) ? 1 : 0;
    }
    if (rc == LM_OK && hipMemset(s->mb->twin, 0, (size_t)max_ccs) != hipSuccess) rc = LM_ERR_HIP;   // stays 0 when twin detection is off
    if (rc == LM_OK) rc = lm_stream_reset(s, nullptr);
    if (rc == LM_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = LM_ERR_HIP;
    if (rc != LM_OK) {
        lm_stream_destroy(s);
        return nullptr;
    }
    return s;
}

static void lm_launch_match(LmStream* s, int f, hipStream_t st)
{
    hipLaunchKernelGGL(lm_k_match, LM_HIP_EMULATED ? dim3(2, 2) : dim3(16, 64), dim3(256), 0, st, s->cc, s->crop, s->frame_cc_off, f,
                       s->active_box, s->active_last, s->active_cc, s->counters, s->best, s->min_recall, s->min_precision, s->max_gap);
    // compact the active list every 16 frames (purely an optimisation: retirement is evaluated lazily)
    hipLaunchKernelGGL(lm_k_update, dim3(1), dim3(1024), 0, st, s->cc, s->frame_cc_off, f, s->active, s->active_cc, s->active_box,
                       s->active_last, s->counters, s->assign, s->best, s->max_gap, s->cap_uniq, (f & 15) == 15 ? 1 : 0);
}

// Matches frames [f0, f0 + n) (all emitted already) in chunks of at most LM_MB_MAX_FRAMES frames.
// ev_pre (optional): recorded on `st` in front of the last chunk's replay kernel, i.e. when the wide kernels of the matching
// (twin detection, joins, pair evaluation) have been issued and only the single-workgroup replay, finish and tempo_count remain
// defer_tempo: the tempo_count kernel of the LAST chunk (a wide launch that only adds to a counter) is not launched here but in front
// of the next call's kernels, or by lm_flush_tempo: in the gated schedule it would run beside the next batch's labelling launches,
// deferred it runs beside that batch's record emission instead.  It reads the active list as the replay left it and the tile
// table of its batch: both are untouched until the next batch's lm_k_mb_nt, which the same queue runs after it.
static void lm_flush_tempo(LmStream* s, hipStream_t st)
{
    if (s->tempo_B <= 0) return;
    const LmMatchBatch mb = *s->mb;
    hipLaunchKernelGGL(lm_k_mb_tempo, dim3(LM_HIP_EMULATED ? 2 : 1024), dim3(256), 0, st, s->cc, s->frame_cc_off, s->tempo_f0, s->tempo_B, s->active_box,
                       s->active_cc, s->active_last, s->counters, mb, s->max_gap, s->active, s->assign);
    s->tempo_B = 0;
}

// st_twin / ev_twin / ev_nt (lm_stream_run_logits, batches of one chunk): the twin-detection kernels of the batch go to st_twin -- the caller has
// made that queue wait for the batch's records -- and `st` picks up behind them (ev_twin); ev_nt is recorded on `st` behind lm_k_mb_nt, the last
// reader of the per-batch non-twin counters the NEXT batch's twin kernels reset (the caller makes st_twin wait for it).  Twin detection needs a
// batch's records only, so it runs while `st` is still replaying the batch before: ~56 us per batch off the matching queue's chain.
static void lm_launch_match_frames(LmStream* s, int f0, int n, hipStream_t st, hipEvent_t ev_pre = nullptr, bool defer_tempo = false,
                                   hipStream_t st_twin = nullptr, hipEvent_t ev_twin = nullptr, hipEvent_t ev_nt = nullptr)
{
    if (!s->match_per_frame) lm_flush_tempo(s, st);
    if (n > LM_MB_MAX_FRAMES) st_twin = nullptr;
    if (s->match_per_frame) {
        for (int i = 0; i < n; i++) lm_launch_match(s, f0 + i, st);
        return;
    }
    const LmMatchBatch mb = *s->mb;
    const dim3 gj(LM_HIP_EMULATED ? 2 : 128, LM_HIP_EMULATED ? 2 : LM_MB_JY), gt(LM_HIP_EMULATED ? 2 : 1024), ge(LM_HIP_EMULATED ? 2 : 2048), gb(LM_HIP_EMULATED ? 2 : 1024);
    for (int done = 0; done < n;) {
        const int B = (n - done < LM_MB_MAX_FRAMES) ? n - done : LM_MB_MAX_FRAMES;
        const int f = f0 + done;
        s->last_match_frames = B;
        const int twins = (s->min_recall <= 1.0 && s->min_precision <= 1.0) ? 1 : 0;     // the twin rule needs "identical crops are accepted"; otherwise twin[] stays 0
        if (twins) {
            const dim3 gc(LM_HIP_EMULATED ? 2 : 256);
            const hipStream_t tq = st_twin ? st_twin : st;
            hipLaunchKernelGGL(lm_k_mb_twin_insert, gc, dim3(256), 0, tq, s->cc, s->chash, s->frame_cc_off, f, B, s->counters, mb);
            hipLaunchKernelGGL(lm_k_mb_twin_probe, gc, dim3(256), 0, tq, s->cc, s->frame_cc_off, f, B, s->counters, mb, s->max_gap);
            hipLaunchKernelGGL(lm_k_mb_twin_cmp, ge, dim3(256), 0, tq, s->cc, s->crop, s->frame_cc_off, f, B, s->counters, mb);
            hipLaunchKernelGGL(lm_k_mb_twin_final, gc, dim3(256), 0, tq, s->cc, s->frame_cc_off, f, B, s->counters, mb);
            if (st_twin) { (void)hipEventRecord(ev_twin, st_twin); (void)hipStreamWaitEvent(st, ev_twin, 0); }
        }
        hipLaunchKernelGGL(lm_k_mb_nt, dim3(B + 1), dim3(1024), 0, st, s->frame_cc_off, f, B, s->active, s->active_cc, s->active_box, s->active_last,
                           s->counters, mb, s->max_gap, twins);
        if (twins && st_twin) (void)hipEventRecord(ev_nt, st);
        hipLaunchKernelGGL((lm_k_mb_join<0, 0>), gj, dim3(256), 0, st, s->cc, s->frame_cc_off, f, B, s->active_box, s->active_cc, s->counters, mb);
        hipLaunchKernelGGL((lm_k_mb_join<0, 1>), gj, dim3(256), 0, st, s->cc, s->frame_cc_off, f, B, s->active_box, s->active_cc, s->counters, mb);
        hipLaunchKernelGGL((lm_k_mb_eval<0>), ge, dim3(256), 0, st, s->cc, s->crop, s->frame_cc_off, f, B, s->active_last, s->counters, mb,
                           s->min_recall, s->min_precision, s->max_gap);
        hipLaunchKernelGGL((lm_k_mb_eval_big<0>), gb, dim3(256), 0, st, s->cc, s->crop, s->active_last, s->counters, mb, s->min_recall, s->min_precision,
                           s->max_gap);
        hipLaunchKernelGGL(lm_k_mb_sources, dim3(1), dim3(1024), 0, st, s->cc, s->frame_cc_off, f, B, s->counters, mb);
        hipLaunchKernelGGL((lm_k_mb_join<1, 0>), gj, dim3(256), 0, st, s->cc, s->frame_cc_off, f, B, s->active_box, s->active_cc, s->counters, mb);
        hipLaunchKernelGGL((lm_k_mb_join<1, 1>), gj, dim3(256), 0, st, s->cc, s->frame_cc_off, f, B, s->active_box, s->active_cc, s->counters, mb);
        hipLaunchKernelGGL((lm_k_mb_eval<1>), ge, dim3(256), 0, st, s->cc, s->crop, s->frame_cc_off, f, B, s->active_last, s->counters, mb,
                           s->min_recall, s->min_precision, s->max_gap);
        hipLaunchKernelGGL((lm_k_mb_eval_big<1>), gb, dim3(256), 0, st, s->cc, s->crop, s->active_last, s->counters, mb, s->min_recall, s->min_precision,
                           s->max_gap);
        if (ev_pre && done + B >= n) (void)hipEventRecord(ev_pre, st);
        hipLaunchKernelGGL(lm_k_mb_resolve, dim3(1), dim3(LM_MB_RT), LM_MB_RESOLVE_SMEM, st, s->cc, s->frame_cc_off, f, B, s->active, s->active_cc,
                           s->active_box, s->active_last, s->counters, s->assign, mb, s->max_gap, s->cap_uniq);
        if (defer_tempo && done + B >= n) { s->tempo_f0 = f; s->tempo_B = B; }
        else
            hipLaunchKernelGGL(lm_k_mb_tempo, gt, dim3(256), 0, st, s->cc, s->frame_cc_off, f, B, s->active_box, s->active_cc, s->active_last,
                               s->counters, mb, s->max_gap, s->active, s->assign);
        done += B;
    }
}

// statistics, kept-CC selection, record + crop emission for the B frames last labelled in the stream's context
static int lm_stream_emit_batch(LmStream* s, int B, void* stream)
{
    LmCtx* c = s->ctx;
    const LmGeom g = c->g;
    hipStream_t st = (hipStream_t)stream;
    int rc = lm_cc_stats_batch(c, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(lm_k_select, dim3(B), dim3(1024), 0, st, c->st_min_y, c->st_max_y, c->st_min_x, c->st_max_x,
                       c->st_count, c->n_labels, c->kept_label, c->kept_cropoff, c->frame_kept, c->frame_cropwords, g.cap,
                       s->min_pixels);
    hipLaunchKernelGGL(lm_k_batch_offsets, dim3(1), dim3(1024), 0, st, c->frame_kept, c->frame_cropwords, B, s->counters,
                       s->frame_cc_off, s->batch_cc_base, s->batch_word_base, s->cap_cc, s->cap_words, s->cap_frames);
    hipLaunchKernelGGL(lm_k_emit, dim3(LM_HIP_EMULATED ? 2 : LM_EMIT_GRID, B), dim3(256), 0, st, c->bits, c->starts, c->prefix, c->rowoff, c->final_label,
                       c->st_min_y, c->st_max_y, c->st_min_x, c->st_max_x, c->st_count, c->kept_label, c->kept_cropoff,
                       c->frame_kept, c->frame_cropwords, s->batch_cc_base, s->batch_word_base, s->cc, s->crop, s->chash, s->frames_pushed, g.WW,
                       g.H, g.cap);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// The second half of lm_stream_push_records for callers that label a batch themselves (lm_label_batch on the stream's context)
// and want to schedule the two halves separately.
extern "C" int lm_stream_push_labelled(LmStream* s, int n_frames, void* stream)
{
    if (!s || n_frames <= 0 || n_frames != s->ctx->last_batch) {
        lm_set_error("lm_stream_push_labelled: %d frames, the context's last labelled batch has %d", n_frames, s ? s->ctx->last_batch : 0);
        return LM_ERR_ARG;
    }
    if (s->frames_pushed + n_frames > s->cap_frames) {
        lm_set_error("lm_stream_push_labelled: stream holds %d frames, capacity %d", s->frames_pushed, s->cap_frames);
        return LM_ERR_CAPACITY;
    }
    const int rc = lm_stream_emit_batch(s, n_frames, stream);
    if (rc) return rc;
    s->frames_pushed += n_frames;
    return LM_OK;
}

static int lm_stream_push_impl(LmStream* s, const uint8_t* d_binary, int n_frames, int32_t* d_labels, int do_match, void* stream)
{
    if (!s || !d_binary || n_frames <= 0) { lm_set_error("lm_stream_push: bad arguments"); return LM_ERR_ARG; }
    LmCtx* c = s->ctx;
    const LmGeom g = c->g;
    hipStream_t st = (hipStream_t)stream;
    if (s->frames_pushed + n_frames > s->cap_frames) {
        lm_set_error("lm_stream_push: stream holds %d frames, capacity %d", s->frames_pushed, s->cap_frames);
        return LM_ERR_CAPACITY;
    }
    if (do_match && s->frames_matched != s->frames_pushed) {
        lm_set_error("lm_stream_push: %d frames pushed without matching; call lm_stream_match first", s->frames_pushed - s->frames_matched);
        return LM_ERR_STATE;
    }
    const size_t px = (size_t)g.W * g.H;
    for (int done = 0; done < n_frames;) {
        const int B = (n_frames - done < c->max_batch) ? n_frames - done : c->max_batch;
        int rc = lm_label_batch(c, d_binary + (size_t)done * px, B, d_labels ? d_labels + (size_t)done * px : nullptr, stream);
        if (rc) return rc;
        rc = lm_stream_emit_batch(s, B, stream);
        if (rc) return rc;
        if (do_match) {
            lm_launch_match_frames(s, s->frames_pushed, B, st);
            s->frames_matched += B;
        }
        LM_HIP(hipGetLastError());
        s->frames_pushed += B;
        done += B;
    }
    return LM_OK;
}

extern "C" int lm_stream_push(LmStream* s, const uint8_t* d_binary, int n_frames, int32_t* d_labels, void* stream)
{
    return lm_stream_push_impl(s, d_binary, n_frames, d_labels, 1, stream);
}

extern "C" int lm_stream_push_records(LmStream* s, const uint8_t* d_binary, int n_frames, int32_t* d_labels, void* stream)
{
    return lm_stream_push_impl(s, d_binary, n_frames, d_labels, 0, stream);
}

// Steps 01-02 of a whole resident stream in one call: per batch threshold+invert -> label -> records on `stream_wide`, temporal
// matching on `stream_match` behind an event per batch (the two must be different streams for the phases to overlap).  The
// launch loop runs here instead of in the caller's interpreter: a 10,000-frame stream is ~5,000 kernel launches, and a Python
// loop around them (plus a second Python thread driving step 03 of the previous stream) was the bottleneck of the pipeline.
// d_binary: scratch for `batch` frames; d_labels: label image of one batch, or NULL.  Asynchronous: when the call returns
// everything is enqueued; the work of the stream is complete when `stream_match` has drained.
static int lm_run_events(LmStream* s, int want)
{
    if (want <= s->n_run_events) return LM_OK;
    want += 192;
    hipEvent_t* ev = (hipEvent_t*)realloc(s->run_events, (size_t)want * sizeof(hipEvent_t));
    if (!ev) { lm_set_error("lm_stream_run_logits: out of memory"); return LM_ERR_HIP; }
    s->run_events = ev;
    for (; s->n_run_events < want; s->n_run_events++) LM_HIP(hipEventCreateWithFlags(&s->run_events[s->n_run_events], hipEventDisableTiming));
    return LM_OK;
}

extern "C" int lm_stream_run_logits(LmStream* s, const float* d_logits, int n_frames, int batch, uint8_t* d_binary, int32_t* d_labels, int thr,
                                    int do_match, int schedule, void* stream_wide, void* stream_match)
{
    if (!s || !d_logits || n_frames < 0 || batch <= 0 || batch > s->ctx->max_batch || schedule < 0 || schedule > 1) {
        lm_set_error("lm_stream_run_logits: bad arguments (batch %d, context batch %d, schedule %d)", batch, s ? s->ctx->max_batch : 0, schedule);
        return LM_ERR_ARG;
    }
    if (do_match && s->frames_matched != s->frames_pushed) {
        lm_set_error("lm_stream_run_logits: %d frames pushed without matching; call lm_stream_match first", s->frames_pushed - s->frames_matched);
        return LM_ERR_STATE;
    }
    hipStream_t sw = (hipStream_t)stream_wide, sm = (hipStream_t)stream_match;
    const bool two = do_match && sm != sw;
    const bool gated = two && schedule == 1 && batch <= LM_MB_MAX_FRAMES && !s->match_per_frame;
    const size_t px = (size_t)s->ctx->g.W * s->ctx->g.H;
    const int nb = (n_frames + batch - 1) / batch;
    if (two) { const int rc = lm_run_events(s, 3 * nb + 2 * nb); if (rc) return rc; }
    // events of batch k: 3k = labelled, 3k + 1 = records appended, 3k + 2 = its matching has reached the replay kernel;
    // behind those, 3 nb + 2k = its twin detection is done, 3 nb + 2k + 1 = its lm_k_mb_nt is done
    hipEvent_t* ev = s->run_events;
    hipEvent_t* ev2 = s->run_events + 3 * nb;
    // LM_TWIN_QUEUE=1 (free schedule): twin detection of batch k on a queue of its own, beside the replay of batch k - 1.  Measured
    // (profiles/r03_s2_operating_points.txt): 62-63 k frames/s against 68-70 k without -- one more queue of wide kernels beside the labelling
    // launches costs more than the ~56 us per batch it takes off the matching queue's chain.  Off by default.
    static const int twin_queue = [] { const char* e = getenv("LM_TWIN_QUEUE"); return e ? atoi(e) : 0; }();
    hipStream_t st_twin = nullptr;
    if (two && twin_queue && !s->match_per_frame && batch <= LM_MB_MAX_FRAMES) {
        if (!s->twin_stream) LM_HIP(hipStreamCreateWithFlags((hipStream_t*)&s->twin_stream, hipStreamNonBlocking));
        st_twin = (hipStream_t)s->twin_stream;
    }
    int prev_n = 0;
    // LM_RUN_AHEAD=R (two-stream forms): the launch loop stays at most R batches ahead of the GPU (it waits for the records of batch
    // k - R before it enqueues batch k).  A stream's ~4,500 launches enqueued at once fill the hardware queues for >100 ms; kernels
    // that another host thread submits meanwhile (step 03 of the previous stream) on a stream that shares a hardware queue with this
    // one's wait behind all of them.  0 (default) = no limit.
    static const int run_ahead = [] { const char* e = getenv("LM_RUN_AHEAD"); return (e && atoi(e) > 0) ? atoi(e) : 0; }();
    for (int k = 0, f0 = 0; f0 < n_frames; f0 += batch, k++) {
        const int n = (n_frames - f0 < batch) ? n_frames - f0 : batch;
        if (two && run_ahead > 0 && k >= run_ahead) LM_HIP(hipEventSynchronize(ev[3 * (k - run_ahead) + 1]));
        if (gated && k >= 2) LM_HIP(hipStreamWaitEvent(sw, ev[3 * (k - 2) + 2], 0));       // the wide kernels of matching k-2 are through
        int rc = lm_label_batch_logits(s->ctx, d_logits + (size_t)f0 * px, n, thr, 1, d_binary, d_labels, stream_wide);
        if (rc) return rc;
        if (gated) LM_HIP(hipEventRecord(ev[3 * k], sw));
        rc = lm_stream_push_labelled(s, n, stream_wide);
        if (rc) return rc;
        if (!do_match) continue;
        if (!two) {
            rc = lm_stream_match(s, n, stream_wide);
            if (rc) return rc;
        } else if (!gated) {
            LM_HIP(hipEventRecord(ev[3 * k + 1], sw));
            LM_HIP(hipStreamWaitEvent(sm, ev[3 * k + 1], 0));
            if (st_twin) {
                LM_HIP(hipStreamWaitEvent(st_twin, ev[3 * k + 1], 0));
                if (k >= 1) LM_HIP(hipStreamWaitEvent(st_twin, ev2[2 * (k - 1) + 1], 0));
                lm_launch_match_frames(s, s->frames_matched, n, sm, nullptr, false, st_twin, ev2[2 * k], ev2[2 * k + 1]);
                s->frames_matched += n;
            } else {
                rc = lm_stream_match(s, n, stream_match);
                if (rc) return rc;
            }
        } else {
            LM_HIP(hipEventRecord(ev[3 * k + 1], sw));
            if (k >= 1) {               // matching of batch k-1 starts when batch k has been labelled (which implies its records are in)
                LM_HIP(hipStreamWaitEvent(sm, ev[3 * k], 0));
                lm_launch_match_frames(s, s->frames_matched, prev_n, sm, ev[3 * (k - 1) + 2], true);
                s->frames_matched += prev_n;
            }
        }
        prev_n = n;
    }
    if (gated && nb > 0) {
        LM_HIP(hipStreamWaitEvent(sm, ev[3 * (nb - 1) + 1], 0));
        lm_launch_match_frames(s, s->frames_matched, prev_n, sm, ev[3 * (nb - 1) + 2]);
        s->frames_matched += prev_n;
    }
    lm_flush_tempo(s, sm);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

extern "C" int lm_stream_match(LmStream* s, int n_frames, void* stream)
{
    if (!s || n_frames < 0 || s->frames_matched + n_frames > s->frames_pushed) {
        lm_set_error("lm_stream_match: %d frames requested, %d pushed, %d matched", n_frames, s ? s->frames_pushed : 0, s ? s->frames_matched : 0);
        return LM_ERR_ARG;
    }
    lm_launch_match_frames(s, s->frames_matched, n_frames, (hipStream_t)stream);
    s->frames_matched += n_frames;
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// Diagnostic: sizes of the LAST matching batch {sources, tiles, pairs vs earlier uniques, pairs vs in-batch sources, frames}.
extern "C" int lm_stream_match_stats(LmStream* s, int64_t* out5, void* stream)
{
    if (!s || !out5 || !s->mb) { lm_set_error("lm_stream_match_stats: bad arguments"); return LM_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const LmMatchBatch* m = s->mb;
    const int B = s->last_match_frames;
    int32_t nsrc = 0, nt = 0;
    uint32_t ta = 0, tb = 0;
    if (B > 0) {
        LM_HIP(hipMemcpyAsync(&nsrc, m->n_src, 4, hipMemcpyDeviceToHost, st));
        LM_HIP(hipMemcpyAsync(&nt, m->ftile + B, 4, hipMemcpyDeviceToHost, st));
        LM_HIP(hipStreamSynchronize(st));
        if (nt > 0) {
            LM_HIP(hipMemcpyAsync(&ta, m->toff[0] + nt, 4, hipMemcpyDeviceToHost, st));
            LM_HIP(hipMemcpyAsync(&tb, m->toff[1] + nt, 4, hipMemcpyDeviceToHost, st));
            LM_HIP(hipStreamSynchronize(st));
        }
    }
    out5[0] = nsrc; out5[1] = nt; out5[2] = ta; out5[3] = tb; out5[4] = B;
    return LM_OK;
}

// drops retired entries from the active list so that it equals the reference's cc_active after the last frame
__global__ void __launch_bounds__(1024) lm_k_compact_active(int32_t* __restrict__ active, int32_t* __restrict__ active_cc,
                                                            unsigned long long* __restrict__ active_box, int32_t* __restrict__ active_last,
                                                            LmCounters* __restrict__ cnt, int max_gap)
{
    if (cnt->error) return;
    const int f = cnt->n_matched;       // index of the next frame
    const int nA = cnt->n_active;
    if (f <= 1) return;
    unsigned kept = 0;
    for (int base = 0; base < nA; base += 1024) {
        int i = base + (int)threadIdx.x;
        int32_t u = 0, uc = 0, ul = 0;
        unsigned long long ub = 0;
        unsigned keep = 0;
        if (i < nA) {
            u = active[i]; uc = active_cc[i]; ub = active_box[i]; ul = active_last[i];
            keep = ((f - 1) - ul < max_gap) ? 1u : 0u;
        }
        unsigned tot;
        unsigned ex = lm_block_excl_scan<1024>(keep, &tot);
        if (keep) { active[kept + ex] = u; active_cc[kept + ex] = uc; active_box[kept + ex] = ub; active_last[kept + ex] = ul; }
        kept += tot;
    }
    if (threadIdx.x == 0) cnt->n_active = (int)kept;
}

extern "C" int lm_stream_counters(LmStream* s, int64_t* out, void* stream)
{
    if (!s || !out) { lm_set_error("lm_stream_counters: bad arguments"); return LM_ERR_ARG; }
    hipLaunchKernelGGL(lm_k_compact_active, dim3(1), dim3(1024), 0, (hipStream_t)stream, s->active, s->active_cc, s->active_box,
                       s->active_last, s->counters, s->max_gap);
    LmCounters h;
    LM_HIP(hipMemcpyAsync(&h, s->counters, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    LM_HIP(hipStreamSynchronize((hipStream_t)stream));
    out[0] = h.n_frames; out[1] = h.n_cc; out[2] = (int64_t)h.n_words; out[3] = h.n_uniq; out[4] = h.n_active;
    out[5] = (int64_t)h.tempo_count; out[6] = h.error;
    if (h.error) {
        lm_set_error("stream capacity exceeded on device (max_ccs=%lld max_crop_words=%llu max_uniques=%d max_frames=%d)",
                     s->cap_cc, s->cap_words, s->cap_uniq, s->cap_frames);
        return h.error;
    }
    return LM_OK;
}

// unpack host records into the device layout (inverse of lm_k_pack_records)
__global__ void __launch_bounds__(256) lm_k_unpack_records(const int32_t* __restrict__ in8, const long long* __restrict__ crop_off,
                                                           long long n, LmCcRec* __restrict__ cc, int32_t* __restrict__ assign)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int32_t* o = in8 + i * 8;
        LmCcRec r;
        r.cc_id = o[0]; r.min_x = (int16_t)o[1]; r.max_x = (int16_t)o[2]; r.min_y = (int16_t)o[3]; r.max_y = (int16_t)o[4];
        r.size = o[5]; r.frame = o[6]; r.pad = 0; r.crop_off = (unsigned long long)crop_off[i];
        cc[i] = r;
        assign[i] = o[7];
    }
}

__global__ void __launch_bounds__(256) lm_k_import_active(const LmCcRec* __restrict__ cc, const int32_t* __restrict__ active_cc, int n,
                                                          unsigned long long* __restrict__ active_box)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) active_box[i] = lm_pack_box(cc[active_cc[i]]);
}

// Rebuilds a stream on the device from host arrays in lm_stream_read's format (the step-02 -> step-03 hand-off through a
// pickle, console_ui_process.py:150-186).  With the active list (unique index, its first-seen CC, last frame matched; may be
// NULL / 0) the stream can also keep receiving frames.
extern "C" int lm_stream_import(LmStream* s, const int32_t* h_rec, const int64_t* h_frame_off, const int64_t* h_crop_off,
                                const uint32_t* h_crop, int n_frames, int64_t n_cc, int64_t n_crop_words, int n_unique,
                                int64_t tempo_count, const int32_t* h_active, const int32_t* h_active_cc,
                                const int32_t* h_active_last, int n_active, int n_matched, void* stream)
{
    if (!s || n_frames < 0 || n_cc < 0 || n_crop_words < 0 || (n_cc > 0 && (!h_rec || !h_crop_off || !h_crop)) || !h_frame_off ||
        n_matched < 0 || n_matched > n_frames) {
        lm_set_error("lm_stream_import: bad arguments");
        return LM_ERR_ARG;
    }
    if (n_frames > s->cap_frames || n_cc > s->cap_cc || (unsigned long long)n_crop_words > s->cap_words || n_unique > s->cap_uniq) {
        lm_set_error("lm_stream_import: stream too small (frames %d/%d, ccs %lld/%lld, crop words %lld/%llu, uniques %d/%d)", n_frames,
                     s->cap_frames, (long long)n_cc, s->cap_cc, (long long)n_crop_words, s->cap_words, n_unique, s->cap_uniq);
        return LM_ERR_CAPACITY;
    }
    hipStream_t st = (hipStream_t)stream;
    int rc = lm_stream_reset(s, stream);
    if (rc) return rc;
    if (n_cc > 0) {
        int32_t* d_rec = nullptr;
        long long* d_off = nullptr;
        LM_HIP(hipMalloc((void**)&d_rec, (size_t)n_cc * 8 * sizeof(int32_t)));
        if (hipMalloc((void**)&d_off, (size_t)n_cc * sizeof(long long)) != hipSuccess) { (void)hipFree(d_rec); lm_set_error("lm_stream_import: hipMalloc failed"); return LM_ERR_HIP; }
        hipError_t e = hipMemcpyAsync(d_rec, h_rec, (size_t)n_cc * 8 * sizeof(int32_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_off, h_crop_off, (size_t)n_cc * sizeof(long long), hipMemcpyHostToDevice, st);
        if (e == hipSuccess && n_crop_words > 0) e = hipMemcpyAsync(s->crop, h_crop, (size_t)n_crop_words * sizeof(uint32_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(lm_k_unpack_records, dim3(lm_blocks(n_cc, 256)), dim3(256), 0, st, d_rec, d_off, (long long)n_cc, s->cc, s->assign);
            hipLaunchKernelGGL(lm_k_crop_hash, dim3(lm_blocks(n_crop_words / 64 + 1, 4, 2048)), dim3(256), 0, st, s->cc, s->crop, 0ll, (long long)n_cc,
                               (unsigned long long)n_crop_words, s->chash);
            e = hipStreamSynchronize(st);
        }
        (void)hipFree(d_rec);
        (void)hipFree(d_off);
        if (e != hipSuccess) { lm_set_error("lm_stream_import: copy failed"); return LM_ERR_HIP; }
    }
    LM_HIP(hipMemcpyAsync(s->frame_cc_off, h_frame_off, (size_t)(n_frames + 1) * sizeof(long long), hipMemcpyHostToDevice, st));
    LmCounters h;
    memset(&h, 0, sizeof(h));
    h.n_cc = n_cc; h.n_words = (unsigned long long)n_crop_words; h.tempo_count = (unsigned long long)tempo_count;
    h.n_frames = n_frames; h.n_matched = n_matched; h.n_uniq = n_unique; h.n_active = 0;
    if (n_active > 0 && h_active && h_active_cc && h_active_last) {
        if (n_active > s->cap_uniq) { lm_set_error("lm_stream_import: active list larger than max_uniques"); return LM_ERR_CAPACITY; }
        LM_HIP(hipMemcpyAsync(s->active, h_active, (size_t)n_active * sizeof(int32_t), hipMemcpyHostToDevice, st));
        LM_HIP(hipMemcpyAsync(s->active_cc, h_active_cc, (size_t)n_active * sizeof(int32_t), hipMemcpyHostToDevice, st));
        LM_HIP(hipMemcpyAsync(s->active_last, h_active_last, (size_t)n_active * sizeof(int32_t), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(lm_k_import_active, dim3(lm_blocks(n_active, 256)), dim3(256), 0, st, s->cc, s->active_cc, n_active, s->active_box);
        h.n_active = n_active;
    }
    LM_HIP(hipMemcpyAsync(s->counters, &h, sizeof(h), hipMemcpyHostToDevice, st));
    LM_HIP(hipStreamSynchronize(st));
    s->frames_pushed = n_frames;
    s->frames_matched = n_matched;
    return LM_OK;
}

// pack records for the host: 8 x int32 per CC
__global__ void __launch_bounds__(256) lm_k_pack_records(const LmCcRec* __restrict__ cc, const int32_t* __restrict__ assign,
                                                         long long n, int32_t* __restrict__ out8, long long* __restrict__ crop_off)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const LmCcRec r = cc[i];
        int32_t* o = out8 + i * 8;
        o[0] = r.cc_id; o[1] = r.min_x; o[2] = r.max_x; o[3] = r.min_y; o[4] = r.max_y; o[5] = r.size; o[6] = r.frame;
        o[7] = assign[i];
        if (crop_off) crop_off[i] = (long long)r.crop_off;
    }
}

extern "C" int lm_stream_read(LmStream* s, int32_t* h_rec, int64_t* h_frame_off, int64_t* h_crop_off, uint32_t* h_crop,
                              int32_t* h_active, void* stream)
{
    if (!s) { lm_set_error("lm_stream_read: null stream"); return LM_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    int64_t k[7];
    int rc = lm_stream_counters(s, k, stream);
    if (rc) return rc;
    const long long n_cc = k[1];
    if ((h_rec || h_crop_off) && n_cc > 0) {
        const size_t need = (size_t)n_cc * (8 * sizeof(int32_t) + sizeof(long long));
        if (s->rd_scratch_bytes < need) {
            if (s->rd_scratch) (void)hipFree(s->rd_scratch);
            s->rd_scratch = nullptr; s->rd_scratch_bytes = 0;
            LM_HIP(hipMalloc(&s->rd_scratch, need + need / 4));
            s->rd_scratch_bytes = need + need / 4;
        }
        long long* d_off = (long long*)s->rd_scratch;                       // 8-byte aligned part first
        int32_t* d_rec = (int32_t*)(d_off + n_cc);
        hipLaunchKernelGGL(lm_k_pack_records, dim3(lm_blocks(n_cc, 256)), dim3(256), 0, st, s->cc, s->assign, n_cc, d_rec, d_off);
        if (h_rec) LM_HIP(hipMemcpyAsync(h_rec, d_rec, (size_t)n_cc * 8 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (h_crop_off) LM_HIP(hipMemcpyAsync(h_crop_off, d_off, (size_t)n_cc * sizeof(long long), hipMemcpyDeviceToHost, st));
        LM_HIP(hipStreamSynchronize(st));
    }
    if (h_frame_off) LM_HIP(hipMemcpyAsync(h_frame_off, s->frame_cc_off, (size_t)(k[0] + 1) * sizeof(long long), hipMemcpyDeviceToHost, st));
    if (h_crop && k[2] > 0) LM_HIP(hipMemcpyAsync(h_crop, s->crop, (size_t)k[2] * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    if (h_active && k[4] > 0) LM_HIP(hipMemcpyAsync(h_active, s->active, (size_t)k[4] * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    LM_HIP(hipStreamSynchronize(st));
    return LM_OK;
}

// ------------------------------------------------------------------------------------------------
// Frame-range sharding: the CC records + crops of whole frames as ONE flat device buffer, so that the gather to
// the rank that replays the temporal matching is one send / recv per rank over RCCL (no pickling, no host copy).
// Layout (bytes): [0,32) header int64 {n_frames, n_cc, n_words, magic}; int64 kept-CC counts per frame, padded to a multiple
// of four entries; the 32-byte LmCcRec records (frame numbers relative to the block's first frame, crop offsets relative to
// its first crop word); the uint32 crop words.
// ------------------------------------------------------------------------------------------------
#define LM_PACK_MAGIC 0x4c4d504b31ll   /* "LMPK1" */

static inline size_t lm_pack_rec_off(long long n_frames) { return 32 + (size_t)((n_frames + 3) & ~3ll) * 8; }
static inline size_t lm_pack_bytes(long long n_frames, long long n_cc, long long n_words)
{
    return lm_pack_rec_off(n_frames) + (size_t)n_cc * sizeof(LmCcRec) + (size_t)n_words * 4;
}

__global__ void __launch_bounds__(256) lm_k_pack_block(const LmCcRec* __restrict__ cc, const uint32_t* __restrict__ crop,
                                                       const long long* __restrict__ frame_cc_off, int first, int n_frames, long long c0,
                                                       long long n_cc, unsigned long long w0, long long nw, long long* __restrict__ head,
                                                       LmCcRec* __restrict__ orec, uint32_t* __restrict__ ocrop)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
    if (tid == 0) { head[0] = n_frames; head[1] = n_cc; head[2] = nw; head[3] = LM_PACK_MAGIC; }
    for (long long i = tid; i < n_frames; i += nth) head[4 + i] = frame_cc_off[first + i + 1] - frame_cc_off[first + i];
    for (long long i = tid; i < n_cc; i += nth) {
        LmCcRec r = cc[c0 + i];
        r.frame -= first;
        r.crop_off -= w0;
        orec[i] = r;
    }
    for (long long i = tid; i < nw; i += nth) ocrop[i] = crop[w0 + i];
}

__global__ void __launch_bounds__(256) lm_k_append_block(const LmCcRec* __restrict__ irec, const uint32_t* __restrict__ icrop, long long n_cc,
                                                         long long nw, LmCcRec* __restrict__ cc, uint32_t* __restrict__ crop,
                                                         const LmCounters* __restrict__ cnt)
{
    const long long c0 = cnt->n_cc;
    const unsigned long long w0 = cnt->n_words;
    const int f0 = cnt->n_frames;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
    for (long long i = tid; i < n_cc; i += nth) {
        LmCcRec r = irec[i];
        r.frame += f0;
        r.crop_off += w0;
        cc[c0 + i] = r;
    }
    for (long long i = tid; i < nw; i += nth) crop[w0 + i] = icrop[i];
}

// one workgroup, after lm_k_append_block: exclusive scan of the block's per-frame counts -> frame_cc_off, then the counters
__global__ void __launch_bounds__(1024) lm_k_append_offsets(const long long* __restrict__ head, long long* __restrict__ frame_cc_off,
                                                            LmCounters* __restrict__ cnt)
{
    const int n_frames = (int)head[0];
    const long long c0 = cnt->n_cc;
    const int f0 = cnt->n_frames;
    unsigned long long carry = 0;
    for (int base = 0; base < n_frames; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const unsigned k = (i < n_frames) ? (unsigned)head[4 + i] : 0u;
        unsigned tot;
        const unsigned ex = lm_block_excl_scan<1024>(k, &tot);
        if (i < n_frames) frame_cc_off[f0 + i] = c0 + (long long)(carry + ex);
        carry += tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        frame_cc_off[f0 + n_frames] = c0 + head[1];
        cnt->n_cc = c0 + head[1];
        cnt->n_words += (unsigned long long)head[2];
        cnt->n_frames = f0 + n_frames;
    }
}

// record range [c0, c0 + n_cc) and crop-word range [w0, w0 + nw) of frames [first, first + n); synchronises the stream
static int lm_pack_query(LmStream* s, int first, int n, long long* c0, long long* n_cc, unsigned long long* w0, long long* nw, hipStream_t st)
{
    LmCounters h;
    long long c01[2] = {0, 0};
    LM_HIP(hipMemcpyAsync(&h, s->counters, sizeof(h), hipMemcpyDeviceToHost, st));
    LM_HIP(hipMemcpyAsync(&c01[0], s->frame_cc_off + first, sizeof(long long), hipMemcpyDeviceToHost, st));
    LM_HIP(hipMemcpyAsync(&c01[1], s->frame_cc_off + first + n, sizeof(long long), hipMemcpyDeviceToHost, st));
    LM_HIP(hipStreamSynchronize(st));
    if (h.error) { lm_set_error("lm_stream_pack: stream capacity exceeded on device"); return h.error; }
    *c0 = c01[0];
    *n_cc = c01[1] - c01[0];
    *w0 = 0;
    *nw = 0;
    if (*n_cc > 0) {
        // records are emitted in (frame, label) order and their crops appended in the same order: the block's words are contiguous
        LmCcRec r0, r1;
        unsigned long long w1 = h.n_words;
        LM_HIP(hipMemcpyAsync(&r0, s->cc + c01[0], sizeof(LmCcRec), hipMemcpyDeviceToHost, st));
        if (c01[1] < h.n_cc) LM_HIP(hipMemcpyAsync(&r1, s->cc + c01[1], sizeof(LmCcRec), hipMemcpyDeviceToHost, st));
        LM_HIP(hipStreamSynchronize(st));
        if (c01[1] < h.n_cc) w1 = r1.crop_off;
        *w0 = r0.crop_off;
        *nw = (long long)(w1 - r0.crop_off);
    }
    return LM_OK;
}

extern "C" int lm_stream_pack_size(LmStream* s, int first_frame, int n_frames, int64_t* h_bytes, void* stream)
{
    if (!s || !h_bytes || first_frame < 0 || n_frames < 0 || first_frame + n_frames > s->frames_pushed) {
        lm_set_error("lm_stream_pack_size: bad arguments (frames [%d, %d) of %d pushed)", first_frame, first_frame + n_frames, s ? s->frames_pushed : 0);
        return LM_ERR_ARG;
    }
    long long c0, n_cc, nw;
    unsigned long long w0;
    const int rc = lm_pack_query(s, first_frame, n_frames, &c0, &n_cc, &w0, &nw, (hipStream_t)stream);
    if (rc) return rc;
    *h_bytes = (int64_t)lm_pack_bytes(n_frames, n_cc, nw);
    return LM_OK;
}

extern "C" int lm_stream_pack(LmStream* s, int first_frame, int n_frames, void* d_buf, int64_t bytes, void* stream)
{
    if (!s || !d_buf || first_frame < 0 || n_frames < 0 || first_frame + n_frames > s->frames_pushed || (((uintptr_t)d_buf) & 31)) {
        lm_set_error("lm_stream_pack: bad arguments (32-byte aligned device buffer, frames [%d, %d) of %d pushed)", first_frame,
                     first_frame + n_frames, s ? s->frames_pushed : 0);
        return LM_ERR_ARG;
    }
    long long c0, n_cc, nw;
    unsigned long long w0;
    const int rc = lm_pack_query(s, first_frame, n_frames, &c0, &n_cc, &w0, &nw, (hipStream_t)stream);
    if (rc) return rc;
    const size_t need = lm_pack_bytes(n_frames, n_cc, nw);
    if ((size_t)bytes < need) { lm_set_error("lm_stream_pack: buffer of %lld bytes, %lld needed", (long long)bytes, (long long)need); return LM_ERR_CAPACITY; }
    char* b = (char*)d_buf;
    const size_t rec_off = lm_pack_rec_off(n_frames);
    hipLaunchKernelGGL(lm_k_pack_block, dim3(lm_blocks(n_cc + nw / 4 + n_frames, 256, 4096)), dim3(256), 0, (hipStream_t)stream, s->cc, s->crop,
                       s->frame_cc_off, first_frame, n_frames, c0, n_cc, w0, nw, (long long*)b, (LmCcRec*)(b + rec_off),
                       (uint32_t*)(b + rec_off + (size_t)n_cc * sizeof(LmCcRec)));
    LM_HIP(hipGetLastError());
    return LM_OK;
}

extern "C" int lm_stream_append_packed(LmStream* s, const void* d_buf, int64_t bytes, void* stream)
{
    if (!s || !d_buf || bytes < 32 || (((uintptr_t)d_buf) & 31)) { lm_set_error("lm_stream_append_packed: bad arguments"); return LM_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    long long head[4];
    LM_HIP(hipMemcpyAsync(head, d_buf, sizeof(head), hipMemcpyDeviceToHost, st));
    LmCounters h;
    LM_HIP(hipMemcpyAsync(&h, s->counters, sizeof(h), hipMemcpyDeviceToHost, st));
    LM_HIP(hipStreamSynchronize(st));
    if (head[3] != LM_PACK_MAGIC || head[0] < 0 || head[1] < 0 || head[2] < 0 || (size_t)bytes < lm_pack_bytes(head[0], head[1], head[2])) {
        lm_set_error("lm_stream_append_packed: not a packed block (or truncated: %lld bytes)", (long long)bytes);
        return LM_ERR_ARG;
    }
    if (h.error) { lm_set_error("lm_stream_append_packed: stream capacity exceeded on device"); return h.error; }
    if (s->frames_pushed + head[0] > s->cap_frames || h.n_cc + head[1] > s->cap_cc || h.n_words + (unsigned long long)head[2] > s->cap_words) {
        lm_set_error("lm_stream_append_packed: stream too small (frames %lld/%d, ccs %lld/%lld, crop words %llu/%llu)", s->frames_pushed + head[0],
                     s->cap_frames, h.n_cc + head[1], s->cap_cc, h.n_words + (unsigned long long)head[2], s->cap_words);
        return LM_ERR_CAPACITY;
    }
    const char* b = (const char*)d_buf;
    const size_t rec_off = lm_pack_rec_off(head[0]);
    hipLaunchKernelGGL(lm_k_append_block, dim3(lm_blocks(head[1] + head[2] / 4 + 1, 256, 4096)), dim3(256), 0, st, (const LmCcRec*)(b + rec_off),
                       (const uint32_t*)(b + rec_off + (size_t)head[1] * sizeof(LmCcRec)), head[1], head[2], s->cc, s->crop, s->counters);
    if (head[1] > 0)
        hipLaunchKernelGGL(lm_k_crop_hash, dim3(lm_blocks(head[2] / 64 + 1, 4, 2048)), dim3(256), 0, st, s->cc, s->crop, (long long)h.n_cc, head[1],
                           h.n_words + (unsigned long long)head[2], s->chash);
    hipLaunchKernelGGL(lm_k_append_offsets, dim3(1), dim3(1024), 0, st, (const long long*)b, s->frame_cc_off, s->counters);
    LM_HIP(hipGetLastError());
    s->frames_pushed += (int)head[0];
    return LM_OK;
}

// Hand-off of a MATCHED stream to another rank (step 03 on a rank of its own): the packed block carries records and crops, this
// carries what the temporal matching added -- the unique index of every kept CC (cc_idx_per_frame, cc_stability_estimator.py:102,117)
// and the counters.  d_out: int32 [n_cc] + 4 x int64 {n_cc, n_unique, tempo_count, n_frames} (d_out must hold lm_stream_assign_bytes).
__global__ void __launch_bounds__(256) lm_k_assign_tail(const LmCounters* __restrict__ cnt, long long* __restrict__ tail)
{
    if (threadIdx.x == 0) { tail[0] = cnt->n_cc; tail[1] = cnt->n_uniq; tail[2] = (long long)cnt->tempo_count; tail[3] = cnt->n_matched; }
}

__global__ void __launch_bounds__(256) lm_k_assign_apply(const long long* __restrict__ tail, LmCounters* __restrict__ cnt)
{
    if (threadIdx.x == 0) { cnt->n_uniq = (int)tail[1]; cnt->tempo_count = (unsigned long long)tail[2]; cnt->n_matched = (int)tail[3]; }
}

static inline size_t lm_assign_bytes(long long n_cc) { return (((size_t)n_cc * 4 + 31) & ~(size_t)31) + 32; }

extern "C" int lm_stream_assign_bytes(LmStream* s, int64_t* bytes, void* stream)
{
    if (!s || !bytes) { lm_set_error("lm_stream_assign_bytes: bad arguments"); return LM_ERR_ARG; }
    LmCounters h;
    LM_HIP(hipMemcpyAsync(&h, s->counters, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    LM_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (h.error) { lm_set_error("lm_stream_assign_bytes: stream capacity exceeded on device"); return h.error; }
    *bytes = (int64_t)lm_assign_bytes(h.n_cc);
    return LM_OK;
}

extern "C" int lm_stream_export_assign(LmStream* s, void* d_out, int64_t bytes, void* stream)
{
    if (!s || !d_out || bytes < 32 || (((uintptr_t)d_out) & 31)) { lm_set_error("lm_stream_export_assign: bad arguments"); return LM_ERR_ARG; }
    if (s->frames_matched != s->frames_pushed) { lm_set_error("lm_stream_export_assign: %d frames are unmatched", s->frames_pushed - s->frames_matched); return LM_ERR_STATE; }
    hipStream_t st = (hipStream_t)stream;
    const long long n_cc = (long long)((bytes - 32) / 4);       // the caller sized the buffer with lm_stream_assign_bytes
    LM_HIP(hipMemcpyAsync(d_out, s->assign, (size_t)(bytes - 32), hipMemcpyDeviceToDevice, st));
    (void)n_cc;
    hipLaunchKernelGGL(lm_k_assign_tail, dim3(1), dim3(64), 0, st, s->counters, (long long*)((char*)d_out + bytes - 32));
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// the stream must hold exactly the frames the assignment belongs to (appended with lm_stream_append_packed), all of them unmatched
extern "C" int lm_stream_import_assign(LmStream* s, const void* d_in, int64_t bytes, void* stream)
{
    if (!s || !d_in || bytes < 32 || (((uintptr_t)d_in) & 31)) { lm_set_error("lm_stream_import_assign: bad arguments"); return LM_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    long long tail[4];
    LmCounters h;
    LM_HIP(hipMemcpyAsync(tail, (const char*)d_in + bytes - 32, 32, hipMemcpyDeviceToHost, st));
    LM_HIP(hipMemcpyAsync(&h, s->counters, sizeof(h), hipMemcpyDeviceToHost, st));
    LM_HIP(hipStreamSynchronize(st));
    if (tail[0] != h.n_cc || tail[3] != s->frames_pushed || (size_t)bytes != lm_assign_bytes(tail[0]) || s->frames_matched != 0 || tail[1] > s->cap_uniq) {
        lm_set_error("lm_stream_import_assign: the assignment (%lld CCs, %lld frames) does not belong to this stream (%lld CCs, %d frames, %d matched)",
                     tail[0], tail[3], h.n_cc, s->frames_pushed, s->frames_matched);
        return LM_ERR_ARG;
    }
    LM_HIP(hipMemcpyAsync(s->assign, d_in, (size_t)h.n_cc * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(lm_k_assign_apply, dim3(1), dim3(64), 0, st, (const long long*)((const char*)d_in + bytes - 32), s->counters);
    LM_HIP(hipGetLastError());
    s->frames_matched = s->frames_pushed;
    return LM_OK;
}

#include "lm_legacy.hip"
#include "lm_resize.hip"
