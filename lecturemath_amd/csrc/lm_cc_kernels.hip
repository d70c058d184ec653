// lm_cc_kernels.hip -- per-frame connected-component path on gfx950 (hand-written HIP).
//
// Replaces, for a batch of frames resident in HBM:
//   scipy.ndimage.label(content)                        AccessMath/preprocessing/content/labeler.py:126
//   CC_AgeBoundaries(labels, ages, ...)                 accessmath_lib.c:357-413
//   the per-CC crop loop / MIN_CC_PIXELS filter         labeler.py:171-189
//   sigmoid -> *255 -> trunc -> >=128 -> 255-x          lecturenet_v1/FCN_lecturenet.py:452,461-467 +
//                                                       video_worker/FCN_lecturenet_binarizer.py:54
// (paths relative to /root/reference/ACCESS2021_release).
//
// Design: the frame is bit-packed once (1 B/px read), all connectivity work happens on
// horizontal RUNS found with bit tricks on the 1-bit image (259 KB per 1080p frame, L2 resident), and
// the int32 label image is written exactly once (4 B/px) -- the algorithmic 5 B/px of the labelling
// roofline.  No provisional label image ever touches HBM.
#include "lm_common.h"

#if LM_HIP_EMULATED
extern char* lm_emu_dynsmem;
#define LM_DYN_SMEM(name) char* name = lm_emu_dynsmem
#else
#define LM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#endif

// ------------------------------------------------------------------------------------------------
// K0: fp32 logits -> inverted binary uint8 {0,255}
// binary = ((uint8)(sigmoid(x) * 255.0f) >= thr) ? 255 : 0 ; out = 255 - binary
// sigmoid(x) = 1 / (1 + exp(-x)) in fp32; see DESIGN.md "threshold edge" for the ulp discussion.
// ------------------------------------------------------------------------------------------------
LM_DEV unsigned lm_thr_px(float x, int thr, unsigned flip)
{
    float s = 1.0f / (1.0f + expf(-x));
    float v = s * 255.0f;
    unsigned u = (unsigned)v;            // truncation, v in [0, 255]
    return ((u >= (unsigned)thr) ? 255u : 0u) ^ flip;   // flip = 0xff: the worker's 255 - binary
}

__global__ void __launch_bounds__(256) lm_k_threshold_invert(const float* __restrict__ logits,
                                                             uint8_t* __restrict__ out, long long n, int thr, unsigned flip)
{
    long long stride = (long long)gridDim.x * blockDim.x;
    long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long n4 = n >> 2;
    const bool aligned = ((((uintptr_t)logits) & 15) == 0) && ((((uintptr_t)out) & 3) == 0);
    if (aligned) {
        for (; i4 < n4; i4 += stride) {
            float4 v = *(const float4*)(logits + i4 * 4);
            unsigned r = lm_thr_px(v.x, thr, flip) | (lm_thr_px(v.y, thr, flip) << 8) | (lm_thr_px(v.z, thr, flip) << 16) |
                         (lm_thr_px(v.w, thr, flip) << 24);
            *(unsigned*)(out + i4 * 4) = r;
        }
        long long tail = n4 * 4 + ((long long)blockIdx.x * blockDim.x + threadIdx.x);
        if (tail < n) out[tail] = (uint8_t)lm_thr_px(logits[tail], thr, flip);
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
            out[i] = (uint8_t)lm_thr_px(logits[i], thr, flip);
    }
}

// The per-pixel formula is monotone in x, so "sigmoid(x) * 255 truncates to >= thr" is "x >= x*" for one float x*.
// lm_k_thr_edge finds x* by bisection over the float ordinals WITH the formula above, then checks the equivalence for the 8192
// floats around x* -- far from the edge the product differs from thr by much more than the formula's rounding error.  A failed
// check (out[1] = 0) keeps the caller on lm_k_threshold_invert.  One wave.
LM_DEV unsigned lm_f2ord(float x) { const unsigned b = __builtin_bit_cast(unsigned, x); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
LM_DEV float lm_ord2f(unsigned o) { return __builtin_bit_cast(float, (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

__global__ void __launch_bounds__(64) lm_k_thr_edge(int thr, float* __restrict__ out)
{
    // ordinals of -inf .. +inf; the formula is false at -inf and true at +inf for 0 < thr <= 255 (the caller checks)
    unsigned lo = lm_f2ord(-__builtin_huge_valf()), hi = lm_f2ord(__builtin_huge_valf());     // invariant: formula(lo) = 0, formula(hi) = 255
    while (hi - lo > 1) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        if (lm_thr_px(lm_ord2f(mid), thr, 0u)) hi = mid; else lo = mid;
    }
    const float edge = lm_ord2f(hi);
    int good = 1;
    for (int k = 0; k < 128; k++) {
        const unsigned o = hi - 4096u + (unsigned)(k * 64) + threadIdx.x;
        const float x = lm_ord2f(o);
        const unsigned want = lm_thr_px(x, thr, 0u);
        const unsigned got = (x >= edge) ? 255u : 0u;
        if (want != got) good = 0;
    }
    good = __ballot(!good) == 0ull;
    if (threadIdx.x == 0) { out[0] = edge; out[1] = good ? 1.0f : 0.0f; }
}

#if LM_HIP_EMULATED
LM_DEV float4 lm_ld_stream(const float4* p) { return *p; }
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#else
typedef float lm_f4v __attribute__((ext_vector_type(4)));
LM_DEV float4 lm_ld_stream(const float4* p)        // read once: do not keep it in the caches
{
    const lm_f4v v = __builtin_nontemporal_load((const lm_f4v*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
#endif

// K0': out = (x >= edge) ? hi : lo per pixel; 16 pixels per thread and trip (four 16-byte loads in flight, one 16-byte store).
LM_DEV unsigned lm_cmp4(float4 v, float edge, unsigned on, unsigned off)
{
    return ((v.x >= edge) ? on : off) | (((v.y >= edge) ? on : off) << 8) | (((v.z >= edge) ? on : off) << 16) | (((v.w >= edge) ? on : off) << 24);
}

__global__ void __launch_bounds__(256) lm_k_threshold_cmp(const float* __restrict__ logits, uint8_t* __restrict__ out, long long n, float edge,
                                                          unsigned flip)
{
    const unsigned on = 255u ^ flip, off = flip;
    const long long n4 = n >> 2;
    const float4* src = (const float4*)logits;
    unsigned* dst = (unsigned*)out;
    // a workgroup takes 4 x 256 consecutive 4-pixel groups per trip: every load / store instruction of a wave is contiguous
    for (long long base = (long long)blockIdx.x * 1024; base < n4; base += (long long)gridDim.x * 1024) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const long long i = base + k * 256 + threadIdx.x;
            v[k] = (i < n4) ? lm_ld_stream(src + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const long long i = base + k * 256 + threadIdx.x;
            if (i < n4) __builtin_nontemporal_store(lm_cmp4(v[k], edge, on, off), dst + i);
        }
    }
    const long long tail = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x;
    if (tail < n) out[tail] = (uint8_t)((logits[tail] >= edge) ? on : off);
}

// ------------------------------------------------------------------------------------------------
// K1 helper: non-zero bytes of a dword -> 4 bits (the bit packing itself is fused into lm_k_band)
// ------------------------------------------------------------------------------------------------
// n / d for n < 2^24, d < 2^16 with one 64-bit multiply: m = floor(2^40 / d) + 1 (host), q = (n * m) >> 40.
// (Integer '/' is a 40-instruction (32-bit) or ~200-instruction (64-bit) software routine on the GPU; per-element divisions
// made lm_k_write_labels and the pack loop of lm_k_band instruction-bound.)
LM_DEV unsigned lm_fastdiv(unsigned n, unsigned long long m) { return (unsigned)(((unsigned long long)n * m) >> 40); }

LM_DEV unsigned lm_nz_nibble(unsigned d)
{
    // bit0 of every byte := OR of the byte's bits, then gather the four bits into a nibble
    d |= d >> 4;
    d |= d >> 2;
    d |= d >> 1;
    d &= 0x01010101u;
    return (d * 0x01020408u) >> 24 & 0xFu;
}

// ------------------------------------------------------------------------------------------------
// K4: union-find over runs, two levels.
//   K4a  lm_k_band: one workgroup per band of LmCtx::band_rows rows packs the band, scans its rows and keeps the band's
//        forest in LDS (ds atomics); the flattened band-local roots are written out as global parents.
//   K4b  seam rows between bands: the few remaining contacts, with device-scope atomics in L2.
// Root of a set = its smallest run id, so roots are the runs holding each component's first pixel.
// ------------------------------------------------------------------------------------------------
// Rows per band (run time, LmCtx::band_rows): the band's bit rows, run starts and prefixes (18 B per 64-px word) and a forest of
// LM_BAND_LDS runs live in LDS.  Two tunings: 32 rows up to 2048 px wide (1080p: WW = 30, 34 KB per workgroup, ~4 per CU);
// 16 rows above (4K: WW = 60, 33 KB; with 32 rows a dense 4K band has more than LM_BAND_LDS runs and falls back to L2
// unions: 617 vs 334 us per 16 dense 4K frames, profiles/r02_label_band_rows.txt).
#define LM_BAND_ROWS_MAX 32
#define LM_BAND_LDS 4096

LM_DEV int lm_find(const int32_t* parent, int x)
{
    // stale reads are harmless: parents only ever move to smaller ids of the same set
    int p = parent[x];
    while (p != x) {
        x = p;
        p = parent[x];
    }
    return x;
}

// find with path halving: every visited node is re-pointed to its grandparent (plain stores: a racing store can only write
// another valid ancestor, smaller than the node).  Dense frames build chains as long as a band is tall; without compression
// the union phase of lm_k_band spent 17 of its 22 us walking them (tools/band_phases.py).
LM_DEV int lm_find_halve(int32_t* parent, int x)
{
    int p = parent[x];
    while (p != x) {
        const int gp = parent[p];
        if (gp != p) parent[x] = gp;
        x = p;
        p = gp;
    }
    return x;
}

template <bool HALVE>
LM_DEV void lm_union_t(int32_t* parent, int a, int b)
{
    for (;;) {
        a = HALVE ? lm_find_halve(parent, a) : lm_find(parent, a);
        b = HALVE ? lm_find_halve(parent, b) : lm_find(parent, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;
    }
}

// LM_SEAM_HALVE (default 1): the finds of the seam unions halve the paths they walk.  A component that
// crosses every band of the frame leaves a chain of band roots as long as the frame has bands (34 at 1080p): without compression
// every contact of a seam row walked it hop by hop through L2 (and the flattening pass after it, run by run).  Plain stores beside the device-
// scope atomicMin are safe: a node that is re-pointed is not a root and never becomes one again, and what is stored is an ancestor
// of it in the same set; another XCD may keep reading the older, longer path.
#ifndef LM_SEAM_HALVE
#define LM_SEAM_HALVE 1
#endif

LM_DEV void lm_union(int32_t* parent, int a, int b)
{
    for (;;) {
#if LM_SEAM_HALVE
        a = lm_find_halve(parent, a);
        b = lm_find_halve(parent, b);
#else
        a = lm_find(parent, a);
        b = lm_find(parent, b);
#endif
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(&parent[a], b);     // global: device-scope RMW, coherent across XCDs; LDS: ds_min_rtn
        if (old == a) return;                   // a was a root and now hangs under b
        a = old;                                // a had been re-parented meanwhile: merge that tree with b too
    }
}

// contacts of cell (row, word w) with the row above: calls uni(a, b) with frame-relative run ids
template <class F>
LM_DEV void lm_cell_contacts(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ starts,
                             const uint16_t* __restrict__ prefix, const uint32_t* __restrict__ rowoff, long long row, int w,
                             int WW, F uni)
{
    const long long gid = row * WW + w;
    unsigned long long cur = bits[gid];
    if (!cur) return;
    unsigned long long v = cur & bits[gid - WW];
    if (!v) return;
    unsigned long long carry = 0;
    if (w > 0) carry = (bits[gid - 1] & bits[gid - WW - 1]) >> 63;
    unsigned long long ps = v & ~((v << 1) | carry);   // first column of every distinct (run, upper run) contact
    if (!ps) return;
    const unsigned long long s_cur = starts[gid], s_up = starts[gid - WW];
    const int base_cur = (int)rowoff[row] + (int)prefix[gid] - 1;
    const int base_up = (int)rowoff[row - 1] + (int)prefix[gid - WW] - 1;
    while (ps) {
        int p = __ffsll((long long)ps) - 1;
        ps &= ps - 1;
        unsigned long long m = lm_lowmask_incl(p);
        uni(base_cur + __popcll(s_cur & m), base_up + __popcll(s_up & m));
    }
}

// K1: uint8 rows -> bit mask (1 B/px HBM read, the only pass over the image), run starts, per-word run-start prefix and runs per
// row.  One wave per row: a lane loads 16 pixels per 16-byte load (two loads in flight at 1080p), four lanes' 16-bit masks make
// a 64-bit word, and the row's run structure follows from wave shuffles.  No LDS, no barriers: tens of thousands of
// independent waves stream the image (the fused version -- pack inside the band kernel -- had every resident workgroup load
// in step and then compute in step: 28 of its ~47 us per band were the load phase).
__global__ void __launch_bounds__(256) lm_k_pack_rows(const uint8_t* __restrict__ img, uint64_t* __restrict__ bits, uint64_t* __restrict__ starts,
                                                      uint16_t* __restrict__ prefix, uint32_t* __restrict__ rowcnt, int W, int WW, long long R)
{
    const int lane = lm_lane();
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= R) return;
    const uint8_t* src = img + row * W;
    const bool aligned = ((W & 15) == 0) && ((((uintptr_t)img) & 15) == 0);
    const int nchunks = (W + 15) >> 4;                 // 16-px chunks of the row
    unsigned running = 0;
    unsigned long long carry = 0;
    for (int w0 = 0; w0 < WW; w0 += 64) {              // 64 words = 4096 px per trip
        // chunk c of this trip = w0 * 4 + k * 64 + lane, k = 0..3: four independent loads per lane
        unsigned m16[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c = w0 * 4 + k * 64 + lane;
            unsigned m = 0;
            if (c < nchunks) {
                if (aligned) {
                    const uint4 v = *(const uint4*)(src + (long long)c * 16);
                    m = lm_nz_nibble(v.x) | (lm_nz_nibble(v.y) << 4) | (lm_nz_nibble(v.z) << 8) | (lm_nz_nibble(v.w) << 12);
                } else {
                    const int x = c * 16, lim = W - x < 16 ? W - x : 16;
                    for (int i = 0; i < lim; i++) m |= (unsigned)(src[x + i] != 0) << i;
                }
            }
            m16[k] = m;
        }
        // word w0 + lane = chunks 4 * lane .. 4 * lane + 3 of this trip: chunk q lives in register q >> 6 of lane q & 63
        unsigned long long bw = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int q = 4 * lane + j;
            unsigned v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned t = (unsigned)__shfl((int)m16[k], q & 63);
                if ((q >> 6) == k) v = t;
            }
            bw |= (unsigned long long)v << (16 * j);
        }
        const int w = w0 + lane;
        if (w >= WW) bw = 0;
        unsigned long long prevtop = __shfl_up(bw >> 63, 1);
        if (lane == 0) prevtop = carry;
        const unsigned long long st = bw & ~((bw << 1) | prevtop);
        const unsigned c = (unsigned)__popcll(st);
        const unsigned incl = lm_wave_incl_scan(c);
        if (w < WW) {
            bits[row * WW + w] = bw;
            starts[row * WW + w] = st;
            prefix[row * WW + w] = (uint16_t)(running + incl - c);
        }
        carry = __shfl(bw >> 63, 63);
        running += __shfl(incl, 63);
    }
    if (lane == 0) rowcnt[row] = running;
}

// K1' = K0' + K1 in one pass: fp32 logits -> bit mask, run starts, prefixes, runs per row; the {0, 255} frame is written only on
// request.  (Separately the threshold kernel wrote 1 B/px that lm_k_pack_rows read straight back: 105 + 29 us per 64 frames at 1080p
// for what is one 4 B/px read.)  A pixel is foreground when ((x >= edge) ? 255 : 0) ^ flip is non-zero (lm_k_threshold_cmp's output).
// One wave per row; per block of 1024 pixels a lane loads the float4s j * 64 + lane, j = 0..3 (every load instruction of the wave is
// 1 KB contiguous), the comparisons come back as wave ballots -- bit l of ballot (j, c) = component c of lane l's j-th float4 =
// pixel 4 * (64 j + l) + c of the block -- and lane w < 16 interleaves the four ballots' 16-bit pieces into word w of the block.
// W must be a multiple of 4 (16-byte rows).
LM_DEV unsigned long long lm_spread16(unsigned long long x)       // bit i -> bit 4 i
{
    x = (x | (x << 24)) & 0x000000ff000000ffull;
    x = (x | (x << 12)) & 0x000f000f000f000full;
    x = (x | (x << 6)) & 0x0303030303030303ull;
    x = (x | (x << 3)) & 0x1111111111111111ull;
    return x;
}

__global__ void __launch_bounds__(256) lm_k_pack_rows_logits(const float* __restrict__ logits, float edge, unsigned flip, uint8_t* __restrict__ img,
                                                             uint64_t* __restrict__ bits, uint64_t* __restrict__ starts, uint16_t* __restrict__ prefix,
                                                             uint32_t* __restrict__ rowcnt, int W, int WW, long long R)
{
    const int lane = lm_lane();
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= R) return;
    const float4* src = (const float4*)(logits + row * W);
    unsigned* dst = img ? (unsigned*)(img + row * W) : nullptr;
    const int n4 = W >> 2;
    const unsigned on = 255u ^ flip, off = flip;
    const bool fg_hi = on != 0u, fg_lo = off != 0u;         // is a pixel at / above the edge foreground?  one below it?
    unsigned running = 0;
    unsigned long long carry = 0;
    for (int w0 = 0; w0 < WW; w0 += 64) {                   // 64 words = 4096 px = 4 blocks per trip
        unsigned long long bw = 0;
#pragma unroll
        for (int blk = 0; blk < 4; blk++) {
            const int q0 = w0 * 16 + blk * 256;             // first float4 of the block
            if (q0 >= n4) break;
            float4 v[4];
            bool ok[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int q = q0 + j * 64 + lane;
                ok[j] = q < n4;
                v[j] = ok[j] ? lm_ld_stream(src + q) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            unsigned long long word = 0;
            const int wj = lane >> 2, sh = 16 * (lane & 3);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool hx = v[j].x >= edge, hy = v[j].y >= edge, hz = v[j].z >= edge, hw = v[j].w >= edge;
                if (dst && ok[j])
                    __builtin_nontemporal_store((hx ? on : off) | ((hy ? on : off) << 8) | ((hz ? on : off) << 16) | ((hw ? on : off) << 24), dst + q0 + j * 64 + lane);
                const unsigned long long b0 = __ballot(ok[j] && (hx ? fg_hi : fg_lo)), b1 = __ballot(ok[j] && (hy ? fg_hi : fg_lo)),
                                         b2 = __ballot(ok[j] && (hz ? fg_hi : fg_lo)), b3 = __ballot(ok[j] && (hw ? fg_hi : fg_lo));
                if (wj == j)
                    word = lm_spread16((b0 >> sh) & 0xffffull) | (lm_spread16((b1 >> sh) & 0xffffull) << 1) | (lm_spread16((b2 >> sh) & 0xffffull) << 2) |
                           (lm_spread16((b3 >> sh) & 0xffffull) << 3);
            }
            // lanes 0..15 hold the block's 16 words: word w0 + 16 blk + l belongs in lane 16 blk + l
            const unsigned long long moved = __shfl(word, lane & 15);
            if ((lane >> 4) == blk) bw = moved;
        }
        const int w = w0 + lane;
        if (w >= WW) bw = 0;
        unsigned long long prevtop = __shfl_up(bw >> 63, 1);
        if (lane == 0) prevtop = carry;
        const unsigned long long st = bw & ~((bw << 1) | prevtop);
        const unsigned c = (unsigned)__popcll(st);
        const unsigned incl = lm_wave_incl_scan(c);
        if (w < WW) {
            bits[row * WW + w] = bw;
            starts[row * WW + w] = st;
            prefix[row * WW + w] = (uint16_t)(running + incl - c);
        }
        carry = __shfl(bw >> 63, 63);
        running += __shfl(incl, 63);
    }
    if (lane == 0) rowcnt[row] = running;
}

// K4a: one workgroup per band: the band's bit rows / run starts / prefixes (written by lm_k_pack_rows, L2-resident) go to LDS,
// band-local run offsets, and the band's union-find forest in LDS.  Run ids are band-structured:
//     gid = band * SLOT + (runs of the band before the run)          SLOT = worst-case runs of a band
// which is still monotone in raster order, so no frame-wide scan is needed before the unions; `rowoff[row]` stores
// band * SLOT + local offset, which is all the later kernels need.
__global__ void __launch_bounds__(512) lm_k_band(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ starts,
                                                 const uint16_t* __restrict__ prefix, const uint32_t* __restrict__ rowcnt,
                                                 uint32_t* __restrict__ rowoff, int32_t* __restrict__ band_runs, int32_t* __restrict__ parent,
                                                 uint8_t* __restrict__ band_fallback, int H, int WW, int slot, int cap, int phases,
                                                 unsigned long long magic_ww, int brows, unsigned long long* __restrict__ stamps, int lds_runs)
{
    LM_DYN_SMEM(smem);
    const int b = blockIdx.y, band = blockIdx.x, nbands = gridDim.x;
#if !LM_HIP_EMULATED
#define LM_BAND_STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define LM_BAND_STAMP(k) do { } while (0)
#endif
    LM_BAND_STAMP(0);
    const int y0 = band * brows;
    const int nrows = (y0 + brows < H) ? brows : H - y0;
    const long long row0 = (long long)b * H + y0;
    unsigned long long* s_bits = (unsigned long long*)smem;                   // [brows][WW]
    unsigned long long* s_starts = s_bits + brows * WW;                       // [brows][WW]
    int32_t* s_par = (int32_t*)(s_starts + brows * WW);                       // [lds_runs]
    unsigned* s_rowoff = (unsigned*)(s_par + lds_runs);                       // [65]
    unsigned* s_rowcnt = s_rowoff + 65;                                       // [64]
    uint16_t* s_prefix = (uint16_t*)(s_rowcnt + 64);                          // [brows][WW]
    const int lane = lm_lane(), wave = (int)(threadIdx.x >> 6);
    // ---- 1. the band's rows (contiguous in the global tables) -> LDS
    {
        const int nw = nrows * WW;
        const uint64_t* gb = bits + row0 * WW;
        const uint64_t* gs = starts + row0 * WW;
        const uint16_t* gp = prefix + row0 * WW;
        for (int i = threadIdx.x; i < nw; i += blockDim.x) { s_bits[i] = gb[i]; s_starts[i] = gs[i]; s_prefix[i] = gp[i]; }
        if ((int)threadIdx.x < nrows) s_rowcnt[threadIdx.x] = rowcnt[row0 + threadIdx.x];
    }
    __syncthreads();
    LM_BAND_STAMP(1);
    // ---- 3. band-local row offsets
    if (wave == 0) {
        const unsigned v = (lane < nrows) ? s_rowcnt[lane] : 0u;
        const unsigned incl = lm_wave_incl_scan(v);
        if (lane < nrows) {
            s_rowoff[lane] = incl - v;
            rowoff[row0 + lane] = (unsigned)(band * slot) + incl - v;
        }
        if (lane == 63) s_rowoff[64] = incl;
    }
    __syncthreads();
    const int n = (int)s_rowoff[64];
    int32_t* par_g = parent + (long long)b * cap + (long long)band * slot;
    if (threadIdx.x == 0) {
        band_runs[b * nbands + band] = n;
        band_fallback[b * nbands + band] = (n > lds_runs) ? 1 : 0;
    }
    if (n == 0) return;
    if (n > lds_runs || phases == 2) {      // too many runs for the LDS forest (phases == 2: profiling aid, unions left to L2): identity parents, lm_k_band_union_global does the unions
        for (int i = threadIdx.x; i < n; i += blockDim.x) par_g[i] = band * slot + i;
        return;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) s_par[i] = i;
    __syncthreads();
    LM_BAND_STAMP(2);
    // ---- 4. unions between vertically adjacent runs, all from LDS
    // Per cell (64 px of a row against the row above): every distinct (run, upper run) contact, united on the spot with path
    // halving.  What this phase costs is the chain of dependent LDS accesses of the busiest lane of each wave (12-17 of the
    // kernel's 17-22 us per band on dense frames, tools/band_phases.py).  Measured and rejected on MI355X: several lanes per
    // cell (the replicated cell set-up costs more than the shorter chains save: 16 -> 23 / 32 / 46 us for 2 / 4 / 8 lanes),
    // listing the contacts in LDS first and uniting one contact per lane (12.6 us, but 16 KB more LDS: 3 instead of 4
    // workgroups per CU, same kernel time), run-centric stores + barrier-separated pointer jumping (round 1).
    for (int cell = threadIdx.x; cell < (nrows - 1) * WW; cell += blockDim.x) {
        const int r = 1 + (int)lm_fastdiv((unsigned)cell, magic_ww), w = cell - (r - 1) * WW;
        const unsigned long long cur = s_bits[r * WW + w];
        if (!cur) continue;
        const unsigned long long v = cur & s_bits[(r - 1) * WW + w];
        if (!v) continue;
        unsigned long long carry = 0;
        if (w > 0) carry = (s_bits[r * WW + w - 1] & s_bits[(r - 1) * WW + w - 1]) >> 63;
        unsigned long long ps = v & ~((v << 1) | carry);
        const unsigned long long s_cur = s_starts[r * WW + w], s_up = s_starts[(r - 1) * WW + w];
        const int base_cur = (int)s_rowoff[r] + (int)s_prefix[r * WW + w] - 1;
        const int base_up = (int)s_rowoff[r - 1] + (int)s_prefix[(r - 1) * WW + w] - 1;
        while (ps) {
            const int p = __ffsll((long long)ps) - 1;
            ps &= ps - 1;
            const unsigned long long m = lm_lowmask_incl(p);
            lm_union_t<true>(s_par, base_cur + __popcll(s_cur & m), base_up + __popcll(s_up & m));
        }
    }
    __syncthreads();
    LM_BAND_STAMP(3);
    for (int i = threadIdx.x; i < n; i += blockDim.x) par_g[i] = band * slot + lm_find(s_par, i);
    __syncthreads();
    LM_BAND_STAMP(4);
    if (stamps && threadIdx.x == 0) stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 5] = (unsigned long long)n;
}

#define LM_SEAM_CAP 1024       // contacts of one seam row kept in LDS (a row of W px has at most W / 2)

// K4b: seam rows between bands (device-scope atomics in L2), and -- for the rare bands whose forest did not fit the LDS
// (very dense frames) -- all of the band's own contacts as well.  One block per (band, frame).
LM_DEV void lm_seam_body(int b, int band, int nbands, const uint64_t* __restrict__ bits, const uint64_t* __restrict__ starts,
                         const uint16_t* __restrict__ prefix, const uint32_t* __restrict__ rowoff,
                         const uint8_t* __restrict__ band_fallback, int32_t* __restrict__ parent, int WW, int H, int cap, int brows)
{
    const int y0 = band * brows;
    const int y1 = (y0 + brows < H) ? y0 + brows : H;
    const long long row0 = (long long)b * H + y0;
    int32_t* par = parent + (long long)b * cap;
    // contacts of the band's first row with the last row of the band above: enumerated into LDS first (bit tricks only),
    // then one union per thread -- the unions are chains of dependent L2 accesses, so they must run side by side
    __shared__ int2 s_contact[LM_SEAM_CAP];
    __shared__ int s_ncontact;
    if (threadIdx.x == 0) s_ncontact = 0;
    __syncthreads();
    if (band > 0)
        for (int w = threadIdx.x; w < WW; w += blockDim.x)
            lm_cell_contacts(bits, starts, prefix, rowoff, row0, w, WW, [&](int a, int c) {
                const int slot = atomicAdd(&s_ncontact, 1);
                if (slot < LM_SEAM_CAP) s_contact[slot] = make_int2(a, c);
                else lm_union(par, a, c);
            });
    __syncthreads();
    const int ncontact = s_ncontact < LM_SEAM_CAP ? s_ncontact : LM_SEAM_CAP;
    for (int i = threadIdx.x; i < ncontact; i += blockDim.x) lm_union(par, s_contact[i].x, s_contact[i].y);
    if (band_fallback[b * nbands + band])
        for (int cell = threadIdx.x; cell < (y1 - y0 - 1) * WW; cell += blockDim.x) {
            const int r = cell / WW, w = cell - r * WW;
            lm_cell_contacts(bits, starts, prefix, rowoff, row0 + 1 + r, w, WW, [&](int a, int c) { lm_union(par, a, c); });
        }
}

__global__ void __launch_bounds__(256) lm_k_seam_union(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ starts,
                                                       const uint16_t* __restrict__ prefix, const uint32_t* __restrict__ rowoff,
                                                       const uint8_t* __restrict__ band_fallback, int32_t* __restrict__ parent,
                                                       int WW, int H, int cap, int brows)
{
    lm_seam_body(blockIdx.y, blockIdx.x, gridDim.x, bits, starts, prefix, rowoff, band_fallback, parent, WW, H, cap, brows);
}

// ------------------------------------------------------------------------------------------------
// K5: number the roots in run order (== raster order of each component's first pixel) and give every
// run its final 1-based label.
//   K5a  flatten + root flags as a bit mask (one ballot per 64 runs)           grid (x, B)
//   K5b  per frame: exclusive scan of the root-mask popcounts                  one block per frame
//   K5c  final[i] = rank(root(i)) + 1                                          grid (x, B)
// ------------------------------------------------------------------------------------------------
// K5a: one block per (band, frame): flatten, root flags (one ballot per 64 runs), and -- since the block sees all of its
// band's flags -- the exclusive popcount prefix of the band's 64-run words and the band's root count.
LM_DEV void lm_flatten_body(int b, int band, int nbands, int32_t* __restrict__ parent, const int32_t* __restrict__ band_runs,
                            unsigned long long* __restrict__ rootbits, uint32_t* __restrict__ wordprefix, uint32_t* __restrict__ band_roots,
                            int slot, int cap, int capw)
{
    const int n = band_runs[b * nbands + band];
    int32_t* par = parent + (long long)b * cap;
    unsigned long long* rb = rootbits + (long long)b * capw + (band * slot >> 6);
    uint32_t* wp = wordprefix + (long long)b * capw + (band * slot >> 6);
    const int lane = lm_lane();
    __shared__ unsigned s_pc[256];
    unsigned carry = 0;
    const int wave = (int)(threadIdx.x >> 6);
    for (int base = 0; base < n; base += 256 * 64) {          // 256 words of 64 runs per pass, words dealt round-robin to the waves
        s_pc[threadIdx.x] = 0;
        __syncthreads();
        for (int k = 0; k < 64; k += 4) {                      // four words per trip: their first parent loads fly together
            if (base + (k * 4 + wave) * 64 >= n) break;        // wave-uniform
            int gid[4], p[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = base + ((k + u) * 4 + wave) * 64 + lane;
                gid[u] = (i < n) ? band * slot + i : -1;
                p[u] = (gid[u] >= 0) ? par[gid[u]] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int word = (k + u) * 4 + wave;           // word index inside this pass
                const int i0 = base + word * 64;
                if (i0 >= n) break;                            // wave-uniform
                bool flag = false;
                if (gid[u] >= 0) {
                    int x = gid[u], q = p[u];
                    // (no path halving here: a node is re-pointed by its owner only, to the root -- a racing halving store could
                    // put a non-root ancestor back after the owner's store, and lm_k_apply_labels takes par[] as flat)
                    while (q != x) { x = q; q = par[x]; }      // lm_find, continued from the prefetched parent
                    par[gid[u]] = x;
                    flag = (x == gid[u]);
                }
                const unsigned long long m = __ballot(flag);
                if (lane == 0) { rb[i0 >> 6] = m; s_pc[word] = (unsigned)__popcll(m); }
            }
        }
        __syncthreads();
        unsigned tot;
        const unsigned ex = lm_block_excl_scan<256>(s_pc[threadIdx.x], &tot);
        const int myword = (base >> 6) + (int)threadIdx.x;
        if (myword * 64 < n) wp[myword] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) band_roots[b * nbands + band] = carry;
}

__global__ void __launch_bounds__(256) lm_k_flatten_flag(int32_t* __restrict__ parent, const int32_t* __restrict__ band_runs,
                                                          unsigned long long* __restrict__ rootbits, uint32_t* __restrict__ wordprefix,
                                                          uint32_t* __restrict__ band_roots, int slot, int cap, int capw)
{
    lm_flatten_body(blockIdx.y, blockIdx.x, gridDim.x, parent, band_runs, rootbits, wordprefix, band_roots, slot, cap, capw);
}

// K5b: final[i] = (roots in the bands above) + (roots before the run's root inside its band) + 1.  One block per
// (band, frame); the band bases are a 64-wide scan of the per-band root counts; block 0 of a frame also writes n_labels.
// the per-label statistics arrays of the frames a launch numbers (CC_AgeBoundaries' "nothing seen yet" values are written by the
// numbering kernel itself, which knows the label count: round 2 had a launch of its own for that, 6 us + a kernel boundary per batch)
struct LmStatInit { int32_t *min_y, *max_y, *min_x, *max_x, *count; int W, H; };

LM_DEV void lm_apply_body(int b, int band, int nbands, const int32_t* parent, const int32_t* __restrict__ band_runs, const unsigned long long* rootbits,
                          const uint32_t* wordprefix, const uint32_t* band_roots, uint32_t* __restrict__ band_base, int32_t* __restrict__ n_labels,
                          int32_t* __restrict__ final_label, int slot, int cap, int capw, const LmStatInit si)
{
    __shared__ unsigned s_base[1024];
    // exclusive scan of the frame's band root counts (nbands <= 1024), redundantly per block: a few hundred loads from L2
    unsigned carry = 0;
    for (int base = 0; base < nbands; base += 256) {
        const int j = base + (int)threadIdx.x;
        const unsigned v = (j < nbands) ? band_roots[b * nbands + j] : 0u;
        unsigned tot;
        const unsigned ex = lm_block_excl_scan<256>(v, &tot);
        if (j < nbands) s_base[j] = carry + ex;
        carry += tot;
    }
    __syncthreads();
    if (band == 0) {
        for (int j = threadIdx.x; j < nbands; j += blockDim.x) band_base[b * nbands + j] = s_base[j];
        if (threadIdx.x == 0) n_labels[b] = (int32_t)carry;
    }
    if (si.count) {     // this band's share of the frame's labels (every workgroup of the frame knows their number)
        const int nl = (int)carry, per = (nl + nbands - 1) / nbands;
        const int i1 = (band + 1) * per < nl ? (band + 1) * per : nl;
        const long long off = (long long)b * cap;
        for (int i = band * per + (int)threadIdx.x; i < i1; i += blockDim.x) {
            si.min_y[off + i] = si.H; si.max_y[off + i] = 0; si.min_x[off + i] = si.W; si.max_x[off + i] = 0; si.count[off + i] = 0;
        }
    }
    const int n = band_runs[b * nbands + band];
    const int32_t* par = parent + (long long)b * cap;
    const unsigned long long* rb = rootbits + (long long)b * capw;
    const uint32_t* wp = wordprefix + (long long)b * capw;
    int32_t* fin = final_label + (long long)b * cap;
    for (int i0 = threadIdx.x; i0 < n; i0 += blockDim.x * 4) {       // four runs per trip: independent lookup chains in flight
        int r[4];
        unsigned w[4];
        unsigned long long bm[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * (int)blockDim.x;
            r[u] = (i < n) ? par[band * slot + i] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { w[u] = wp[r[u] >> 6]; bm[u] = rb[r[u] >> 6]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * (int)blockDim.x;
            if (i < n)
                fin[band * slot + i] = (int32_t)(s_base[r[u] / slot] + w[u] + (unsigned)__popcll(bm[u] & lm_lowmask_excl(r[u] & 63)) + 1u);
        }
    }
}

__global__ void __launch_bounds__(256) lm_k_apply_labels(const int32_t* __restrict__ parent, const int32_t* __restrict__ band_runs,
                                                         const unsigned long long* __restrict__ rootbits,
                                                         const uint32_t* __restrict__ wordprefix, const uint32_t* __restrict__ band_roots,
                                                         uint32_t* __restrict__ band_base, int32_t* __restrict__ n_labels,
                                                         int32_t* __restrict__ final_label, int slot, int cap, int capw, const LmStatInit si)
{
    lm_apply_body(blockIdx.y, blockIdx.x, gridDim.x, parent, band_runs, rootbits, wordprefix, band_roots, band_base, n_labels, final_label, slot, cap, capw, si);
}

// ------------------------------------------------------------------------------------------------
// K4b + K5a + K5b as ONE launch ("per-frame pipelined middle"): a workgroup per (band, frame) runs its seam unions, waits
// until the frame's other bands have done theirs, flattens and flags its band, waits again (the numbering needs every band's root
// count and root flags), and numbers its runs.  The two waits are per-FRAME arrival counters in HBM -- three grid-wide kernel
// boundaries become two 34-workgroup rendezvous that other frames' workgroups compute under.  Hand-off per MI355X_MICROARCH.md
// ("Valid forms"): every wave's stores drained + workgroup barrier (__syncthreads), lane 0: agent-scope release, drain, relaxed
// agent-scope arrival; then one relaxed poll loop, one agent-scope acquire, drain, workgroup barrier, plain loads.
// Workgroups take (frame, band) from a ticket counter in arrival order, so a waiting workgroup only ever waits for workgroups that
// have started -- no assumption on dispatch order or residency.  Counters are never reset: the host passes the launch's base
// values (cumulative tickets / arrivals of the earlier launches on the same counter set).  GPU builds only: the CPU emulator runs
// workgroups one after another and keeps the three launches.
// ------------------------------------------------------------------------------------------------
#if !LM_HIP_EMULATED
LM_DEV void lm_frame_rendezvous(unsigned* counter, unsigned target)
{
    __syncthreads();                    // every wave: its stores are drained (s_waitcnt vmcnt(0)) and it has reached this point
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(8);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) lm_k_middle(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ starts, const uint16_t* __restrict__ prefix,
                                                   const uint32_t* __restrict__ rowoff, const uint8_t* __restrict__ band_fallback, int32_t* parent,
                                                   const int32_t* __restrict__ band_runs, unsigned long long* rootbits, uint32_t* wordprefix,
                                                   uint32_t* band_roots, uint32_t* __restrict__ band_base, int32_t* __restrict__ n_labels,
                                                   int32_t* __restrict__ final_label, int WW, int H, int cap, int brows, int slot, int capw, int nbands,
                                                   unsigned* __restrict__ sync, unsigned ticket_base, unsigned arrive_base, const LmStatInit si)
{
    // sync[0]: tickets; sync[1 + 2 * frame + phase]: arrivals of the frame's workgroups at rendezvous `phase`
    __shared__ unsigned s_ticket;
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ticket_base;
    __syncthreads();
    const int b = (int)(s_ticket / (unsigned)nbands), band = (int)(s_ticket - (unsigned)b * nbands);
    lm_seam_body(b, band, nbands, bits, starts, prefix, rowoff, band_fallback, parent, WW, H, cap, brows);
    lm_frame_rendezvous(sync + 1 + 2 * b, arrive_base + (unsigned)nbands);
    lm_flatten_body(b, band, nbands, parent, band_runs, rootbits, wordprefix, band_roots, slot, cap, capw);
    lm_frame_rendezvous(sync + 2 + 2 * b, arrive_base + (unsigned)nbands);
    lm_apply_body(b, band, nbands, parent, band_runs, rootbits, wordprefix, band_roots, band_base, n_labels, final_label, slot, cap, capw, si);
}
#endif

// ------------------------------------------------------------------------------------------------
// K6: write the int32 label image (4 B/px, the only HBM write of the labelling).  One thread handles LM_WL_Q quads
// of 4 pixels, quad = base + k*64 + lane, so every store instruction of a wave covers a contiguous 1 KiB while the
// (dependent) run-table lookups of the LM_WL_Q quads are independent of each other and issued phase by phase.
// ------------------------------------------------------------------------------------------------
#define LM_WL_Q 4

__global__ void __launch_bounds__(256) lm_k_write_labels(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ starts,
                                                         const uint16_t* __restrict__ prefix, const uint32_t* __restrict__ rowoff,
                                                         const int32_t* __restrict__ final_label, int32_t* __restrict__ labels,
                                                         int W, int H, int WW, int cap, unsigned long long magic_q, int contiguous)
{
    // The kernel is instruction-bound before it is memory-bound (the first version spent ~870 instructions and 54
    // exec-mask branches per trip): 32-bit index math, one lookup for the common "all ink of the quad is one run" case.
    const unsigned Q = (unsigned)(W + 3) >> 2;
    const unsigned total = (unsigned)H * Q;                 // quads of one frame (grid.y = frame)
    const int b = blockIdx.y;
    const bool vec = ((W & 3) == 0) && ((((uintptr_t)labels) & 15) == 0);
    const unsigned lane = (unsigned)lm_lane();
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint64_t* fbits = bits + (long long)b * H * WW;
    const uint64_t* fstarts = starts + (long long)b * H * WW;
    const uint16_t* fprefix = prefix + (long long)b * H * WW;
    const uint32_t* frowoff = rowoff + (long long)b * H;
    const int32_t* fin = final_label + (long long)b * cap;
    int32_t* lab_frame = labels + (long long)b * H * W;
    // contiguous != 0: every workgroup owns one contiguous span of the frame's quads (its waves interleave inside the span)
    const unsigned wpb = blockDim.x >> 6, chunk = 64 * LM_WL_Q;
    const unsigned span = contiguous ? ((total + gridDim.x - 1) / gridDim.x + wpb * chunk - 1) / (wpb * chunk) * (wpb * chunk) : 0u;
    const unsigned first = contiguous ? blockIdx.x * span + (threadIdx.x >> 6) * chunk : wave * chunk;
    const unsigned stop = contiguous ? ((blockIdx.x + 1) * span < total ? (blockIdx.x + 1) * span : total) : total;
    const unsigned step = contiguous ? wpb * chunk : nwaves * chunk;
    for (unsigned base = first; base < stop; base += step) {
        unsigned y[LM_WL_Q], x[LM_WL_Q], rw[LM_WL_Q], nib[LM_WL_Q];
#pragma unroll
        for (int k = 0; k < LM_WL_Q; k++) {
            const unsigned gid = base + k * 64 + lane;
            const unsigned g = gid < total ? gid : total - 1;        // clamp: duplicate work, the store is predicated
            y[k] = lm_fastdiv(g, magic_q);
            x[k] = (g - y[k] * Q) << 2;
            rw[k] = y[k] * (unsigned)WW + (x[k] >> 6);
            nib[k] = (unsigned)(fbits[rw[k]] >> (x[k] & 63u)) & 0xFu;
        }
#pragma unroll
        for (int k = 0; k < LM_WL_Q; k++) {
            int o0 = 0, o1 = 0, o2 = 0, o3 = 0;
            if (nib[k]) {
                const unsigned sh = x[k] & 63u;
                const unsigned long long st = fstarts[rw[k]];
                const int idb = (int)frowoff[y[k]] + (int)fprefix[rw[k]] + __popcll(st & lm_lowmask_excl((int)sh)) - 1;
                const unsigned sn = (unsigned)(st >> sh) & 0xFu;
                // run index of pixel j = idb + popc(sn & ((2 << j) - 1))
                const int jf = __ffs((int)nib[k]) - 1, jl = 31 - __clz((int)nib[k]);
                const int idf = idb + __popc(sn & ((2u << jf) - 1u)), idl = idb + __popc(sn & ((2u << jl) - 1u));
                const int lf = fin[idf];
                if (idf == idl) {                                   // one run: broadcast
                    o0 = (nib[k] & 1u) ? lf : 0; o1 = (nib[k] & 2u) ? lf : 0; o2 = (nib[k] & 4u) ? lf : 0; o3 = (nib[k] & 8u) ? lf : 0;
                } else {                                            // up to four runs (alternating pixels)
                    o0 = (nib[k] & 1u) ? fin[idb + __popc(sn & 1u)] : 0;
                    o1 = (nib[k] & 2u) ? fin[idb + __popc(sn & 3u)] : 0;
                    o2 = (nib[k] & 4u) ? fin[idb + __popc(sn & 7u)] : 0;
                    o3 = (nib[k] & 8u) ? fin[idb + __popc(sn & 15u)] : 0;
                }
            }
            const unsigned gid = base + k * 64 + lane;
            if (gid < total) {
                int32_t* dst = lab_frame + y[k] * (unsigned)W + x[k];
                if (vec) {
                    typedef int lm_i4 __attribute__((ext_vector_type(4)));
                    lm_i4 v;
                    v[0] = o0; v[1] = o1; v[2] = o2; v[3] = o3;
                    __builtin_nontemporal_store(v, (lm_i4*)dst);      // write-once stream: keep it out of the L2 working set
                } else {
                    const int oo[4] = {o0, o1, o2, o3};
                    for (unsigned j = 0; j < 4 && x[k] + j < (unsigned)W; j++) dst[j] = oo[j];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K7: per-label statistics from the runs (CC_AgeBoundaries semantics, ages all-zero on this path).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) lm_k_stats_init(int32_t* __restrict__ st_min_y, int32_t* __restrict__ st_max_y,
                                                       int32_t* __restrict__ st_min_x, int32_t* __restrict__ st_max_x,
                                                       int32_t* __restrict__ st_count, const int32_t* __restrict__ n_labels,
                                                       int W, int H, int cap)
{
    const int b = blockIdx.y;
    const int n = n_labels[b];
    long long off = (long long)b * cap;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        st_min_y[off + i] = H;
        st_max_y[off + i] = 0;
        st_min_x[off + i] = W;
        st_max_x[off + i] = 0;
        st_count[off + i] = 0;
    }
}

// One block per tile (shape below): pieces are first combined per label in an LDS
// hash table (ds atomics), then every (tile, label) pair costs at most five device-scope atomics.
#define LM_ST_SLOTS 512
// Tile = LM_ST_WORDS 64-px words x LM_ST_ROWS rows; consecutive lanes take consecutive words of a row (16 words = two full cache
// lines per row and table), a thread LM_ST_ROWS / LM_ST_RPP cells, all loaded together.  Per 64 dense 1080p frames, one lease: 4 words x 128 rows
// 82.9 us, 8 x 64 71.0, 16 x 32 69.0, 16 x 64 79.1; another lease: 16 x 32 75.4, 32 x 16 75.7, 32 x 32 76.8, 16 x 16 (one cell per thread) 86.7.
#ifndef LM_ST_ROWS
#define LM_ST_ROWS 32
#endif
#ifndef LM_ST_WORDS
#define LM_ST_WORDS 16     // tile width in 64-px words (a power of two <= 64)
#endif
#define LM_ST_RPP (256 / LM_ST_WORDS)      // rows of the tile one pass of the workgroup's 256 threads covers
#define LM_ST_PRE 4        // final labels fetched ahead per cell

__global__ void __launch_bounds__(256) lm_k_stats(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ starts,
                                                  const uint16_t* __restrict__ prefix, const uint32_t* __restrict__ rowoff,
                                                  const int32_t* __restrict__ final_label, int32_t* __restrict__ st_min_y,
                                                  int32_t* __restrict__ st_max_y, int32_t* __restrict__ st_min_x,
                                                  int32_t* __restrict__ st_max_x, int32_t* __restrict__ st_count, int WW, int H,
                                                  int cap)
{
    __shared__ int s_key[LM_ST_SLOTS], s_cnt[LM_ST_SLOTS], s_mnx[LM_ST_SLOTS], s_mxx[LM_ST_SLOTS], s_mny[LM_ST_SLOTS],
        s_mxy[LM_ST_SLOTS];
    const int b = blockIdx.z;
    const long long foff = (long long)b * cap;
    for (int i = threadIdx.x; i < LM_ST_SLOTS; i += blockDim.x) {
        s_key[i] = 0; s_cnt[i] = 0; s_mnx[i] = 0x7fffffff; s_mxx[i] = -1; s_mny[i] = 0x7fffffff; s_mxy[i] = -1;
    }
    __syncthreads();
    const int w = blockIdx.x * LM_ST_WORDS + (int)(threadIdx.x % LM_ST_WORDS);
    const int32_t* fin = final_label + foff;
    const int lane = lm_lane();
    // Round 3.  (1) The run-table loads of a non-empty cell are issued together and the final labels of its first LM_ST_PRE
    // pieces in one batch (pieces of a word have consecutive run ids) instead of one dependent load per piece.  (2) Consecutive
    // pieces of one label are combined in registers.  (3) The LDS table is updated in wave-synchronous rounds: late in a lecture
    // most ink belongs to a few large components, so most lanes of a wave report to the SAME slot, and 64 atomics on one LDS
    // address are served one after the other.  Per round the lanes that share the label of the first reporting lane (twice) add
    // their counts up in registers and send one atomic; a slot that already holds the label is found by a plain (broadcast) read,
    // and the box only moves for pieces on the component's rim (looked at before the atomic, as before).
    // (4) A thread's LM_ST_CELLS cells (one 64-px word in every 64th row of the tile) are loaded together, phase by phase: the
    // kernel's time was rounds of workgroups x a chain of five dependent memory latencies, not arithmetic.
    constexpr int NC = LM_ST_ROWS / LM_ST_RPP;
    unsigned long long firsts[NC], lasts[NC], sbits[NC];
    int id0[NC], labs[NC][LM_ST_PRE];
    bool next_cont[NC];
    long long gidc[NC], rowc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int y = blockIdx.y * LM_ST_ROWS + c * LM_ST_RPP + (int)(threadIdx.x / LM_ST_WORDS);
        gidc[c] = -1; firsts[c] = 0; lasts[c] = 0; sbits[c] = 0; id0[c] = 0; next_cont[c] = false;
        rowc[c] = (long long)b * H + y;
        if (y < H && w < WW) gidc[c] = rowc[c] * WW + w;
    }
    unsigned long long remc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) remc[c] = gidc[c] >= 0 ? bits[gidc[c]] : 0ull;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if (remc[c]) {
            sbits[c] = starts[gidc[c]];
            id0[c] = (int)rowoff[rowc[c]] + (int)prefix[gidc[c]] - 1;
            next_cont[c] = (w + 1 < WW) ? (bits[gidc[c] + 1] & 1ull) : false;
        }
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        int npieces = 0;
        if (remc[c]) {
            firsts[c] = remc[c] & ~(remc[c] << 1);         // first / last pixel of every piece of the word
            lasts[c] = remc[c] & ~(remc[c] >> 1);
            // run id of the first piece (idbase + 1 if it starts in this word, idbase if it continues from the left) and the
            // number of pieces: one per run start, plus the continuing one
            const int first_start = (int)((sbits[c] >> (__ffsll((long long)remc[c]) - 1)) & 1ull);
            id0[c] += first_start;
            npieces = __popcll(sbits[c]) + 1 - first_start;
        }
        // (entries behind the cell's last piece are read and ignored: final_label has slack behind its last frame)
        if (npieces) lm_load4(fin + id0[c], labs[c]);
        else { labs[c][0] = 0; labs[c][1] = 0; labs[c][2] = 0; labs[c][3] = 0; }
    }
#if defined(LM_STATS_CUT) && LM_STATS_CUT == 1
    { int acc = 0;
#pragma unroll
      for (int c = 0; c < NC; c++) for (int z = 0; z < LM_ST_PRE; z++) acc += labs[c][z];
      if (acc == 0x12345678) st_count[0] = 1;
      return; }
#endif
#pragma unroll
    for (int c = 0; c < NC; c++) {
    const int y = blockIdx.y * LM_ST_ROWS + c * LM_ST_RPP + (int)(threadIdx.x / LM_ST_WORDS);
    const unsigned long long s = sbits[c];
    int q = 0;
    int nxt = 0;                // label of the next piece (0: none left)
    if (firsts[c]) nxt = labs[c][0];
    while (__ballot(nxt != 0)) {
        // ---- per lane: gather the pieces up to the next change of label
        const int lab = nxt;
        int cnt = 0, mnx = 0x7fffffff, mxx = -1;
        bool has_start = false;
        while (nxt != 0 && nxt == lab) {
            const int lo = __ffsll((long long)firsts[c]) - 1, hi = __ffsll((long long)lasts[c]) - 1;
            firsts[c] &= firsts[c] - 1; lasts[c] &= lasts[c] - 1;
            cnt += hi - lo + 1;
            if ((s >> lo) & 1ull) { has_start = true; if (w * 64 + lo < mnx) mnx = w * 64 + lo; }
            if (hi < 63 || !next_cont[c]) mxx = w * 64 + hi;        // pieces come left to right
            q++;
            nxt = 0;
            if (firsts[c]) {
                if (q < LM_ST_PRE) {
                    nxt = labs[c][0];
#pragma unroll
                    for (int z = 1; z < LM_ST_PRE; z++) nxt = (q == z) ? labs[c][z] : nxt;
                } else {
                    nxt = fin[id0[c] + q];                            // 1-based
                }
            }
        }
#if defined(LM_STATS_CUT) && LM_STATS_CUT == 2
        if (cnt + mnx + mxx + (int)has_start == 0x12345678) st_count[0] = 1;
        continue;
#endif
        // ---- the wave reports (lab, cnt, box) of its lanes
        const bool valid = lab != 0;
        int slot = (int)(((unsigned)lab * 2654435761u) >> 23) & (LM_ST_SLOTS - 1);
        bool placed = false;
        if (valid) {
            for (int tries = 0; tries < 24; tries++) {
                int k = s_key[slot];
                if (k == 0) k = atomicCAS(&s_key[slot], 0, lab);
                if (k == 0 || k == lab) { placed = true; break; }
                slot = (slot + 1) & (LM_ST_SLOTS - 1);
            }
        }
        bool counted = !valid;
        unsigned long long todo = __ballot(valid && placed);
        for (int round = 0; round < 2 && todo; round++) {
            const int leader = __ffsll((long long)todo) - 1;
            const int ll = __shfl(lab, leader);
            const bool mine = valid && placed && lab == ll;
            const unsigned long long same = __ballot(mine);
            if (__popcll(same) >= 4) {
                const int sum = lm_wave_sum(mine ? cnt : 0);
                if (lane == leader) atomicAdd(&s_cnt[slot], sum);
                counted = counted || mine;
            }
            todo &= ~same;
        }
        if (valid && placed) {
            if (!counted) atomicAdd(&s_cnt[slot], cnt);
            if (has_start) {
                if (mnx < s_mnx[slot]) atomicMin(&s_mnx[slot], mnx);
                if (y < s_mny[slot]) atomicMin(&s_mny[slot], y);
                if (y > s_mxy[slot]) atomicMax(&s_mxy[slot], y);
            }
            if (mxx >= 0 && mxx > s_mxx[slot]) atomicMax(&s_mxx[slot], mxx);
        } else if (valid) {      // table full (very dense tile): straight to L2
            const long long cc = foff + lab - 1;
            atomicAdd(&st_count[cc], cnt);
            if (has_start) {
                atomicMin(&st_min_x[cc], mnx);
                atomicMin(&st_min_y[cc], y);
                atomicMax(&st_max_y[cc], y);
            }
            if (mxx >= 0) atomicMax(&st_max_x[cc], mxx);
        }
    }
    }
    __syncthreads();
#if defined(LM_STATS_CUT) && LM_STATS_CUT == 3
    if (s_cnt[threadIdx.x] == 0x12345678) st_count[0] = 1;
    return;
#endif
    for (int i = threadIdx.x; i < LM_ST_SLOTS; i += blockDim.x) {
        const int lab = s_key[i];
        if (!lab) continue;
        const long long cc = foff + lab - 1;
        atomicAdd(&st_count[cc], s_cnt[i]);
        // same here: hundreds of tiles report the same large label, nearly all of them from inside its box (the plain loads
        // may be stale -- values only move one way, so that only costs an atomic that changes nothing)
        if (s_mnx[i] != 0x7fffffff && s_mnx[i] < st_min_x[cc]) atomicMin(&st_min_x[cc], s_mnx[i]);
        if (s_mny[i] != 0x7fffffff && s_mny[i] < st_min_y[cc]) atomicMin(&st_min_y[cc], s_mny[i]);
        if (s_mxy[i] >= 0 && s_mxy[i] > st_max_y[cc]) atomicMax(&st_max_y[cc], s_mxy[i]);
        if (s_mxx[i] >= 0 && s_mxx[i] > st_max_x[cc]) atomicMax(&st_max_x[cc], s_mxx[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// K7b: CC_AgeBoundaries from an arbitrary int32 label image (+ fp32 ages): the literal drop-in
// entry point (labels may come from elsewhere, e.g. is_labeled=True callers, labeler.py:127-130).
// Age semantics (accessmath_lib.c:405-407, a sequential raster scan with -1 as "nothing yet"):
//     if (age_out < 0 || age[px] < age_out) age_out = age[px];
// so a NEGATIVE age makes the next pixel of the label overwrite it whatever its value.  In closed form, with j the last
// pixel of the label (raster order) whose age is negative: the result is that pixel's age if it is the label's last pixel,
// otherwise the minimum over the pixels after j (all of them >= 0); without negative ages, the plain minimum.  Two passes:
// lm_k_ab_scan finds, per label, the raster index of its last pixel and of its last negative-age pixel, lm_k_ab_age takes the
// minimum (bit patterns of non-negative floats order like ints) over the pixels behind the latter.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) lm_k_ab_init(int32_t* mny, int32_t* mxy, int32_t* mnx, int32_t* mxx, int32_t* cnt,
                                                    int32_t* age_bits, int32_t* last_px, int32_t* last_neg, int W, int H, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        mny[i] = H; mxy[i] = 0; mnx[i] = W; mxx[i] = 0; cnt[i] = 0;
        age_bits[i] = 0x7fffffff;   // "no pixel seen"
        last_px[i] = -1; last_neg[i] = -1;
    }
}

__global__ void __launch_bounds__(256) lm_k_ab_scan(const int32_t* __restrict__ labels, const float* __restrict__ ages,
                                                    int W, int H, int n, int32_t* mny, int32_t* mxy, int32_t* mnx,
                                                    int32_t* mxx, int32_t* cnt, int32_t* last_px, int32_t* last_neg)
{
    long long total = (long long)W * H;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int l = labels[i];
        if (l <= 0 || l > n) continue;   // labels above count_labels would be out of bounds in the reference; ignored here
        int y = (int)(i / W), x = (int)(i - (long long)y * W);
        int k = l - 1;
        atomicMin(&mny[k], y);
        atomicMax(&mxy[k], y);
        atomicMin(&mnx[k], x);
        atomicMax(&mxx[k], x);
        atomicAdd(&cnt[k], 1);
        if (ages) {
            atomicMax(&last_px[k], (int)i);
            if (ages[i] < 0.0f) atomicMax(&last_neg[k], (int)i);
        }
    }
}

__global__ void __launch_bounds__(256) lm_k_ab_age(const int32_t* __restrict__ labels, const float* __restrict__ ages, int W, int H, int n,
                                                   const int32_t* __restrict__ last_neg, int32_t* age_bits)
{
    long long total = (long long)W * H;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int l = labels[i];
        if (l <= 0 || l > n) continue;
        const int k = l - 1;
        const float a = ages ? ages[i] : 0.0f;
        if ((int)i <= last_neg[k] || !(a >= 0.0f)) continue;
        int ab;
        memcpy(&ab, &a, 4);
        atomicMin(&age_bits[k], ab & 0x7fffffff);       // -0.0f counts as 0.0f
    }
}

__global__ void __launch_bounds__(256) lm_k_ab_finish(const int32_t* age_bits, const int32_t* last_px, const int32_t* last_neg,
                                                      const float* __restrict__ ages, float* out_age, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int ab = age_bits[i];
        float a = -1.0f;
        if (ages && last_neg[i] >= 0 && last_neg[i] == last_px[i]) a = ages[last_neg[i]];     // the label ends on a negative age
        else if (ab != 0x7fffffff) memcpy(&a, &ab, 4);
        out_age[i] = a;
    }
}

// ------------------------------------------------------------------------------------------------
// K8: kept-CC selection (count >= min_pixels) + crop layout. One block per frame.
// Crop of a CC = its pixels as bit rows aligned to ABSOLUTE 32-pixel columns of the frame:
// words (min_x>>5 .. max_x>>5) for each row min_y..max_y, so two crops AND together without shifts.
// ------------------------------------------------------------------------------------------------
#define LM_SEL_ITEMS 4       // labels per thread and pass: their loads are in flight together, one block scan per 4096 labels

__global__ void __launch_bounds__(1024) lm_k_select(const int32_t* __restrict__ st_min_y, const int32_t* __restrict__ st_max_y,
                                                    const int32_t* __restrict__ st_min_x, const int32_t* __restrict__ st_max_x,
                                                    const int32_t* __restrict__ st_count, const int32_t* __restrict__ n_labels,
                                                    int32_t* __restrict__ kept_label, uint32_t* __restrict__ kept_cropoff,
                                                    int32_t* __restrict__ frame_kept, uint32_t* __restrict__ frame_cropwords,
                                                    int cap, int min_pixels)
{
    __shared__ unsigned long long s_wsum[16];
    __shared__ unsigned long long s_tot;
    const int b = blockIdx.x;
    const long long off = (long long)b * cap;
    const int n = n_labels[b];
    const int lane = lm_lane(), wid = (int)(threadIdx.x >> 6);
    // kept count (high 24 bits) and crop words (low 40 bits; a frame's crop words fit 32) scanned as one 64-bit value
    unsigned long long carry = 0;
    for (int base = 0; base < n; base += 1024 * LM_SEL_ITEMS) {
        int cnt[LM_SEL_ITEMS], mnx[LM_SEL_ITEMS], mxx[LM_SEL_ITEMS], mny[LM_SEL_ITEMS], mxy[LM_SEL_ITEMS];
#pragma unroll
        for (int k = 0; k < LM_SEL_ITEMS; k++) {
            const int i = base + (int)threadIdx.x * LM_SEL_ITEMS + k;
            const bool in = i < n;
            cnt[k] = in ? st_count[off + i] : -1;
            mnx[k] = in ? st_min_x[off + i] : 0; mxx[k] = in ? st_max_x[off + i] : 0;
            mny[k] = in ? st_min_y[off + i] : 0; mxy[k] = in ? st_max_y[off + i] : 0;
        }
        unsigned long long v[LM_SEL_ITEMS], mine = 0;
#pragma unroll
        for (int k = 0; k < LM_SEL_ITEMS; k++) {
            v[k] = 0;
            if (cnt[k] >= min_pixels && cnt[k] >= 0) {
                const unsigned nw = (unsigned)((mxx[k] >> 5) - (mnx[k] >> 5) + 1);
                v[k] = (1ull << 40) | (unsigned long long)(nw * (unsigned)(mxy[k] - mny[k] + 1));
            }
            mine += v[k];
        }
        const unsigned long long incl = lm_wave_incl_scan(mine);
        __syncthreads();            // s_wsum free again
        if (lane == 63) s_wsum[wid] = incl;
        __syncthreads();
        if (wid == 0) {
            const unsigned long long t = (lane < 16) ? s_wsum[lane] : 0ull;
            const unsigned long long ti = lm_wave_incl_scan(t);
            if (lane < 16) s_wsum[lane] = ti - t;
            if (lane == 15) s_tot = ti;
        }
        __syncthreads();
        unsigned long long o = carry + s_wsum[wid] + incl - mine;
#pragma unroll
        for (int k = 0; k < LM_SEL_ITEMS; k++) {
            if (v[k]) {
                kept_label[off + (long long)(o >> 40)] = base + (int)threadIdx.x * LM_SEL_ITEMS + k;
                kept_cropoff[off + (long long)(o >> 40)] = (uint32_t)(o & ((1ull << 40) - 1ull));
            }
            o += v[k];
        }
        carry += s_tot;
    }
    if (threadIdx.x == 0) {
        frame_kept[b] = (int32_t)(carry >> 40);
        frame_cropwords[b] = (uint32_t)(carry & ((1ull << 40) - 1ull));
    }
}
