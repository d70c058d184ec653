// lm_match_kernels.hip -- CC record emission + temporal CC matching on gfx950 (hand-written HIP).
//
// Replaces (paths relative to /root/reference/ACCESS2021_release):
//   the ConnectedComponent objects built in labeler.py:171-189            -> lm_k_emit (records + bit crops)
//   CCStabilityEstimator.add_frame, content/cc_stability_estimator.py:41-155
//       bbox join via IntervalIndex (tools/interval_index.py:42-99)       -> lm_k_match candidate scan
//       ConnectedComponent.getOverlapFMeasure (connected_component.py:202-250) -> wave-cooperative AND+popcount
//       first-match-wins / new unique / retire (:90-145)                   -> lm_k_match + lm_k_update
//
// Within one frame every current CC decides independently (uniques born in the frame are not
// candidates, :80-84 are computed before the loop), so a frame is one data-parallel step; frames are
// sequential (stream order of kernel launches), the state lives in HBM between launches.
#include "lm_stream.h"

// ------------------------------------------------------------------------------------------------
// E1: batch offsets. One block; B <= 1024 frames per batch.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) lm_k_batch_offsets(const int32_t* __restrict__ frame_kept,
                                                           const uint32_t* __restrict__ frame_cropwords, int B,
                                                           LmCounters* __restrict__ cnt, long long* __restrict__ frame_cc_off,
                                                           long long* __restrict__ batch_cc_base,
                                                           unsigned long long* __restrict__ batch_word_base,
                                                           long long cap_cc, unsigned long long cap_words, int cap_frames)
{
    const int b = threadIdx.x;
    unsigned k = (b < B) ? (unsigned)frame_kept[b] : 0u;
    unsigned w = (b < B) ? frame_cropwords[b] : 0u;
    unsigned ktot, wtot;
    // 32-bit partial sums are fine per batch: B * cap < 2^32 is checked on the host
    unsigned kex = lm_block_excl_scan<1024>(k, &ktot);
    unsigned wex = lm_block_excl_scan<1024>(w, &wtot);
    const long long cc0 = cnt->n_cc;
    const unsigned long long w0 = cnt->n_words;
    const int f0 = cnt->n_frames;
    bool ok = (cc0 + (long long)ktot <= cap_cc) && (w0 + wtot <= cap_words) && (f0 + B <= cap_frames);
    __syncthreads();
    if (b < B) {
        batch_cc_base[b] = ok ? cc0 + kex : -1;
        batch_word_base[b] = w0 + wex;
        if (ok) frame_cc_off[f0 + b] = cc0 + kex;
    }
    if (threadIdx.x == 0) {
        if (ok) {
            frame_cc_off[f0 + B] = cc0 + ktot;
            cnt->n_cc = cc0 + ktot;
            cnt->n_words = w0 + wtot;
            cnt->n_frames = f0 + B;
        } else {
            cnt->error = LM_ERR_CAPACITY;
        }
    }
}

LM_DEV unsigned lm_mix32(unsigned h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

// ------------------------------------------------------------------------------------------------
// E2: emit kept-CC records and their bit crops; grid.y = frame in batch.
// Work is distributed over the CROP WORDS of the frame, not over its CCs: after an hour of lecture most of the ink belongs to
// a few large components (crops of 10^4..10^5 words) next to hundreds of glyph-sized ones, and a wave per CC left the GPU to
// one wave.
// Round 3: a workgroup takes a CONTIGUOUS range of the frame's crop words.  Before, every wave located its 64 words on its own --
// a binary search over kept_cropoff[] in L2 (10 dependent loads), then label -> box -> image word -> run tables -> final labels,
// ~16 dependent memory latencies for one word per lane (155 us per 64 dense 1080p frames, the waves parked 82 % of their time).
// Now the workgroup finds its first CC with a 256-way search (two rounds), stages the descriptors of the next LM_EM_WIN kept CCs in
// LDS once (offset, label, box), and every thread handles LM_EM_UNR words whose loads are issued together, phase by phase:
// image words, then run tables, then the final labels of the pieces (consecutive run ids: one batch of loads, not a chain).
// ------------------------------------------------------------------------------------------------
#define LM_EM_WIN 256       // kept CCs whose descriptors a workgroup stages per window
#ifndef LM_EM_UNR
#define LM_EM_UNR 4         // crop words per thread and pass
#endif
#define LM_EM_PRE 4         // final labels fetched ahead per crop word (its pieces have consecutive run ids)
#ifndef LM_EMIT_GRID
#define LM_EMIT_GRID 128   // workgroups per frame (1024 crop words per workgroup and pass: a dense 1080p frame has ~80 k)
#endif

// number of leading entries t = 0 .. 255 whose flag is set (flags are a prefix: the probed table ascends); every thread calls
LM_DEV int lm_em_count256(bool flag, int* s_cnt4)
{
    const unsigned long long bal = __ballot(flag);
    __syncthreads();            // s_cnt4 free again
    if (lm_lane() == 0) s_cnt4[threadIdx.x >> 6] = (int)__popcll(bal);
    __syncthreads();
    return s_cnt4[0] + s_cnt4[1] + s_cnt4[2] + s_cnt4[3];
}

__global__ void __launch_bounds__(256) lm_k_emit(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ starts,
                                                 const uint16_t* __restrict__ prefix, const uint32_t* __restrict__ rowoff,
                                                 const int32_t* __restrict__ final_label, const int32_t* __restrict__ st_min_y,
                                                 const int32_t* __restrict__ st_max_y, const int32_t* __restrict__ st_min_x,
                                                 const int32_t* __restrict__ st_max_x, const int32_t* __restrict__ st_count,
                                                 const int32_t* __restrict__ kept_label, const uint32_t* __restrict__ kept_cropoff,
                                                 const int32_t* __restrict__ frame_kept, const uint32_t* __restrict__ frame_cropwords,
                                                 const long long* __restrict__ batch_cc_base,
                                                 const unsigned long long* __restrict__ batch_word_base, LmCcRec* __restrict__ cc,
                                                 uint32_t* __restrict__ crop, uint32_t* __restrict__ chash, int first_frame, int WW, int H, int cap)
{
    __shared__ unsigned s_off[LM_EM_WIN + 1];
    __shared__ int4 s_d[LM_EM_WIN];         // x = label - 1, y = min_x | max_x << 16, z = min_y, w = words per crop row
    __shared__ unsigned s_hash[LM_EM_WIN];  // crop-hash contributions of this workgroup's words, per CC of the window
    __shared__ int s_cnt4[4];
    const int b = blockIdx.y;
    const long long cc_base = batch_cc_base[b];
    if (cc_base < 0) return;    // capacity error raised by lm_k_batch_offsets
    const long long off = (long long)b * cap;
    const int nk = frame_kept[b];
    const unsigned nwords = frame_cropwords[b];
    const unsigned long long wbase = batch_word_base[b];
    const int32_t* fin = final_label + off;
    const uint32_t* coffs = kept_cropoff + off;
    // ---- records: one thread per kept CC
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nk; k += gridDim.x * blockDim.x) {
        const int l0 = kept_label[off + k];
        LmCcRec r;
        r.cc_id = l0;
        r.size = st_count[off + l0];
        r.min_x = (int16_t)st_min_x[off + l0]; r.max_x = (int16_t)st_max_x[off + l0];
        r.min_y = (int16_t)st_min_y[off + l0]; r.max_y = (int16_t)st_max_y[off + l0];
        r.crop_off = wbase + coffs[k];
        r.frame = first_frame + b;
        r.pad = 0;
        cc[cc_base + k] = r;
    }
    // ---- crops: this workgroup's words [wb0, wb1) of the frame
    constexpr unsigned STEP = 256u * LM_EM_UNR;
    const unsigned per = (((nwords + gridDim.x - 1) / gridDim.x + STEP - 1) / STEP) * STEP;
    const unsigned long long wb0l = (unsigned long long)blockIdx.x * per;
    if (nk <= 0 || wb0l >= nwords) return;          // uniform in the workgroup
    const unsigned wb0 = (unsigned)wb0l;
    const unsigned wb1 = (nwords - wb0 < per) ? nwords : wb0 + per;
    const int tid = (int)threadIdx.x, lane = lm_lane();
    // largest k with coffs[k] <= wb0 (coffs[0] == 0): 256-way search, every round one load per thread
    int klo = 0;
    for (int len = nk; len > 1;) {
        const int stride = (len + 255) >> 8;
        const bool in = (long long)tid * stride < len;
        const int c = lm_em_count256(in && coffs[klo + tid * stride] <= wb0, s_cnt4);       // >= 1
        klo += (c - 1) * stride;
        len = (len - (c - 1) * stride < stride) ? len - (c - 1) * stride : stride;
    }
    for (unsigned cur = wb0; cur < wb1;) {
        // window: descriptors of kept CCs klo .. klo + nwin - 1; s_off[nwin ..] = where the next CC's words begin
        const int nwin = (nk - klo < LM_EM_WIN) ? nk - klo : LM_EM_WIN;
        __syncthreads();        // the previous window has been read
        const unsigned wnext = (klo + nwin < nk) ? coffs[klo + nwin] : nwords;
        if (tid < nwin) {
            const int l0 = kept_label[off + klo + tid];
            s_off[tid] = coffs[klo + tid];
            const int mnx = st_min_x[off + l0], mxx = st_max_x[off + l0];
            s_d[tid] = make_int4(l0, mnx | (mxx << 16), st_min_y[off + l0], (mxx >> 5) - (mnx >> 5) + 1);
        } else {
            s_off[tid] = wnext;
        }
        s_hash[tid] = 0;
        if (tid == 0) s_off[LM_EM_WIN] = wnext;
        __syncthreads();
        const unsigned wend = wnext < wb1 ? wnext : wb1;
        for (unsigned w0 = cur + (unsigned)(tid & ~63) * LM_EM_UNR; w0 < wend; w0 += STEP) {       // wave-uniform trip count
            // the CCs of the wave's first and last word: entries of the window at or below them, counted by ballots (scalar)
            const unsigned wlast = (wend - w0 < 64u * LM_EM_UNR) ? wend - 1 : w0 + 64u * LM_EM_UNR - 1;
            int jf = -1, jl = -1;
#pragma unroll
            for (int t = 0; t < LM_EM_WIN / 64; t++) {
                const unsigned o = s_off[t * 64 + lane];
                jf += (int)__popcll(__ballot(o <= w0));
                jl += (int)__popcll(__ballot(o <= wlast));
            }
            bool live[LM_EM_UNR];
            int jj[LM_EM_UNR], idx[LM_EM_UNR], half[LM_EM_UNR], l0[LM_EM_UNR];
            long long row[LM_EM_UNR], rw[LM_EM_UNR];
            unsigned msk[LM_EM_UNR], b32[LM_EM_UNR];
            // phase A: locate the words (LDS), request the image words
#pragma unroll
            for (int u = 0; u < LM_EM_UNR; u++) {
                const unsigned w = w0 + (unsigned)(u * 64 + lane);
                live[u] = w < wend;
                jj[u] = jf; idx[u] = 0; half[u] = 0; l0[u] = 0; row[u] = 0; rw[u] = 0; msk[u] = 0; b32[u] = 0;
                if (live[u]) {
                    int j = jf;
                    if (jf != jl)       // several CCs in the wave's words: walk forward from the first (s_off[nwin] stops the walk)
                        while (s_off[j + 1] <= w) j++;
                    jj[u] = j;
                    const int4 d = s_d[j];
                    l0[u] = d.x;
                    const int mnx = d.y & 0xffff, mxx = (int)((unsigned)d.y >> 16), mny = d.z, nw = d.w;
                    const int wx0 = mnx >> 5;
                    idx[u] = (int)(w - s_off[j]);
                    // row and column of the word in the crop: idx / nw by reciprocal (exact after one correction below 2^22), else by division
                    int r;
                    if (idx[u] < (1 << 22)) {
                        r = (int)((float)idx[u] * (1.0f / (float)nw));
                        const int rem = idx[u] - r * nw;
                        r += (rem >= nw) ? 1 : ((rem < 0) ? -1 : 0);
                    } else {
                        r = idx[u] / nw;
                    }
                    const int wx = wx0 + (idx[u] - r * nw);
                    row[u] = (long long)b * H + mny + r;
                    rw[u] = row[u] * WW + (wx >> 1);
                    half[u] = wx & 1;
                    const int x_lo = wx * 32;
                    unsigned m = 0xffffffffu;           // clip to the box in x
                    if (mnx > x_lo) m &= 0xffffffffu << (mnx - x_lo);
                    if (mxx < x_lo + 31) m &= 0xffffffffu >> (x_lo + 31 - mxx);
                    msk[u] = m;
                }
            }
#pragma unroll
            for (int u = 0; u < LM_EM_UNR; u++)
                if (live[u]) b32[u] = (unsigned)(bits[rw[u]] >> (32 * half[u])) & msk[u];
            // phase B: run tables of the non-empty words
            unsigned long long sw[LM_EM_UNR];
            int idfirst[LM_EM_UNR];
#pragma unroll
            for (int u = 0; u < LM_EM_UNR; u++) {
                sw[u] = 0; idfirst[u] = 0;
                if (b32[u]) {
                    sw[u] = starts[rw[u]];
                    idfirst[u] = (int)rowoff[row[u]] + (int)prefix[rw[u]] - 1;
                }
            }
            // phase C: final labels of the first LM_EM_PRE pieces of every word.  A piece's run id = idbase + run starts at or
            // left of its first pixel; a later piece of the same word begins with a start bit: consecutive ids.
            int labs[LM_EM_UNR][LM_EM_PRE], npieces[LM_EM_UNR];
#pragma unroll
            for (int u = 0; u < LM_EM_UNR; u++) {
                npieces[u] = 0;
                if (b32[u]) {
                    const int lo2 = __ffs((int)b32[u]) - 1;
                    idfirst[u] += __popcll(sw[u] & lm_lowmask_incl(half[u] * 32 + lo2));
                    npieces[u] = __popc(b32[u] & ~(b32[u] << 1));
                }
                if (npieces[u]) lm_load4(fin + idfirst[u], labs[u]);       // entries behind the word's last piece are read and ignored
                else { labs[u][0] = 0; labs[u][1] = 0; labs[u][2] = 0; labs[u][3] = 0; }
            }
            // phase D: keep the pieces of this CC, store, hash.  With the first pixels of the wanted pieces in `pick`, adding
            // `pick` to the word clears exactly those pieces (a carry runs through a piece and stops in the gap behind it).
            unsigned acc = 0;
#pragma unroll
            for (int u = 0; u < LM_EM_UNR; u++) {
                unsigned contrib = 0;
                if (live[u]) {
                    const unsigned want = (unsigned)(l0[u] + 1);
                    unsigned firsts = b32[u] & ~(b32[u] << 1), pick = 0;
#pragma unroll
                    for (int q = 0; q < LM_EM_PRE; q++) {
                        const unsigned fq = firsts & (0u - firsts);
                        pick |= ((unsigned)labs[u][q] == want) ? fq : 0u;
                        firsts ^= fq;
                    }
                    for (int q = LM_EM_PRE; firsts; q++) {      // words with more pieces than were fetched ahead
                        const unsigned fq = firsts & (0u - firsts);
                        pick |= ((unsigned)fin[idfirst[u] + q] == want) ? fq : 0u;
                        firsts ^= fq;
                    }
                    const unsigned out = b32[u] & (b32[u] ^ (b32[u] + pick));
                    crop[wbase + w0 + (unsigned)(u * 64 + lane)] = out;
                    contrib = lm_mix32(out + 0x9e3779b9u * (unsigned)idx[u]);
                }
                // crop hash for the twin detection, gathered per CC in LDS (global atomics on a CC's one hash word, one per crop word
                // or even per wave, were half of this kernel's time): the words of the wave's first CC (all of them in a large
                // component's waves) add up in registers first
                if (live[u] && jj[u] != jf) atomicAdd(&s_hash[jj[u]], contrib);
                else acc += contrib;
            }
            acc = lm_wave_sum(acc);
            if (lane == 0) atomicAdd(&s_hash[jf], acc);
        }
        __syncthreads();
        if (tid < nwin && s_hash[tid]) atomicAdd(&chash[cc_base + klo + tid], s_hash[tid]);        // other workgroups hold the CC's other words
        cur = wend;
        klo += nwin;
    }
}

// ------------------------------------------------------------------------------------------------
// M1: match the kept CCs of frame f against the active uniques. One wave per current CC.
// assign[c] = matched unique index, or -1 (new unique).
//
// Active list (ascending unique index, so scan order == the reference's sorted candidate order, :84):
//   active[a] unique index, active_cc[a] its first-seen CC record, active_box[a] its box,
//   active_last[a] last frame it was matched.
// Retirement is evaluated lazily: the reference removes u after frame g when g - last_u >= max_gap
// (:126-145, only for g >= 1); a removed unique is never touched again, so "u is a candidate at frame f"
// <=> f <= 1 or (f-1) - last_u < max_gap.  The list is compacted only now and then (lm_k_update).
// ------------------------------------------------------------------------------------------------
LM_DEV bool lm_box_hit(const LmCcRec& c, unsigned long long ub)
{
    int ux0 = (int)(ub & 0xffff), ux1 = (int)((ub >> 16) & 0xffff), uy0 = (int)((ub >> 32) & 0xffff), uy1 = (int)(ub >> 48);
    // inclusive boxes intersect  <=>  half-open intervals [min, max+1) overlap on both axes
    return c.min_x <= ux1 && ux0 <= c.max_x && c.min_y <= uy1 && uy0 <= c.max_y;
}

struct LmIsect { int y0, y1, wc0, nwc, awx0, anw, uwx0, unw; };

LM_DEV LmIsect lm_isect(const LmCcRec& a, const LmCcRec& u)
{
    LmIsect s;
    s.y0 = a.min_y > u.min_y ? a.min_y : u.min_y;
    s.y1 = a.max_y < u.max_y ? a.max_y : u.max_y;
    s.awx0 = a.min_x >> 5; s.anw = (a.max_x >> 5) - s.awx0 + 1;
    s.uwx0 = u.min_x >> 5; s.unw = (u.max_x >> 5) - s.uwx0 + 1;
    s.wc0 = s.awx0 > s.uwx0 ? s.awx0 : s.uwx0;
    int wc1 = (a.max_x >> 5) < (u.max_x >> 5) ? (a.max_x >> 5) : (u.max_x >> 5);
    s.nwc = wc1 - s.wc0 + 1;
    return s;
}

// AND + popcount over the intersection, words idx = first, first+step, ... (step 1: one lane does it all;
// step 64 with first = lane: the wave shares it and the caller sums the lanes)
LM_DEV int lm_overlap_words(const LmCcRec& a, const LmCcRec& u, const LmIsect& s, const uint32_t* __restrict__ crop,
                            int first, int step)
{
    const int total = s.nwc * (s.y1 - s.y0 + 1);
    int sum = 0;
    for (int idx = first; idx < total; idx += step) {
        int r = idx / s.nwc, j = idx - r * s.nwc;
        int y = s.y0 + r, wc = s.wc0 + j;
        unsigned wa = crop[a.crop_off + (unsigned long long)((y - a.min_y) * s.anw + (wc - s.awx0))];
        unsigned wu = crop[u.crop_off + (unsigned long long)((y - u.min_y) * s.unw + (wc - s.uwx0))];
        sum += __popc(wa & wu);
    }
    return sum;
}

LM_DEV bool lm_accept(int match, int size_cur, int size_uni, double min_recall, double min_precision)
{
    double recall = (double)match / (double)size_cur;        // connected_component.py:239
    double precision = (double)match / (double)size_uni;     // :240
    return recall >= min_recall && precision >= min_precision;   // cc_stability_estimator.py:99
}

#define LM_SCAN_CCS 16      // CC boxes per tile (LDS); one active box per thread -> at most 16 * 256 hits per step
#define LM_SCAN_PAIRS (LM_SCAN_CCS * 256)

LM_DEV bool lm_box_hit_packed(unsigned long long c, unsigned long long u)
{
    int cx0 = (int)(c & 0xffff), cx1 = (int)((c >> 16) & 0xffff), cy0 = (int)((c >> 32) & 0xffff), cy1 = (int)(c >> 48);
    int ux0 = (int)(u & 0xffff), ux1 = (int)((u >> 16) & 0xffff), uy0 = (int)((u >> 32) & 0xffff), uy1 = (int)(u >> 48);
    return cx0 <= ux1 && ux0 <= cx1 && cy0 <= uy1 && uy0 <= cy1;
}

LM_DEV unsigned long long lm_pack_box(const LmCcRec& r)
{
    return (unsigned long long)(unsigned short)r.min_x | ((unsigned long long)(unsigned short)r.max_x << 16) |
           ((unsigned long long)(unsigned short)r.min_y << 32) | ((unsigned long long)(unsigned short)r.max_y << 48);
}

// best[] key of frame f: smaller for later frames, so one 64-bit atomicMin array serves the whole stream without being
// re-initialised: a slot holds a decision of frame f iff its high word equals lm_frame_tag(f).
LM_DEV unsigned long long lm_frame_tag(int f) { return (unsigned long long)(0x7fffffff - f) << 32; }

// M1: candidate scan + evaluation, fused.  Box join (current CCs of frame f) x (active uniques), tiled: a step is
// (tile of LM_SCAN_CCS CC boxes in LDS) x (chunk of 256 active boxes, one per thread, alive-masked).  Hits of a step go to
// an LDS pair list (ds_add_rtn slot allocation), then the block evaluates them itself, 16 lanes per pair (AND + popcount on
// the bit crops, float64 recall/precision), and keeps the SMALLEST accepted active position per CC with a device-scope
// atomicMin -- the active list is ascending in unique index, so that is the reference's first match (:90-108).
// No global pair list, no per-hit global atomics (a returning atomic on one address serialises at ~88/us on gfx950).
__global__ void __launch_bounds__(256) lm_k_match(const LmCcRec* __restrict__ cc, const uint32_t* __restrict__ crop,
                                                  const long long* __restrict__ frame_cc_off, int f,
                                                  const unsigned long long* __restrict__ active_box, const int32_t* __restrict__ active_last,
                                                  const int32_t* __restrict__ active_cc, LmCounters* __restrict__ cnt,
                                                  unsigned long long* __restrict__ best, double min_recall, double min_precision,
                                                  int max_gap)
{
    __shared__ unsigned long long s_cbox[LM_SCAN_CCS];
    __shared__ int2 s_pairs[LM_SCAN_PAIRS];          // (cc index in tile, active position)
    __shared__ int s_np;
    __shared__ unsigned s_hits;
    if (cnt->error) return;     // a capacity error leaves frame_cc_off unwritten: do not touch it
    const long long c0 = frame_cc_off[f], c1 = frame_cc_off[f + 1];
    const int nC = (int)(c1 - c0);
    const int nA = cnt->n_active;
    const unsigned long long tag = lm_frame_tag(f);
    const int sub = (int)(threadIdx.x & 15), grp = (int)(threadIdx.x >> 4);
    if (threadIdx.x == 0) s_hits = 0;
    for (int ct = blockIdx.y * LM_SCAN_CCS; ct < nC; ct += gridDim.y * LM_SCAN_CCS) {
        const int tile = (nC - ct < LM_SCAN_CCS) ? nC - ct : LM_SCAN_CCS;
        for (int a0 = blockIdx.x * 256; a0 < nA; a0 += gridDim.x * 256) {
            __syncthreads();
            if ((int)threadIdx.x < tile) s_cbox[threadIdx.x] = lm_pack_box(cc[c0 + ct + threadIdx.x]);
            if (threadIdx.x == 0) s_np = 0;
            __syncthreads();
            const int a = a0 + (int)threadIdx.x;
            if (a < nA && ((f <= 1) || ((f - 1) - active_last[a] < max_gap))) {        // lazily retired entries are skipped
                const unsigned long long box = active_box[a];
                for (int j = 0; j < tile; j++)
                    if (lm_box_hit_packed(s_cbox[j], box)) s_pairs[atomicAdd(&s_np, 1)] = make_int2(j, a);
            }
            __syncthreads();
            const int np = s_np;
            if (threadIdx.x == 0) s_hits += (unsigned)np;
            // evaluate the step's pairs: 16 groups of 16 lanes
            const int np_pad = (np + 3) & ~3;       // whole waves stay in the loop for the shuffles
            for (int p = grp; p < np_pad; p += 16) {
                int m = 0;
                LmCcRec rec, urec;
                int2 pr = make_int2(0, 0);
                const bool live = p < np;
                if (live) {
                    pr = s_pairs[p];
                    rec = cc[c0 + ct + pr.x];
                    urec = cc[active_cc[pr.y]];
                    const LmIsect is = lm_isect(rec, urec);
                    m = lm_overlap_words(rec, urec, is, crop, sub, 16);
                }
#pragma unroll
                for (int d = 8; d >= 1; d >>= 1) m += __shfl_xor(m, d, 16);
                if (live && sub == 0 && lm_accept(m, rec.size, urec.size, min_recall, min_precision))
                    atomicMin(&best[ct + pr.x], tag | (unsigned long long)(unsigned)pr.y);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_hits) atomicAdd(&cnt->tempo_count, (unsigned long long)s_hits);
}

// ------------------------------------------------------------------------------------------------
// M2: first accepted candidate per CC (ascending unique index) -> assign; then number the frame's new
// uniques in CC order and append them to the active list; compact the list (drop retired entries, order
// preserved) when `compact` is set.  One block.
// ------------------------------------------------------------------------------------------------
#define LM_UPD_ITEMS 4

__global__ void __launch_bounds__(1024) lm_k_update(const LmCcRec* __restrict__ cc, const long long* __restrict__ frame_cc_off,
                                                    int f, int32_t* __restrict__ active, int32_t* __restrict__ active_cc,
                                                    unsigned long long* __restrict__ active_box, int32_t* __restrict__ active_last,
                                                    LmCounters* __restrict__ cnt, int32_t* __restrict__ assign,
                                                    const unsigned long long* __restrict__ best, int max_gap, int cap_uniq, int compact)
{
    if (cnt->error) return;
    const long long c0 = frame_cc_off[f], c1 = frame_cc_off[f + 1];
    const int n = (int)(c1 - c0);
    const int nU = cnt->n_uniq;
    int nA = cnt->n_active;
    __shared__ int s_overflow;
    if (threadIdx.x == 0) s_overflow = 0;
    __syncthreads();
    // ---- decisions
    const unsigned long long tag = lm_frame_tag(f);
    for (int i = threadIdx.x; i < n; i += 1024) {
        const unsigned long long k = best[i];
        int found = -1;
        if ((k & 0xffffffff00000000ull) == tag) {       // an accepted candidate of THIS frame
            const int pos = (int)(unsigned)k;
            found = active[pos];
            active_last[pos] = f;             // several CCs may hit the same unique: same value
        }
        assign[c0 + i] = found;
    }
    __syncthreads();
    // ---- optional compaction (entries that cannot be candidates at frame f + 1 any more)
    if (compact && f >= 1) {
        unsigned kept = 0;
        for (int base = 0; base < nA; base += 1024 * LM_UPD_ITEMS) {
            int32_t u[LM_UPD_ITEMS], uc[LM_UPD_ITEMS], ul[LM_UPD_ITEMS];
            unsigned long long ub[LM_UPD_ITEMS];
            unsigned keep[LM_UPD_ITEMS], mine = 0;
#pragma unroll
            for (int k = 0; k < LM_UPD_ITEMS; k++) {
                int i = base + (int)threadIdx.x * LM_UPD_ITEMS + k;
                keep[k] = 0;
                if (i < nA) {
                    u[k] = active[i]; uc[k] = active_cc[i]; ub[k] = active_box[i]; ul[k] = active_last[i];
                    keep[k] = (f - ul[k] < max_gap) ? 1u : 0u;
                }
                mine += keep[k];
            }
            unsigned tot;
            unsigned ex = lm_block_excl_scan<1024>(mine, &tot);   // barriers inside: all reads of this pass precede its writes
            unsigned o = kept + ex;
#pragma unroll
            for (int k = 0; k < LM_UPD_ITEMS; k++)
                if (keep[k]) { active[o] = u[k]; active_cc[o] = uc[k]; active_box[o] = ub[k]; active_last[o] = ul[k]; o++; }
            kept += tot;
        }
        nA = (int)kept;
        __syncthreads();
    }
    // ---- new uniques, in CC order
    unsigned carry = 0;
    for (int base = 0; base < n; base += 1024 * LM_UPD_ITEMS) {
        unsigned isnew[LM_UPD_ITEMS], mine = 0;
#pragma unroll
        for (int k = 0; k < LM_UPD_ITEMS; k++) {
            int i = base + (int)threadIdx.x * LM_UPD_ITEMS + k;
            isnew[k] = (i < n && assign[c0 + i] < 0) ? 1u : 0u;
            mine += isnew[k];
        }
        unsigned tot;
        unsigned ex = lm_block_excl_scan<1024>(mine, &tot);
        unsigned o = carry + ex;
#pragma unroll
        for (int k = 0; k < LM_UPD_ITEMS; k++) {
            if (!isnew[k]) continue;
            long long ci = c0 + base + (long long)threadIdx.x * LM_UPD_ITEMS + k;
            long long idx = (long long)nU + o;
            if (idx < cap_uniq) {
                const LmCcRec r = cc[ci];
                assign[ci] = (int32_t)idx;
                active[nA + o] = (int32_t)idx;
                active_cc[nA + o] = (int32_t)ci;
                active_last[nA + o] = f;
                active_box[nA + o] =
                    (unsigned long long)(unsigned short)r.min_x | ((unsigned long long)(unsigned short)r.max_x << 16) |
                    ((unsigned long long)(unsigned short)r.min_y << 32) | ((unsigned long long)(unsigned short)r.max_y << 48);
            } else {
                s_overflow = 1;
            }
            o++;
        }
        carry += tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_overflow) {
            cnt->error = LM_ERR_CAPACITY;
        } else {
            cnt->n_uniq = nU + (int)carry;
            cnt->n_active = nA + (int)carry;
            cnt->n_matched = f + 1;
        }
    }
}
