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

// ------------------------------------------------------------------------------------------------
// E2: emit kept-CC records and their bit crops. One wave per kept CC; grid.y = frame in batch.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) lm_k_emit(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ starts,
                                                 const uint16_t* __restrict__ prefix, const uint32_t* __restrict__ rowoff,
                                                 const int32_t* __restrict__ final_label, const int32_t* __restrict__ st_min_y,
                                                 const int32_t* __restrict__ st_max_y, const int32_t* __restrict__ st_min_x,
                                                 const int32_t* __restrict__ st_max_x, const int32_t* __restrict__ st_count,
                                                 const int32_t* __restrict__ kept_label, const uint32_t* __restrict__ kept_cropoff,
                                                 const int32_t* __restrict__ frame_kept, const long long* __restrict__ batch_cc_base,
                                                 const unsigned long long* __restrict__ batch_word_base, LmCcRec* __restrict__ cc,
                                                 uint32_t* __restrict__ crop, int first_frame, int WW, int H, int cap)
{
    const int b = blockIdx.y;
    const long long cc_base = batch_cc_base[b];
    if (cc_base < 0) return;    // capacity error raised by lm_k_batch_offsets
    const long long off = (long long)b * cap;
    const int nk = frame_kept[b];
    const int lane = lm_lane();
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    const int32_t* fin = final_label + off;
    for (int k = wave; k < nk; k += nwaves) {
        const int l0 = kept_label[off + k];
        const int mnx = st_min_x[off + l0], mxx = st_max_x[off + l0], mny = st_min_y[off + l0], mxy = st_max_y[off + l0];
        const unsigned long long coff = batch_word_base[b] + kept_cropoff[off + k];
        if (lane == 0) {
            LmCcRec r;
            r.cc_id = l0;
            r.size = st_count[off + l0];
            r.min_x = (int16_t)mnx; r.max_x = (int16_t)mxx; r.min_y = (int16_t)mny; r.max_y = (int16_t)mxy;
            r.crop_off = coff;
            r.frame = first_frame + b;
            r.pad = 0;
            cc[cc_base + k] = r;
        }
        const int wx0 = mnx >> 5;
        const int nw = (mxx >> 5) - wx0 + 1;
        const int total = nw * (mxy - mny + 1);
        for (int idx = lane; idx < total; idx += 64) {
            int r = idx / nw, j = idx - r * nw;
            int wx = wx0 + j;
            long long row = (long long)b * H + mny + r;
            long long rw = row * WW + (wx >> 1);
            int half = wx & 1;
            unsigned b32 = (unsigned)(bits[rw] >> (32 * half));
            int x_lo = wx * 32;
            // clip to the box in x
            unsigned m = 0xffffffffu;
            if (mnx > x_lo) m &= 0xffffffffu << (mnx - x_lo);
            if (mxx < x_lo + 31) m &= 0xffffffffu >> (x_lo + 31 - mxx);
            b32 &= m;
            unsigned out = 0;
            if (b32) {
                const unsigned long long s = starts[rw];
                const int idbase = (int)rowoff[row] + (int)prefix[rw] - 1;
                unsigned rem = b32;
                while (rem) {
                    int lo = __ffs((int)rem) - 1;
                    unsigned t = ~(rem >> lo);
                    int len = t ? (__ffs((int)t) - 1) : 32;
                    if (len > 32 - lo) len = 32 - lo;
                    unsigned piece = ((len >= 32) ? 0xffffffffu : ((1u << len) - 1u)) << lo;
                    int id = idbase + __popcll(s & lm_lowmask_incl(half * 32 + lo));
                    if (fin[id] == l0 + 1) out |= piece;
                    rem &= ~piece;
                }
            }
            crop[coff + idx] = out;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// M1: match the kept CCs of frame f against the active uniques. One wave per current CC.
// assign[c] = matched unique index, or -1 (new unique).
// ------------------------------------------------------------------------------------------------
LM_DEV bool lm_box_hit(const LmCcRec& c, unsigned long long ub)
{
    int ux0 = (int)(ub & 0xffff), ux1 = (int)((ub >> 16) & 0xffff), uy0 = (int)((ub >> 32) & 0xffff), uy1 = (int)(ub >> 48);
    // inclusive boxes intersect  <=>  half-open intervals [min, max+1) overlap on both axes
    return c.min_x <= ux1 && ux0 <= c.max_x && c.min_y <= uy1 && uy0 <= c.max_y;
}

LM_DEV int lm_pixel_overlap(const LmCcRec& a, const LmCcRec& u, const uint32_t* __restrict__ crop, int lane)
{
    int y0 = a.min_y > u.min_y ? a.min_y : u.min_y;
    int y1 = a.max_y < u.max_y ? a.max_y : u.max_y;
    int awx0 = a.min_x >> 5, anw = (a.max_x >> 5) - awx0 + 1;
    int uwx0 = u.min_x >> 5, unw = (u.max_x >> 5) - uwx0 + 1;
    int wc0 = awx0 > uwx0 ? awx0 : uwx0;
    int wc1 = (a.max_x >> 5) < (u.max_x >> 5) ? (a.max_x >> 5) : (u.max_x >> 5);
    int nwc = wc1 - wc0 + 1;
    int total = nwc * (y1 - y0 + 1);
    int sum = 0;
    for (int idx = lane; idx < total; idx += 64) {
        int r = idx / nwc, j = idx - r * nwc;
        int y = y0 + r, wc = wc0 + j;
        unsigned wa = crop[a.crop_off + (unsigned long long)((y - a.min_y) * anw + (wc - awx0))];
        unsigned wu = crop[u.crop_off + (unsigned long long)((y - u.min_y) * unw + (wc - uwx0))];
        sum += __popc(wa & wu);
    }
    return lm_wave_sum(sum);
}

__global__ void __launch_bounds__(256) lm_k_match(const LmCcRec* __restrict__ cc, const uint32_t* __restrict__ crop,
                                                  const long long* __restrict__ frame_cc_off, int f,
                                                  const int32_t* __restrict__ uniq_cc, const unsigned long long* __restrict__ uniq_box16,
                                                  const int32_t* __restrict__ active, LmCounters* __restrict__ cnt,
                                                  int32_t* __restrict__ assign, double min_recall, double min_precision)
{
    const int lane = lm_lane();
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    if (cnt->error) return;     // a capacity error leaves frame_cc_off unwritten: do not touch it
    const long long c0 = frame_cc_off[f], c1 = frame_cc_off[f + 1];
    const int nA = cnt->n_active;
    for (long long c = c0 + wave; c < c1; c += nwaves) {
        const LmCcRec rec = cc[c];
        int found = -1;
        unsigned long long pairs = 0;
        for (int a0 = 0; a0 < nA; a0 += 64) {
            int a = a0 + lane;
            int u = (a < nA) ? active[a] : -1;
            bool hit = false;
            if (u >= 0) hit = lm_box_hit(rec, uniq_box16[u]);
            unsigned long long mask = __ballot(hit);
            pairs += (unsigned long long)__popcll(mask);
            while (found < 0 && mask) {
                int l = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                int uu = __shfl(u, l);
                const LmCcRec urec = cc[uniq_cc[uu]];
                int match = lm_pixel_overlap(rec, urec, crop, lane);
                double recall = (double)match / (double)rec.size;        // connected_component.py:239
                double precision = (double)match / (double)urec.size;    // :240
                if (recall >= min_recall && precision >= min_precision) found = uu;
            }
        }
        if (lane == 0) {
            assign[c] = found;
            if (pairs) atomicAdd(&cnt->tempo_count, pairs);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// M2: apply the frame's decisions: number new uniques in CC order, touch matched ones, retire.
// One block.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) lm_k_update(const LmCcRec* __restrict__ cc, const long long* __restrict__ frame_cc_off,
                                                    int f, int32_t* __restrict__ uniq_cc, unsigned long long* __restrict__ uniq_box16,
                                                    int32_t* __restrict__ uniq_last, int32_t* __restrict__ active,
                                                    LmCounters* __restrict__ cnt, int32_t* __restrict__ assign, int max_gap,
                                                    int cap_uniq)
{
    if (cnt->error) return;
    const long long c0 = frame_cc_off[f], c1 = frame_cc_off[f + 1];
    const int n = (int)(c1 - c0);
    const int nU = cnt->n_uniq, nA = cnt->n_active;
    __shared__ int s_overflow;
    if (threadIdx.x == 0) s_overflow = 0;
    __syncthreads();
    unsigned carry = 0;
    for (int base = 0; base < n; base += 1024) {
        int i = base + (int)threadIdx.x;
        int a = (i < n) ? assign[c0 + i] : 0;
        unsigned isnew = (i < n && a < 0) ? 1u : 0u;
        unsigned tot;
        unsigned ex = lm_block_excl_scan<1024>(isnew, &tot);
        if (i < n) {
            if (isnew) {
                long long idx = (long long)nU + carry + ex;
                if (idx < cap_uniq) {
                    const LmCcRec r = cc[c0 + i];
                    assign[c0 + i] = (int32_t)idx;
                    uniq_cc[idx] = (int32_t)(c0 + i);
                    uniq_box16[idx] = (unsigned long long)(unsigned short)r.min_x | ((unsigned long long)(unsigned short)r.max_x << 16) |
                                      ((unsigned long long)(unsigned short)r.min_y << 32) | ((unsigned long long)(unsigned short)r.max_y << 48);
                    uniq_last[idx] = f;
                    active[nA + carry + ex] = (int32_t)idx;
                } else {
                    s_overflow = 1;
                }
            } else {
                uniq_last[a] = f;     // several CCs may hit the same unique: same value
            }
        }
        carry += tot;
    }
    __syncthreads();
    const int nA2 = nA + (int)carry;
    unsigned kept = 0;
    if (f > 0) {        // the reference only retires inside the t > 0 branch (cc_stability_estimator.py:126-145)
        for (int base = 0; base < nA2; base += 1024) {
            int i = base + (int)threadIdx.x;
            int u = (i < nA2) ? active[i] : 0;
            unsigned keep = (i < nA2 && (f - uniq_last[u] < max_gap)) ? 1u : 0u;
            unsigned tot;
            unsigned ex = lm_block_excl_scan<1024>(keep, &tot);   // barriers inside: every read of this chunk precedes its writes
            if (keep) active[kept + ex] = u;
            kept += tot;
        }
    } else {
        kept = (unsigned)nA2;
    }
    if (threadIdx.x == 0) {
        if (s_overflow) {
            cnt->error = LM_ERR_CAPACITY;
        } else {
            cnt->n_uniq = nU + (int)carry;
            cnt->n_active = (int)kept;
            cnt->n_matched = f + 1;
        }
    }
}
