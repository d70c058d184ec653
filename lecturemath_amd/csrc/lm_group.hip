// lm_group.hip -- step 03 (CC grouping in space-time) over a finished device-resident stream.
//
// Replaces pre_ST3D_v3.0_03_cc_grouping.py:22-118 and the CCStabilityEstimator methods it calls
// (AccessMath/preprocessing/content/cc_stability_estimator.py, paths relative to
// /root/reference/ACCESS2021_release):
//   split_stable_cc_by_gaps :181-228, get_stable_cc_idxs :230-236            host (list bookkeeping)
//   compute_overlapping_stable_cc :245-306   box self-join + pixel overlaps    DEVICE (lm_k_selfjoin, lm_k_pair_overlap)
//                                            recall/precision/time window      host (float64, same expressions)
//   compute_groups :308-413                  order-dependent list merging      host (sequential by definition)
//   compute_groups_temporal_information :415-444, compute_conflicting_groups :446-500   host
//   compute_group_images :575-636            accumulate crops x counts, /max, >= thr   DEVICE (lm_k_gimg_max, lm_k_gimg_write)
//   frames_from_groups :638-681              uint8 wrap-around compositing of channel 0   DEVICE (lm_k_render_frames)
// The dead work of the reference is not reproduced: rebuilt_binary_images (result unused,
// 03_cc_grouping.py:41) and the channel-1/2 painting that is never encoded (:661-671,678).
#include "lm_stream.h"

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <unordered_map>
#include <vector>

// ------------------------------------------------------------------------------------------------
// G1: box self-join of the stable uniques (pairs k1 < k2 in the stable list, i.e. idx1 < idx2).
// Thread = one "column" box k2, block tile = 256 "row" boxes k1 in LDS.  Count / reserve / fill.
// ------------------------------------------------------------------------------------------------
#define LM_SJ_TILE 256

__global__ void __launch_bounds__(256) lm_k_selfjoin(const unsigned long long* __restrict__ box, int n, int* __restrict__ n_pairs,
                                                     int2* __restrict__ pairs, int cap_pairs)
{
    __shared__ unsigned long long s_box[LM_SJ_TILE];
    __shared__ int s_base;
    unsigned mine = 0, off = 0;
    for (int pass = 0; pass < 2; pass++) {
        for (int rt = blockIdx.y * LM_SJ_TILE; rt < n; rt += gridDim.y * LM_SJ_TILE) {
            const int tile = (n - rt < LM_SJ_TILE) ? n - rt : LM_SJ_TILE;
            __syncthreads();
            if ((int)threadIdx.x < tile) s_box[threadIdx.x] = box[rt + threadIdx.x];
            __syncthreads();
            for (int c0 = blockIdx.x * 256; c0 < n; c0 += gridDim.x * 256) {
                if (c0 + 255 <= rt) continue;               // every column of the chunk <= every row of the tile
                const int k2 = c0 + (int)threadIdx.x;
                if (k2 >= n) continue;
                const unsigned long long b2 = box[k2];
                int lim = k2 - rt;                          // rows k1 = rt + j with k1 < k2
                if (lim > tile) lim = tile;
                for (int j = 0; j < lim; j++) {
                    if (!lm_box_hit_packed(s_box[j], b2)) continue;
                    if (pass == 0) {
                        mine++;
                    } else {
                        if ((int)off < cap_pairs) pairs[off] = make_int2(rt + j, k2);
                        off++;
                    }
                }
            }
        }
        if (pass == 0) {
            unsigned tot;
            unsigned ex = lm_block_excl_scan<256>(mine, &tot);
            if (threadIdx.x == 0) s_base = tot ? atomicAdd(n_pairs, (int)tot) : 0;
            __syncthreads();
            if (tot == 0) return;
            off = (unsigned)s_base + ex;
        }
    }
}

// G2: pixel overlap of every pair; 16 lanes share one pair.
__global__ void __launch_bounds__(256) lm_k_pair_overlap(const LmCcRec* __restrict__ cc, const uint32_t* __restrict__ crop,
                                                         const int32_t* __restrict__ su_cc, const int2* __restrict__ pairs, int np,
                                                         int32_t* __restrict__ match)
{
    const int sub = (int)(threadIdx.x & 15);
    const int group = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 4);
    const int ngroups = (int)((gridDim.x * blockDim.x) >> 4);
    const int np_pad = (np + 3) & ~3;
    for (int p = group; p < np_pad; p += ngroups) {
        int m = 0;
        const bool live = p < np;
        if (live) {
            const int2 pr = pairs[p];
            const LmCcRec a = cc[su_cc[pr.x]], u = cc[su_cc[pr.y]];
            const LmIsect is = lm_isect(a, u);
            m = lm_overlap_words(a, u, is, crop, sub, 16);
        }
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) m += __shfl_xor(m, d, 16);
        if (live && sub == 0) match[p] = m;
    }
}

// ------------------------------------------------------------------------------------------------
// G3: group images.  Item = (group, age segment); its image covers the group's box.  For every item the
// host lists the members present in the segment with the number of frame entries they have in it.
// Work unit = (item, 64x64 tile of its box): accumulate (crop bit) * count into an LDS int32 tile.
//   pass A (lm_k_gimg_max)    per-item maximum of the accumulated mask; items of ONE tile are thresholded and written here
//   pass B (lm_k_gimg_write)  items of several tiles: recompute and write  ((double)v / (double)max >= thr) ? 255 : 0
// ------------------------------------------------------------------------------------------------
struct LmGimgItem {
    int32_t x0, y0, w, h;        // group box origin and size
    long long img_off;           // byte offset of the item's (h x w) uint8 image
    long long bits_off;          // word offset of the same image as bit rows (ceil(w / 32) words per row), what the renderer reads
};
struct LmGimgMember { int32_t cc; int32_t count; };
// work unit = one 64 x 64 tile of an item's box with the members whose boxes touch it (a group box is mostly empty and a
// member is a glyph-sized CC: listing the members per tile keeps the work linear in the members, not members x tiles)
struct LmGimgUnit { int32_t item; int16_t tx, ty; uint32_t mem_off, mem_cnt; };

#define LM_GT 64   // tile side

LM_DEV void lm_gimg_accumulate(const LmGimgItem& it, const LmGimgUnit& un, const LmGimgMember* __restrict__ members,
                               const LmCcRec* __restrict__ cc, const uint32_t* __restrict__ crop, int* s_mask)
{
    // tile covers x in [X0, X0+64), y in [Y0, Y0+64) of the frame
    const int X0 = it.x0 + un.tx * LM_GT, Y0 = it.y0 + un.ty * LM_GT;
    for (int i = threadIdx.x; i < LM_GT * LM_GT; i += blockDim.x) s_mask[i] = 0;
    __syncthreads();
    const int wave = (int)(threadIdx.x >> 6), nwaves = (int)(blockDim.x >> 6), lane = lm_lane();
    for (int m = wave; m < (int)un.mem_cnt; m += nwaves) {
        const LmGimgMember mem = members[un.mem_off + m];
        const LmCcRec r = cc[mem.cc];
        if (r.max_x < X0 || r.min_x >= X0 + LM_GT || r.max_y < Y0 || r.min_y >= Y0 + LM_GT) continue;
        const int wx0 = r.min_x >> 5, nw = (r.max_x >> 5) - wx0 + 1;
        const int ya = r.min_y > Y0 ? r.min_y : Y0;
        const int yb = r.max_y < Y0 + LM_GT - 1 ? r.max_y : Y0 + LM_GT - 1;
        // only the crop words under the tile's 64 columns (a large component's rows are tens of words wide)
        const int ja = (X0 >> 5) > wx0 ? (X0 >> 5) - wx0 : 0;
        const int jb = ((X0 + LM_GT - 1) >> 5) - wx0 < nw - 1 ? ((X0 + LM_GT - 1) >> 5) - wx0 : nw - 1;
        const int tw = jb - ja + 1;
        const int total = tw * (yb - ya + 1);
        for (int idx = lane; idx < total; idx += 64) {
            const int rr = idx / tw, j = ja + (idx - rr * tw);
            const int y = ya + rr;
            unsigned wbits = crop[r.crop_off + (unsigned long long)((y - r.min_y) * nw + j)];
            const int xw = (wx0 + j) * 32;
            while (wbits) {
                const int b = __ffs((int)wbits) - 1;
                wbits &= wbits - 1;
                const int x = xw + b;
                if (x >= X0 && x < X0 + LM_GT) atomicAdd(&s_mask[(y - Y0) * LM_GT + (x - X0)], mem.count);
            }
        }
    }
    __syncthreads();
}

// The tile's pixels with (double)v / mx >= thr as bit rows of the item (two words per 64-px tile row).  s_mask holds the tile's sums.
LM_DEV void lm_gimg_write_tile(const LmGimgItem& it, const LmGimgUnit& un, const int* s_mask, int item_max, double thr, uint32_t* __restrict__ bits)
{
    const double mx = (double)item_max;
    // (double)v / mx >= thr is monotone in v: one exact integer bound per tile instead of a float64 division per pixel
    int vmin = (int)(thr * mx);
    if (vmin < 0) vmin = 0;
    while (vmin > 0 && (double)(vmin - 1) / mx >= thr) vmin--;
    while ((double)vmin / mx < thr) vmin++;
    const int tw = (it.w - un.tx * LM_GT < LM_GT) ? it.w - un.tx * LM_GT : LM_GT;
    const int th = (it.h - un.ty * LM_GT < LM_GT) ? it.h - un.ty * LM_GT : LM_GT;
    const int bw = (it.w + 31) >> 5;
    // a lane per pixel of a tile row (conflict-free LDS reads), the ballot is the row's two words
    const int wave = (int)(threadIdx.x >> 6), nwaves = (int)(blockDim.x >> 6), lane = lm_lane();
    for (int yy = wave; yy < LM_GT; yy += nwaves) {         // every wave runs all its trips: ballots need whole waves
        const bool on = yy < th && lane < tw && s_mask[yy * LM_GT + lane] >= vmin;
        const unsigned long long m = __ballot(on);
        if (yy < th && lane < 2 && un.tx * 2 + lane < bw)
            bits[it.bits_off + (long long)(un.ty * LM_GT + yy) * bw + un.tx * 2 + lane] = (unsigned)(m >> (32 * lane));
    }
}

// ---- tiles with ONE member (most of them: 1.7 members per visited tile on the 10,000-frame stream): the tile's sums are count x (the
// member's pixels), so there is nothing to accumulate -- the maximum is `count` if the member has a pixel in the tile, and the image is the
// member's bit rows, shifted from the crop's absolute 32-px column alignment to the tile's, when count passes the threshold.
// does the member own a pixel inside the tile?  (every thread calls; result through s_hit)
LM_DEV void lm_gimg_single_hit(const LmGimgItem& it, const LmGimgUnit& un, const LmCcRec& r, const uint32_t* __restrict__ crop, int* s_hit)
{
    const int X0 = it.x0 + un.tx * LM_GT, Y0 = it.y0 + un.ty * LM_GT;
    const int wx0 = r.min_x >> 5, nw = (r.max_x >> 5) - wx0 + 1;
    const int ya = r.min_y > Y0 ? r.min_y : Y0, yb = r.max_y < Y0 + LM_GT - 1 ? r.max_y : Y0 + LM_GT - 1;
    const int ja = (X0 >> 5) > wx0 ? (X0 >> 5) - wx0 : 0;
    const int jb = ((X0 + LM_GT - 1) >> 5) - wx0 < nw - 1 ? ((X0 + LM_GT - 1) >> 5) - wx0 : nw - 1;
    const int tw = jb - ja + 1, total = (tw > 0 && yb >= ya) ? tw * (yb - ya + 1) : 0;
    bool hit = false;
    for (int idx = threadIdx.x; idx < total && !hit; idx += blockDim.x) {
        const int rr = idx / tw, j = ja + (idx - rr * tw);
        unsigned wbits = crop[r.crop_off + (unsigned long long)((ya + rr - r.min_y) * nw + j)];
        const int xw = (wx0 + j) * 32;              // frame column of the word's bit 0: keep the bits with X0 <= x < X0 + 64
        if (xw < X0) wbits &= (X0 - xw >= 32) ? 0u : (0xffffffffu << (X0 - xw));
        if (xw + 31 >= X0 + LM_GT) wbits &= (xw >= X0 + LM_GT) ? 0u : (0xffffffffu >> (xw + 32 - (X0 + LM_GT)));
        hit = wbits != 0u;
    }
    if (hit) *s_hit = 1;
}

// the member's pixels as the tile's bit rows (the words outside the member's rows stay cleared)
LM_DEV void lm_gimg_single_write(const LmGimgItem& it, const LmGimgUnit& un, const LmCcRec& r, const uint32_t* __restrict__ crop, uint32_t* __restrict__ bits)
{
    const int X0 = it.x0 + un.tx * LM_GT, Y0 = it.y0 + un.ty * LM_GT;
    const int wx0 = r.min_x >> 5, nw = (r.max_x >> 5) - wx0 + 1;
    const int tw = (it.w - un.tx * LM_GT < LM_GT) ? it.w - un.tx * LM_GT : LM_GT;
    const int th = (it.h - un.ty * LM_GT < LM_GT) ? it.h - un.ty * LM_GT : LM_GT;
    const int bw = (it.w + 31) >> 5;
    for (int t = threadIdx.x; t < 2 * LM_GT; t += blockDim.x) {
        const int yy = t >> 1, k = t & 1, y = Y0 + yy;
        if (yy >= th || un.tx * 2 + k >= bw || y < r.min_y || y > r.max_y) continue;
        const int xa = X0 + 32 * k;                 // frame column of the output word's bit 0
        const int c0 = (xa >> 5) - wx0, sh = xa & 31;
        const uint32_t* row = crop + r.crop_off + (unsigned long long)(y - r.min_y) * nw;
        const unsigned lo = (c0 >= 0 && c0 < nw) ? row[c0] : 0u;
        const unsigned hi = (sh && c0 + 1 >= 0 && c0 + 1 < nw) ? row[c0 + 1] : 0u;
        unsigned w = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
        const int valid = tw - 32 * k;              // columns of this word inside the item
        if (valid < 32) w &= valid <= 0 ? 0u : ((1u << valid) - 1u);
        if (w) bits[it.bits_off + (long long)(un.ty * LM_GT + yy) * bw + un.tx * 2 + k] = w;
    }
}

// the exact integer bound of ((double)v / mx >= thr), as lm_gimg_write_tile computes it
LM_DEV int lm_gimg_vmin(int item_max, double thr)
{
    const double mx = (double)item_max;
    int vmin = (int)(thr * mx);
    if (vmin < 0) vmin = 0;
    while (vmin > 0 && (double)(vmin - 1) / mx >= thr) vmin--;
    while ((double)vmin / mx < thr) vmin++;
    return vmin;
}

// an item that is a single 64 x 64 tile: its maximum is the tile's, known to the workgroup that accumulates the tile
LM_DEV bool lm_gimg_single_tile(const LmGimgItem& it) { return it.w <= LM_GT && it.h <= LM_GT; }

// Pass 1: per-item maximum of the accumulated member masks (the reference's mask.max(), :628).  Items of one tile -- glyph-sized
// groups, most of them -- are finished here: threshold and bit rows straight from the sums in LDS (round 2 accumulated every tile
// twice, once for the maximum and once for the image: 2 x 17.8 ms per 10,000-frame stream).
__global__ void __launch_bounds__(256) lm_k_gimg_max(const LmGimgItem* __restrict__ items, const LmGimgUnit* __restrict__ units,
                                                     const unsigned* __restrict__ n_units_p, const LmGimgMember* __restrict__ members,
                                                     const LmCcRec* __restrict__ cc, const uint32_t* __restrict__ crop,
                                                     int32_t* __restrict__ item_max, double thr, uint32_t* __restrict__ bits)
{
    __shared__ int s_mask[LM_GT * LM_GT];
    __shared__ int s_max;
    const int n_units = (int)*n_units_p;        // the unit list is compacted on the device (lm_k_gimg_units)
    for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
        const LmGimgUnit un = units[u];
        const LmGimgItem it = items[un.item];
        if (threadIdx.x == 0) s_max = 0;
        if (un.mem_cnt == 1 && thr > 0.0) {             // one member: its count, if it owns a pixel of the tile (block-uniform branch)
            const LmGimgMember mem = members[un.mem_off];
            const LmCcRec r = cc[mem.cc];
            __syncthreads();
            lm_gimg_single_hit(it, un, r, crop, &s_max);
            __syncthreads();
            const int tile_max = s_max ? mem.count : 0;
            if (threadIdx.x == 0 && tile_max) atomicMax(&item_max[un.item], tile_max);
            if (lm_gimg_single_tile(it) && tile_max && tile_max >= lm_gimg_vmin(tile_max, thr)) lm_gimg_single_write(it, un, r, crop, bits);
            __syncthreads();
            continue;
        }
        lm_gimg_accumulate(it, un, members, cc, crop, s_mask);
        int mx = 0;
        for (int i = threadIdx.x; i < LM_GT * LM_GT; i += blockDim.x) mx = s_mask[i] > mx ? s_mask[i] : mx;
        if (mx) atomicMax(&s_max, mx);
        __syncthreads();
        const int tile_max = s_max;
        if (threadIdx.x == 0 && tile_max) atomicMax(&item_max[un.item], tile_max);
        if (lm_gimg_single_tile(it) && tile_max) lm_gimg_write_tile(it, un, s_mask, tile_max, thr, bits);
        __syncthreads();
    }
}

// The segment image ((mask / mask.max()) >= thr, float64 like numpy, :630; max >= 1 by construction) is kept as bit rows: the
// renderer reads bits, and the reference's uint8 arrays are expanded from them on demand (lm_k_gimg_expand).  Tiles that no
// member touches are never visited: their words were cleared before the launch (all zero <=> below any positive threshold).
__global__ void __launch_bounds__(256) lm_k_gimg_write(const LmGimgItem* __restrict__ items, const LmGimgUnit* __restrict__ units,
                                                       const unsigned* __restrict__ n_units_p, const LmGimgMember* __restrict__ members,
                                                       const LmCcRec* __restrict__ cc, const uint32_t* __restrict__ crop,
                                                       const int32_t* __restrict__ item_max, double thr, uint32_t* __restrict__ bits)
{
    __shared__ int s_mask[LM_GT * LM_GT];
    const int n_units = (int)*n_units_p;
    for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
        const LmGimgUnit un = units[u];
        const LmGimgItem it = items[un.item];
        if (lm_gimg_single_tile(it) && item_max[un.item] > 0) continue;     // finished by lm_k_gimg_max (block-uniform)
        if (un.mem_cnt == 1 && thr > 0.0 && item_max[un.item] > 0) {        // one member: copy its bit rows if its count passes
            const LmGimgMember mem = members[un.mem_off];
            if (mem.count >= lm_gimg_vmin(item_max[un.item], thr)) lm_gimg_single_write(it, un, cc[mem.cc], crop, bits);
            continue;
        }
        lm_gimg_accumulate(it, un, members, cc, crop, s_mask);
        lm_gimg_write_tile(it, un, s_mask, item_max[un.item], thr, bits);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// G4: frames_from_groups, channel 0: canvas[y, x] = sum over the frame's live groups of their current
// segment image, uint8 wrap-around.  Every contribution is 0 or 255 (== -1 mod 256), so the pixel is
// (-k) mod 256 with k = number of contributing segment pixels.  Block = (frame, 64-row x 256-col tile),
// k counted in LDS.
// ------------------------------------------------------------------------------------------------
struct LmRenderItem { int32_t x0, y0, w, h; long long bits_off; };

#define LM_RT_ROWS 64
#define LM_RT_COLS 256
#define LM_RT_PITCH (LM_RT_COLS / 32 + 1)      // words per tile row of one counter plane (+1: rows of a narrow item fall into distinct banks)
#define LM_RT_PLANE (LM_RT_ROWS * LM_RT_PITCH)
#define LM_RT_MAXHIT 96     // items of one frame that touch one tile and are painted cooperatively (more: painted by their finder)
#define LM_RT_WPL 4         // tile words per lane in flight while painting

// The reference adds every group image (0 / 255) into a uint8 frame (:674-678), so a pixel covered by k images holds -k mod 256.
// k is kept bit-sliced: plane p holds bit p of k for 32 pixels per word.  Adding an image word = toggle plane 0, carry the bits
// that went 1 -> 0 into plane 1, ... (the carry out of plane 7 is the mod 256).  Concurrent adds commute: a bit that is toggled
// N times from b generates floor((b + N) / 2) carries whatever the order.  One LDS atomic per 32 pixels where images do not
// overlap (nearly everywhere), instead of one per set pixel.
LM_DEV void lm_render_paint(const LmRenderItem& it, int X0, int Y0, const uint32_t* __restrict__ bits, unsigned* s_pl, int lane, int nl)
{
    const int xa = it.x0 > X0 ? it.x0 : X0, xb = (it.x0 + it.w < X0 + LM_RT_COLS) ? it.x0 + it.w : X0 + LM_RT_COLS;
    const int ya = it.y0 > Y0 ? it.y0 : Y0, yb = (it.y0 + it.h < Y0 + LM_RT_ROWS) ? it.y0 + it.h : Y0 + LM_RT_ROWS;
    const int bw = (it.w + 31) >> 5;
    const int jlo = (xa - X0) >> 5, nw = ((xb - 1 - X0) >> 5) - jlo + 1;        // tile words the item touches per row
    const int total = nw * (yb - ya);
    const float inv = 1.0f / (float)nw;
    const uint32_t* src = bits + it.bits_off + (long long)(ya - it.y0) * bw;
    for (int idx0 = lane; idx0 < total; idx0 += nl * LM_RT_WPL) {
        unsigned lo[LM_RT_WPL], hi[LM_RT_WPL];
        int col[LM_RT_WPL], slot[LM_RT_WPL];
#pragma unroll
        for (int u = 0; u < LM_RT_WPL; u++) {
            const int idx = idx0 + u * nl;
            int yy = (int)((float)idx * inv);
            int j = idx - yy * nw;
            if (j < 0) { yy--; j += nw; }
            if (j >= nw) { yy++; j -= nw; }
            col[u] = X0 + 32 * (jlo + j);                   // frame column of the tile word's bit 0
            const int d = col[u] - it.x0;                   // ... and the image column under it (> -32)
            const int sw = d >> 5;                          // floor
            const uint32_t* r = src + (long long)yy * bw;
            const bool on = idx < total;
            lo[u] = (on && sw >= 0 && sw < bw) ? r[sw] : 0u;
            hi[u] = (on && (d & 31) && sw + 1 < bw) ? r[sw + 1] : 0u;
            slot[u] = (ya - Y0 + yy) * LM_RT_PITCH + jlo + j;
        }
#pragma unroll
        for (int u = 0; u < LM_RT_WPL; u++) {
            const int sh = (col[u] - it.x0) & 31;
            unsigned c = sh ? ((lo[u] >> sh) | (hi[u] << (32 - sh))) : lo[u];
            if (col[u] < xa) c &= 0xffffffffu << (xa - col[u]);             // clip to the tile / item intersection in x
            if (col[u] + 32 > xb) c &= 0xffffffffu >> (col[u] + 32 - xb);
            for (int p = 0; p < 8 && c; p++) c &= atomicXor(&s_pl[p * LM_RT_PLANE + slot[u]], c);
        }
    }
}

// one LmRenderItem per (frame, live group): the group's current segment image (:650-652 `while ages[ptr+1] < f: ptr += 1`)
__global__ void __launch_bounds__(256) lm_k_render_items(const long long* __restrict__ frame_item_off, int F, const int32_t* __restrict__ gpf,
                                                         const int32_t* __restrict__ ages, const int32_t* __restrict__ ages_off,
                                                         const int32_t* __restrict__ bounds, const int64_t* __restrict__ gitem_first,
                                                         const int64_t* __restrict__ gbits_off, LmRenderItem* __restrict__ out)
{
    // a workgroup per frame (its items are contiguous); round 2 dealt the items to threads and found each one's frame by a 14-step
    // binary search over frame_item_off[] in L2 (3 ms for the 10,000-frame stream's four million items)
    for (int f = blockIdx.x; f < F; f += gridDim.x)
    for (long long i = frame_item_off[f] + threadIdx.x; i < frame_item_off[f + 1]; i += blockDim.x) {
    const int gi = gpf[i];
    const int32_t* a = ages + ages_off[gi];
    const int na = ages_off[gi + 1] - ages_off[gi];
    LmRenderItem ri;
    ri.x0 = bounds[gi * 4 + 0]; ri.y0 = bounds[gi * 4 + 2];
    ri.w = bounds[gi * 4 + 1] - ri.x0 + 1; ri.h = bounds[gi * 4 + 3] - ri.y0 + 1;
    ri.bits_off = 0;
    if (na < 2) {               // no segment image (the reference would raise IndexError at :650 here): an item that hits no tile
        ri.w = 0; ri.h = 0;
    } else {
        // the reference walks `while ages[ptr + 1] < f: ptr += 1` (bounded by the last segment); ages ascend, so that is the number of
        // k in [1, na - 2] with a[k] < f -- by bisection (a group that was edited fifty times made the walk fifty dependent loads)
        int lo = 0, hi = na - 1;        // invariant: a[k] < f for 1 <= k <= lo, a[k] >= f for hi <= k <= na - 2 (hi == na - 1: none known)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a[mid] < f) lo = mid; else hi = mid;
        }
        const int sidx = lo;
        ri.bits_off = gbits_off[gitem_first[gi] + sidx];
    }
    out[i] = ri;
    }
}

__global__ void __launch_bounds__(256) lm_k_render_frames(const long long* __restrict__ frame_item_off,
                                                          const LmRenderItem* __restrict__ items, const uint32_t* __restrict__ bits,
                                                          int first_frame, int W, int H, uint8_t* __restrict__ out)
{
    __shared__ unsigned s_pl[8 * LM_RT_PLANE];      // bit-sliced k per pixel (see lm_render_paint)
    __shared__ LmRenderItem s_hit[LM_RT_MAXHIT];
    __shared__ unsigned s_pre[LM_RT_MAXHIT + 1];
    __shared__ int s_nhit;
    const int f = first_frame + blockIdx.z;
    const int X0 = blockIdx.x * LM_RT_COLS, Y0 = blockIdx.y * LM_RT_ROWS;
    for (int i = threadIdx.x; i < 8 * LM_RT_PLANE; i += blockDim.x) s_pl[i] = 0;
    if (threadIdx.x == 0) s_nhit = 0;
    __syncthreads();
    const long long i0 = frame_item_off[f], i1 = frame_item_off[f + 1];
    // screen the frame's items (one per thread and trip), collect the few that touch the tile
#pragma unroll 2
    for (long long ib = i0 + threadIdx.x; ib < i1; ib += blockDim.x) {
        const LmRenderItem mine = items[ib];
        if (mine.x0 < X0 + LM_RT_COLS && mine.x0 + mine.w > X0 && mine.y0 < Y0 + LM_RT_ROWS && mine.y0 + mine.h > Y0) {
            const int slot = atomicAdd(&s_nhit, 1);
            if (slot < LM_RT_MAXHIT) s_hit[slot] = mine;
            else lm_render_paint(mine, X0, Y0, bits, s_pl, 0, 1);      // crowded tile: the finder paints it alone
        }
    }
    __syncthreads();
    const int nhit = s_nhit < LM_RT_MAXHIT ? s_nhit : LM_RT_MAXHIT;
    // Round 3: the tile words of ALL listed items are dealt to the threads as one flat list (an item per wave left a wave walking
    // through its items one after the other, two memory latencies each: a tile under a large component and a dozen glyphs took
    // eight).  s_pre[h] = tile words of the items before h.
    {
        const int lane = lm_lane();
        if (threadIdx.x < 64) {
            unsigned carry = 0;
            for (int h0 = 0; h0 < nhit; h0 += 64) {
                const int h = h0 + lane;
                unsigned words = 0;
                if (h < nhit) {
                    const LmRenderItem it = s_hit[h];
                    const int xa = it.x0 > X0 ? it.x0 : X0, xb = (it.x0 + it.w < X0 + LM_RT_COLS) ? it.x0 + it.w : X0 + LM_RT_COLS;
                    const int ya = it.y0 > Y0 ? it.y0 : Y0, yb = (it.y0 + it.h < Y0 + LM_RT_ROWS) ? it.y0 + it.h : Y0 + LM_RT_ROWS;
                    const int jlo = (xa - X0) >> 5, nw = ((xb - 1 - X0) >> 5) - jlo + 1;
                    words = (nw > 0 && yb > ya) ? (unsigned)(nw * (yb - ya)) : 0u;
                }
                const unsigned incl = lm_wave_incl_scan(words);
                if (h < nhit) s_pre[h] = carry + incl - words;
                carry += (unsigned)__shfl((int)incl, 63);
            }
            if (lane == 0) s_pre[nhit] = carry;
        }
    }
    __syncthreads();
    {
        const unsigned total = s_pre[nhit];
        for (unsigned g0 = threadIdx.x; g0 < total; g0 += blockDim.x * LM_RT_WPL) {
            unsigned lo[LM_RT_WPL], hi[LM_RT_WPL];
            int col[LM_RT_WPL], slot[LM_RT_WPL], xa_[LM_RT_WPL], xb_[LM_RT_WPL], x0_[LM_RT_WPL];
#pragma unroll
            for (int u = 0; u < LM_RT_WPL; u++) {
                const unsigned g = g0 + (unsigned)u * blockDim.x;
                lo[u] = 0; hi[u] = 0; col[u] = 0; slot[u] = 0; xa_[u] = 0; xb_[u] = 0; x0_[u] = 0;
                if (g < total) {
                    int a = 0, b2 = nhit;            // largest h with s_pre[h] <= g
                    while (b2 - a > 1) {
                        const int mid = (a + b2) >> 1;
                        if (s_pre[mid] <= g) a = mid; else b2 = mid;
                    }
                    const LmRenderItem it = s_hit[a];
                    const int xa = it.x0 > X0 ? it.x0 : X0, xb = (it.x0 + it.w < X0 + LM_RT_COLS) ? it.x0 + it.w : X0 + LM_RT_COLS;
                    const int ya = it.y0 > Y0 ? it.y0 : Y0;
                    const int bw = (it.w + 31) >> 5;
                    const int jlo = (xa - X0) >> 5, nw = ((xb - 1 - X0) >> 5) - jlo + 1;
                    const int idx = (int)(g - s_pre[a]);
                    int yy = (int)((float)idx * (1.0f / (float)nw));
                    int j = idx - yy * nw;
                    if (j < 0) { yy--; j += nw; }
                    if (j >= nw) { yy++; j -= nw; }
                    col[u] = X0 + 32 * (jlo + j);                   // frame column of the tile word's bit 0
                    const int d = col[u] - it.x0;                   // ... and the image column under it (> -32)
                    const int sw = d >> 5;                          // floor
                    const uint32_t* r = bits + it.bits_off + (long long)(ya - it.y0 + yy) * bw;
                    lo[u] = (sw >= 0 && sw < bw) ? r[sw] : 0u;
                    hi[u] = ((d & 31) && sw + 1 < bw) ? r[sw + 1] : 0u;
                    slot[u] = (ya - Y0 + yy) * LM_RT_PITCH + jlo + j;
                    xa_[u] = xa; xb_[u] = xb; x0_[u] = it.x0;
                }
            }
#pragma unroll
            for (int u = 0; u < LM_RT_WPL; u++) {
                if (g0 + (unsigned)u * blockDim.x >= total) continue;      // (its zeroed descriptors would shift by 32)
                const int sh = (col[u] - x0_[u]) & 31;
                unsigned c = sh ? ((lo[u] >> sh) | (hi[u] << (32 - sh))) : lo[u];
                if (col[u] < xa_[u]) c &= 0xffffffffu << (xa_[u] - col[u]);             // clip to the tile / item intersection in x
                if (col[u] + 32 > xb_[u]) c &= 0xffffffffu >> (col[u] + 32 - xb_[u]);
                for (int p = 0; p < 8 && c; p++) c &= atomicXor(&s_pl[p * LM_RT_PLANE + slot[u]], c);
            }
        }
    }
    __syncthreads();
    uint8_t* dst = out + (long long)blockIdx.z * W * H;
    const bool vec = ((W & 15) == 0) && ((((uintptr_t)out) & 15) == 0);
    // 16 pixels (half a plane word) -> one 16-byte store
    for (int i = threadIdx.x; i < LM_RT_ROWS * LM_RT_COLS / 16; i += blockDim.x) {
        const int yy = i / (LM_RT_COLS / 16), xq = i - yy * (LM_RT_COLS / 16);
        const int y = Y0 + yy, x = X0 + xq * 16;
        if (y >= H || x >= W) continue;
        unsigned pl[8], upper = 0;
#pragma unroll
        for (int p = 0; p < 8; p++) {
            pl[p] = (s_pl[p * LM_RT_PLANE + yy * LM_RT_PITCH + (xq >> 1)] >> (16 * (xq & 1))) & 0xffffu;
            if (p) upper |= pl[p];
        }
        unsigned o[4];
        if (!upper) {       // k <= 1 everywhere: bit -> 0x00 / 0xff (the multiply spreads four bits over four bytes)
#pragma unroll
            for (int q = 0; q < 4; q++) o[q] = ((((pl[0] >> (4 * q)) & 15u) * 0x00204081u) & 0x01010101u) * 0xffu;
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                o[q] = 0;
                for (int t = 0; t < 4; t++) {
                    unsigned k = 0;
                    for (int p = 0; p < 8; p++) k |= ((pl[p] >> (4 * q + t)) & 1u) << p;
                    o[q] |= ((0u - k) & 0xffu) << (8 * t);
                }
            }
        }
        uint8_t* d = dst + (long long)y * W + x;
        if (vec) {
            *(uint4*)d = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
            for (int k = 0; k < 16 && x + k < W; k++) d[k] = (uint8_t)(o[k >> 2] >> (8 * (k & 3)));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// G5: per-frame pixel sums of uint8 frames (step 04, VideoSegmenter.compute_binary_sums,
// AccessMath/preprocessing/content/video_segmenter.py:22-28: `binary.sum() / 255`; the division stays on the host in
// float64).  Exact integer sums: 16-B loads, v_sad_u8 adds four bytes per instruction, one 64-bit atomic per workgroup.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) lm_k_frame_sums(const uint8_t* __restrict__ frames, long long px, unsigned long long* __restrict__ sums)
{
    const uint8_t* f = frames + (long long)blockIdx.y * px;
    const bool vec = ((((uintptr_t)f) & 15) == 0);
    unsigned long long acc = 0;
    const long long n16 = vec ? (px >> 4) : 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) {
        const uint4 v = *(const uint4*)(f + (i << 4));
        unsigned s4 = 0;
#if LM_HIP_EMULATED
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        for (int k = 0; k < 4; k++) s4 += (w[k] & 0xffu) + ((w[k] >> 8) & 0xffu) + ((w[k] >> 16) & 0xffu) + (w[k] >> 24);
#else
        s4 = __builtin_amdgcn_sad_u8(v.x, 0u, s4); s4 = __builtin_amdgcn_sad_u8(v.y, 0u, s4);
        s4 = __builtin_amdgcn_sad_u8(v.z, 0u, s4); s4 = __builtin_amdgcn_sad_u8(v.w, 0u, s4);
#endif
        acc += s4;
    }
    for (long long i = (n16 << 4) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < px; i += (long long)gridDim.x * blockDim.x) acc += f[i];
    __shared__ unsigned long long s_acc;
    if (threadIdx.x == 0) s_acc = 0;
    __syncthreads();
    // wave reduction of the 64-bit partial sums (two 32-bit halves through the 32-bit wave sum)
    const unsigned lo = lm_wave_sum((unsigned)(acc & 0xffffffu)), mid = lm_wave_sum((unsigned)((acc >> 24) & 0xffffffu));
    const unsigned hi = lm_wave_sum((unsigned)(acc >> 48));
    if (lm_lane() == 0) atomicAdd(&s_acc, (unsigned long long)lo + ((unsigned long long)mid << 24) + ((unsigned long long)hi << 48));
    __syncthreads();
    if (threadIdx.x == 0 && s_acc) atomicAdd(&sums[blockIdx.y], s_acc);
}

// ------------------------------------------------------------------------------------------------
// G6 (step 05): which pairs of {0,255} images placed in the frame share an ink pixel?  Replaces the all-pairs
// ConnectedComponent.getOverlapFMeasure loop of CCStabilityEstimator.compute_overlapping_CC_groups
// (cc_stability_estimator.py:696-714; "recall > 0 or precision > 0" <=> one common pixel) and the incompatibility tests of
// KeyframeExtractor.GenerateFromST3DForIntervals (keyframe_extractor.py:85-91): box self-join, then a bit test per pair.
// Images are packed to bit rows relative to their own x0 (ceil(w / 32) words per row).
// ------------------------------------------------------------------------------------------------
struct LmBitImage { int32_t x0, y0, w, h; long long src_off, bits_off; };

__global__ void __launch_bounds__(256) lm_k_img_pack(const LmBitImage* __restrict__ items, int n, const uint8_t* __restrict__ src,
                                                     uint32_t* __restrict__ bits)
{
    for (int it = blockIdx.y; it < n; it += gridDim.y) {
        const LmBitImage im = items[it];
        const int bw = (im.w + 31) >> 5;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < im.h * bw; i += gridDim.x * blockDim.x) {
            const int y = i / bw, k = i - y * bw;
            const uint8_t* p = src + im.src_off + (long long)y * im.w + k * 32;
            const int lim = (im.w - k * 32 < 32) ? im.w - k * 32 : 32;
            unsigned word = 0;
            for (int b = 0; b < lim; b++) word |= (unsigned)(p[b] != 0) << b;
            bits[im.bits_off + i] = word;
        }
    }
}

// 32 bits of row `y` (image coordinates) starting at column `x` (may start anywhere; bits beyond the row are zero)
LM_DEV unsigned lm_bitrow32(const uint32_t* __restrict__ bits, const LmBitImage& im, int bw, int y, int x)
{
    const uint32_t* row = bits + im.bits_off + (long long)y * bw;
    const int k = x >> 5, off = x & 31;
    unsigned lo = (k < bw) ? row[k] : 0u;
    unsigned hi = (off && k + 1 < bw) ? row[k + 1] : 0u;
    return off ? ((lo >> off) | (hi << (32 - off))) : lo;
}

__global__ void __launch_bounds__(256) lm_k_bitimg_pair_any(const LmBitImage* __restrict__ items, const uint32_t* __restrict__ bits,
                                                            const int2* __restrict__ pairs, int np, int32_t* __restrict__ hit)
{
    const int sub = (int)(threadIdx.x & 15);
    const int group = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 4), ngroups = (int)((gridDim.x * blockDim.x) >> 4);
    const int np_pad = (np + 3) & ~3;       // whole waves stay in the loop for the shuffles
    for (int p = group; p < np_pad; p += ngroups) {
        int any = 0;
        if (p < np) {
            const LmBitImage a = items[pairs[p].x], b = items[pairs[p].y];
            const int x0 = a.x0 > b.x0 ? a.x0 : b.x0, x1 = (a.x0 + a.w < b.x0 + b.w) ? a.x0 + a.w : b.x0 + b.w;     // [x0, x1)
            const int y0 = a.y0 > b.y0 ? a.y0 : b.y0, y1 = (a.y0 + a.h < b.y0 + b.h) ? a.y0 + a.h : b.y0 + b.h;
            const int abw = (a.w + 31) >> 5, bbw = (b.w + 31) >> 5;
            const int chunks = (x1 - x0 + 31) >> 5, total = chunks * (y1 - y0);
            for (int i = sub; i < total && !any; i += 16) {
                const int r = i / chunks, c = i - r * chunks;
                const int x = x0 + c * 32, y = y0 + r;
                unsigned m = lm_bitrow32(bits, a, abw, y - a.y0, x - a.x0) & lm_bitrow32(bits, b, bbw, y - b.y0, x - b.x0);
                if (x1 - x < 32) m &= (1u << (x1 - x)) - 1u;
                any |= (m != 0u);
            }
        }
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) any |= __shfl_xor(any, d, 16);
        if (p < np && sub == 0) hit[p] = any;
    }
}

// ------------------------------------------------------------------------------------------------
// G7: the neighbour structure of the stable uniques, built and kept on the device.
// compute_overlapping_stable_cc (:245-306) joins ALL stable uniques by box (no time limit), so a 10,000-frame lecture has
// ~10^5 stable uniques and ~10^7 overlapping pairs: lists of that size are built, filtered and aggregated here; the host only
// sees what the order-dependent group merge (:308-413) needs.
//   adjacency      row i (stable index) = every j != i whose box overlaps, ASCENDING (count / scan / fill, one wave per row:
//                  the ballot order is the output order).  The reference's pair list sorted by (idx1, idx2) (:274) visits, for
//                  a unique u, first the pairs (c, u) with c < u by ascending c, then the pairs (u, b) by ascending b: u's
//                  neighbour lists are its partners in ascending order, i.e. the rows of this adjacency.
//   match          pixel overlap of every entry (computed for i < j, mirrored by binary search in row j)
//   aov / tov      all_overlapping_cc / time_overlapping_cc rows (:291-304) compacted out of the adjacency in order
//   strong         tov entries with recall >= min_recall (:336): the only edges compute_groups follows -> host
// ------------------------------------------------------------------------------------------------
// Round 3: a wave takes LM_ADJ_R consecutive rows and tests every box it loads against all of them.  One row per wave read the
// whole box array (650 KB for the 10,000-frame stream's 81 k stable uniques) once per ROW: 53 GB through L2 per launch, 3.5 ms.
#define LM_ADJ_R 4

template <int FILL>
__global__ void __launch_bounds__(256) lm_k_adj_rows(const unsigned long long* __restrict__ box, int n, unsigned* __restrict__ cnt_or_off,
                                                     int32_t* __restrict__ adj, int32_t* __restrict__ arow)
{
    const int lane = lm_lane();
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    for (int i0 = wave * LM_ADJ_R; i0 < n; i0 += nwaves * LM_ADJ_R) {
        unsigned long long bi[LM_ADJ_R];
        unsigned base[LM_ADJ_R], c[LM_ADJ_R];
#pragma unroll
        for (int r = 0; r < LM_ADJ_R; r++) {
            const int i = (i0 + r < n) ? i0 + r : n - 1;        // rows past the end repeat the last one; nothing is stored for them
            bi[r] = box[i];
            base[r] = FILL ? cnt_or_off[i] : 0u;
            c[r] = 0;
        }
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            const unsigned long long bj = box[j < n ? j : n - 1];
#pragma unroll
            for (int r = 0; r < LM_ADJ_R; r++) {
                const bool hit = j < n && j != i0 + r && i0 + r < n && lm_box_hit_packed(bi[r], bj);
                const unsigned long long m = __ballot(hit);
                if (FILL && hit) {
                    const unsigned p = base[r] + c[r] + (unsigned)__popcll(m & lm_lowmask_excl(lane));
                    adj[p] = j;
                    arow[p] = i0 + r;
                }
                c[r] += (unsigned)__popcll(m);
            }
        }
        if (!FILL && lane == 0) {
#pragma unroll
            for (int r = 0; r < LM_ADJ_R; r++)
                if (i0 + r < n) cnt_or_off[i0 + r] = c[r];
        }
    }
}

// exclusive scan of n counters into n + 1 offsets (in place allowed); one workgroup.  *total receives the 64-bit sum.
__global__ void __launch_bounds__(1024) lm_k_scan_u32(const unsigned* __restrict__ in, unsigned* __restrict__ out, int n,
                                                      unsigned long long* __restrict__ total)
{
    unsigned long long carry = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const unsigned v = (i < n) ? in[i] : 0u;
        unsigned tot;
        const unsigned ex = lm_block_excl_scan<1024>(v, &tot);
        if (i < n) out[i] = (unsigned)(carry + ex);
        carry += tot;
    }
    if (threadIdx.x == 0) { out[n] = (unsigned)carry; *total = carry; }
}

// pixel overlap of the entries with row < partner; 16 lanes share one entry
__global__ void __launch_bounds__(256) lm_k_adj_match(const LmCcRec* __restrict__ cc, const uint32_t* __restrict__ crop,
                                                      const int32_t* __restrict__ scc, const int32_t* __restrict__ adj,
                                                      const int32_t* __restrict__ arow, long long ne, int32_t* __restrict__ match)
{
    const int sub = (int)(threadIdx.x & 15);
    const long long group = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ngroups = ((long long)gridDim.x * blockDim.x) >> 4;
    const long long ne_pad = (ne + 3) & ~3ll;        // whole waves stay in the loop for the shuffles
    for (long long p = group; p < ne_pad; p += ngroups) {
        int m = 0;
        const bool live = p < ne && arow[p] < adj[p];
        if (live) {
            const LmCcRec a = cc[scc[arow[p]]], u = cc[scc[adj[p]]];
            const LmIsect is = lm_isect(a, u);
            m = lm_overlap_words(a, u, is, crop, sub, 16);
        }
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) m += __shfl_xor(m, d, 16);
        if (live && sub == 0) match[p] = m;
    }
}

// entries with row > partner: the value of the mirrored entry (rows are ascending: binary search)
__global__ void __launch_bounds__(256) lm_k_adj_mirror(const unsigned* __restrict__ off, const int32_t* __restrict__ adj,
                                                       const int32_t* __restrict__ arow, long long ne, int32_t* __restrict__ match)
{
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < ne; p += (long long)gridDim.x * blockDim.x) {
        const int i = arow[p], j = adj[p];
        if (i < j) continue;
        unsigned lo = off[j], hi = off[j + 1];          // first entry of row j that is >= i
        while (lo < hi) {
            const unsigned mid = (lo + hi) >> 1;
            if (adj[mid] < i) lo = mid + 1; else hi = mid;
        }
        match[p] = match[lo];
    }
}

struct LmAdjTab {
    const unsigned* off; const int32_t* adj; const int32_t* match;
    const int32_t *size, *first, *last, *uid;       // per stable unique: pixels of its first-seen CC, first / last frame, unique index
    int n, t_window;
    double min_recall;
};

// 0: no common pixel (not a neighbour at all, :291); 1: all_overlapping only; 2: also inside the temporal window (:299)
LM_DEV int lm_adj_kind(const LmAdjTab& A, int i, int j, int m)
{
    if (m <= 0) return 0;       // recall > 0 or precision > 0  <=>  match > 0 (sizes are positive)
    return (A.last[i] + A.t_window >= A.first[j] && A.last[j] + A.t_window >= A.first[i]) ? 2 : 1;
}

// MODE 0: per-row counts {aov, tov, strong} + total_intersections;  MODE 1: compacted rows at the scanned offsets
struct LmAdjOut {
    unsigned *aov_off, *tov_off, *str_off;           // MODE 0: counts written here; MODE 1: scanned offsets
    int32_t *aov_j, *aov_other, *aov_matched, *aov_size_other, *aov_size_self;
    int32_t* tov_other; double *tov_recall, *tov_precision;
    int32_t* str_j;
    unsigned long long* total_intersections;
};

template <int MODE>
__global__ void __launch_bounds__(256) lm_k_adj_lists(const LmAdjTab A, const LmAdjOut O)
{
    const int lane = lm_lane();
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    for (int i = wave; i < A.n; i += nwaves) {
        const unsigned e0 = A.off[i], e1 = A.off[i + 1];
        const int si = A.size[i];
        unsigned ca = 0, ct = 0, cs = 0, inter = 0;
        const unsigned ba = MODE ? O.aov_off[i] : 0u, bt = MODE ? O.tov_off[i] : 0u, bs = MODE ? O.str_off[i] : 0u;
        for (unsigned eb = e0; eb < e1; eb += 64) {
            const unsigned e = eb + (unsigned)lane;
            int kind = 0, j = 0, m = 0;
            if (e < e1) { j = A.adj[e]; m = A.match[e]; kind = lm_adj_kind(A, i, j, m); }
            const int sj = kind ? A.size[j] : 1;
            const double r_self = (double)m / (double)si, r_other = (double)m / (double)sj;     // connected_component.py:239-240
            const bool strong = kind == 2 && r_self >= A.min_recall;                               // :336, recall relative to the iterating CC
            const unsigned long long ma = __ballot(kind >= 1), mt = __ballot(kind == 2), ms = __ballot(strong);
            if (MODE) {
                const unsigned long long below = lm_lowmask_excl(lane);
                if (kind >= 1) {
                    const unsigned p = ba + ca + (unsigned)__popcll(ma & below);
                    const int sa = (i < j) ? si : sj;                                           // the pair's idx1 < idx2 side
                    O.aov_j[p] = j;
                    O.aov_other[p] = A.uid[j];
                    O.aov_matched[p] = (int)((double)sa * ((double)m / (double)sa));          // float64 round trip, can be match - 1 (:294)
                    O.aov_size_other[p] = sj;
                    O.aov_size_self[p] = si;
                }
                if (kind == 2) {
                    const unsigned p = bt + ct + (unsigned)__popcll(mt & below);
                    O.tov_other[p] = A.uid[j];
                    O.tov_recall[p] = r_self;
                    O.tov_precision[p] = r_other;
                }
                if (strong) O.str_j[bs + cs + (unsigned)__popcll(ms & below)] = j;
            } else {
                inter += (unsigned)__popcll(__ballot(kind == 2 && i < j));
            }
            ca += (unsigned)__popcll(ma); ct += (unsigned)__popcll(mt); cs += (unsigned)__popcll(ms);
        }
        if (!MODE && lane == 0) {
            O.aov_off[i] = ca; O.tov_off[i] = ct; O.str_off[i] = cs;
            if (inter) atomicAdd(O.total_intersections, (unsigned long long)inter);
        }
    }
}

// per stable unique: box, size of its first-seen CC
__global__ void __launch_bounds__(256) lm_k_stable_gather(const LmCcRec* __restrict__ cc, const int32_t* __restrict__ scc, int n,
                                                          unsigned long long* __restrict__ box, int32_t* __restrict__ size)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const LmCcRec r = cc[scc[i]];
        box[i] = lm_pack_box(r);
        size[i] = r.size;
    }
}

// ------------------------------------------------------------------------------------------------
// G8: compute_conflicting_groups (:446-500): every all_overlapping pair (idx1 < idx2) whose members sit in different groups
// adds {matched, unmatched, area_union, area_intersection} to the (g1, g2) and (g2, g1) entries.  Aggregated in a hash table
// keyed by the unordered group pair (64-bit CAS, integer atomics: all four quantities are integers); `first` keeps the smallest
// entry index that touched the key = the pair's insertion rank in the reference's dict-of-dicts (rows ascending, partners
// ascending).  The host only orders the few distinct group pairs.
// ------------------------------------------------------------------------------------------------
struct LmConfTable { unsigned long long* key; unsigned long long *matched, *unmatched, *uni, *inter, *first; unsigned cap_mask; int* overflow; };
#define LM_CONF_EMPTY 0xffffffffffffffffull

__global__ void __launch_bounds__(256) lm_k_conflicts(const unsigned* __restrict__ aov_off, const int32_t* __restrict__ aov_j,
                                                      const int32_t* __restrict__ aov_matched, const int32_t* __restrict__ aov_size_other,
                                                      const int32_t* __restrict__ aov_size_self, const unsigned long long* __restrict__ box,
                                                      const int32_t* __restrict__ sgid, int n, const LmConfTable T)
{
    const int lane = lm_lane();
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    for (int i = wave; i < n; i += nwaves) {
        const unsigned e0 = aov_off[i], e1 = aov_off[i + 1];
        const int gi = sgid[i];
        const unsigned long long bi = box[i];
        const int ix0 = (int)(bi & 0xffff), ix1 = (int)((bi >> 16) & 0xffff), iy0 = (int)((bi >> 32) & 0xffff), iy1 = (int)(bi >> 48);
        for (unsigned e = e0 + (unsigned)lane; e < e1; e += 64) {
            const int j = aov_j[e];
            if (j < i) continue;                                                 // `if not idx1 < idx2: continue`
            const int gj = sgid[j];
            if (gi == gj) continue;
            const unsigned long long bj = box[j];
            const int jx0 = (int)(bj & 0xffff), jx1 = (int)((bj >> 16) & 0xffff), jy0 = (int)((bj >> 32) & 0xffff), jy1 = (int)(bj >> 48);
            long long inter = 0;
            if (ix0 <= jx1 && jx0 <= ix1 && iy0 <= jy1 && jy0 <= iy1)
                inter = (long long)((ix1 < jx1 ? ix1 : jx1) - (ix0 > jx0 ? ix0 : jx0) + 1) * ((iy1 < jy1 ? iy1 : jy1) - (iy0 > jy0 ? iy0 : jy0) + 1);
            const long long uni = (long long)(ix1 - ix0 + 1) * (iy1 - iy0 + 1) + (long long)(jx1 - jx0 + 1) * (jy1 - jy0 + 1) - inter;
            const long long matched = aov_matched[e];
            const long long unmatched = (long long)aov_size_self[e] + aov_size_other[e] - 2 * matched;
            const unsigned glo = (unsigned)(gi < gj ? gi : gj), ghi = (unsigned)(gi < gj ? gj : gi);
            const unsigned long long key = ((unsigned long long)glo << 32) | ghi;
            unsigned h = (unsigned)((key * 0x9e3779b97f4a7c15ull) >> 32) & T.cap_mask;
            bool placed = false;
            for (unsigned probe = 0; probe <= T.cap_mask; probe++) {
                const unsigned long long old = atomicCAS(&T.key[h], LM_CONF_EMPTY, key);
                if (old == LM_CONF_EMPTY || old == key) { placed = true; break; }
                h = (h + 1) & T.cap_mask;
            }
            if (!placed) { *T.overflow = 1; continue; }
            atomicAdd(&T.matched[h], (unsigned long long)matched);
            atomicAdd(&T.unmatched[h], (unsigned long long)unmatched);
            atomicAdd(&T.uni[h], (unsigned long long)uni);
            atomicAdd(&T.inter[h], (unsigned long long)inter);
            atomicMin(&T.first[h], (unsigned long long)e);
        }
    }
}

struct LmConfRow { unsigned long long key, matched, unmatched, uni, inter, first; };

__global__ void __launch_bounds__(256) lm_k_conf_compact(const LmConfTable T, unsigned cap, LmConfRow* __restrict__ out, unsigned* __restrict__ n_out,
                                                         unsigned out_cap)
{
    for (unsigned h = blockIdx.x * blockDim.x + threadIdx.x; h < cap; h += gridDim.x * blockDim.x) {
        const unsigned long long k = T.key[h];
        if (k == LM_CONF_EMPTY) continue;
        const unsigned p = atomicAdd(n_out, 1u);
        if (p < out_cap) out[p] = LmConfRow{k, T.matched[h], T.unmatched[h], T.uni[h], T.inter[h], T.first[h]};
    }
}

// ------------------------------------------------------------------------------------------------
// G9: tables of the group-image kernels, built on the device.  For every (group, member) the segments [ages[j], ages[j+1]]
// (both ends included, :617-619) that hold entries of the member: binary searches in the member's ascending frame list.
// The member goes into the list of every 64 x 64 tile of the item's box that its own box touches: MODE 0 counts per tile,
// MODE 1 writes {first-seen CC, entry count} at the scanned offsets (order inside a tile is irrelevant: the image is an
// integer sum).  Tiles without members are never visited.
// ------------------------------------------------------------------------------------------------
struct LmGimgTab {
    const int32_t *slot_group, *slot_unique, *slot_cc;       // per group-member slot: group, unique index, first-seen CC
    int n_slots;
    const int32_t* ages; const int64_t* ages_off;             // group_ages CSR
    const int64_t* gitem_first;                                // first item of every group
    const int64_t* ulist_off; const int32_t* ulist_frame;      // frames of every unique's entries, ascending
    const int64_t* tile_off;                                   // first tile of every item
    const int32_t* bounds;                                     // group boxes
};

template <int MODE>
__global__ void __launch_bounds__(256) lm_k_gimg_members(const LmGimgTab T, unsigned* __restrict__ tile_cnt, const unsigned* __restrict__ tile_moff,
                                                         LmGimgMember* __restrict__ members, const LmCcRec* __restrict__ cc)
{
    // one wave per (group, member) slot, its lanes take the segments: a member that stays for thousands of frames spans
    // hundreds of segments
    const int lane = lm_lane();
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    for (int s = wave; s < T.n_slots; s += nwaves) {
        const int g = T.slot_group[s], u = T.slot_unique[s];
        const int32_t* ag = T.ages + T.ages_off[g];
        const int na = (int)(T.ages_off[g + 1] - T.ages_off[g]), ns = na - 1;
        if (ns <= 0) continue;
        const int32_t* fe = T.ulist_frame + T.ulist_off[u];
        const int n = (int)(T.ulist_off[u + 1] - T.ulist_off[u]);
        const int first = fe[0], last = fe[n - 1];
        // segments that can hold an entry: ag[j+1] >= first and ag[j] <= last
        int lo = 0, hi = ns;                        // smallest j with ag[j + 1] >= first
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (ag[mid + 1] < first) lo = mid + 1; else hi = mid; }
        const int j_lo = lo;
        lo = 0; hi = ns;                            // number of j with ag[j] <= last
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (ag[mid] <= last) lo = mid + 1; else hi = mid; }
        const int j_hi = lo - 1;
        if (j_hi < j_lo) continue;
        // tiles of the group box the member's box touches (the same for every segment)
        const int gx0 = T.bounds[g * 4 + 0], gy0 = T.bounds[g * 4 + 2], gw = T.bounds[g * 4 + 1] - gx0 + 1;
        const LmCcRec r = cc[T.slot_cc[s]];
        const int ntx = (gw + LM_GT - 1) / LM_GT;
        const int tx0 = (r.min_x - gx0) / LM_GT, tx1 = (r.max_x - gx0) / LM_GT, ty0 = (r.min_y - gy0) / LM_GT, ty1 = (r.max_y - gy0) / LM_GT;
        const long long item0 = T.gitem_first[g];
        for (int j = j_lo + lane; j <= j_hi; j += 64) {
            int a = 0, b = n;                       // first entry >= ag[j]
            while (a < b) { const int mid = (a + b) >> 1; if (fe[mid] < ag[j]) a = mid + 1; else b = mid; }
            const int e_lo = a;
            b = n;                                  // first entry > ag[j + 1]
            while (a < b) { const int mid = (a + b) >> 1; if (fe[mid] <= ag[j + 1]) a = mid + 1; else b = mid; }
            const int count = a - e_lo;
            if (count <= 0) continue;
            const long long t0 = T.tile_off[item0 + j];
            for (int ty = ty0; ty <= ty1; ty++)
                for (int tx = tx0; tx <= tx1; tx++) {
                    const long long t = t0 + (long long)ty * ntx + tx;
                    const unsigned pos = atomicAdd(&tile_cnt[t], 1u);
                    if (MODE) members[tile_moff[t] + pos] = LmGimgMember{T.slot_cc[s], count};
                }
        }
    }
}

// tiles with members (all tiles when `all`) -> work units (any order)
__global__ void __launch_bounds__(256) lm_k_gimg_units(const unsigned* __restrict__ tile_moff, long long n_tiles, const int64_t* __restrict__ tile_off,
                                                       const LmGimgItem* __restrict__ items, int n_items, int all, LmGimgUnit* __restrict__ units,
                                                       unsigned* __restrict__ n_units)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n_tiles; t += (long long)gridDim.x * blockDim.x) {
        const unsigned m0 = tile_moff[t], m1 = tile_moff[t + 1];
        if (m1 == m0 && !all) continue;
        int lo = 0, hi = n_items;           // item of tile t: largest k with tile_off[k] <= t
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tile_off[mid] <= t) lo = mid; else hi = mid; }
        const int ntx = (items[lo].w + LM_GT - 1) / LM_GT;
        const long long rel = t - tile_off[lo];
        const unsigned p = atomicAdd(n_units, 1u);
        units[p] = LmGimgUnit{lo, (int16_t)(rel % ntx), (int16_t)(rel / ntx), m0, m1 - m0};
    }
}

// exclusive scan of a large counter array in three launches: per-chunk scans, scan of the chunk sums (one workgroup), add
#define LM_SCAN_CHUNK 4096
__global__ void __launch_bounds__(1024) lm_k_scan_chunks(const unsigned* __restrict__ in, unsigned* __restrict__ out, long long n,
                                                         unsigned* __restrict__ chunk_sum)
{
    const long long c0 = (long long)blockIdx.x * LM_SCAN_CHUNK;
    unsigned carry = 0;
    for (int base = 0; base < LM_SCAN_CHUNK; base += 1024) {
        const long long i = c0 + base + (long long)threadIdx.x;
        const unsigned v = (i < n) ? in[i] : 0u;
        unsigned tot;
        const unsigned ex = lm_block_excl_scan<1024>(v, &tot);
        if (i < n) out[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) chunk_sum[blockIdx.x] = carry;
}

__global__ void __launch_bounds__(256) lm_k_scan_add(unsigned* __restrict__ out, long long n, const unsigned* __restrict__ chunk_off, int n_chunks)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (long long)gridDim.x * blockDim.x)
        out[i] = (i < n) ? out[i] + chunk_off[i / LM_SCAN_CHUNK] : chunk_off[n_chunks];
}

// bit rows of one item -> the reference's uint8 0 / 255 image (lm_group_array(LM_G_GIMG), on demand)
__global__ void __launch_bounds__(256) lm_k_gimg_expand(const LmGimgItem* __restrict__ items, int n_items, const uint32_t* __restrict__ bits,
                                                        uint8_t* __restrict__ images)
{
    for (int k = blockIdx.y; k < n_items; k += gridDim.y) {
        const LmGimgItem it = items[k];
        const int bw = (it.w + 31) >> 5;
        const long long px = (long long)it.w * it.h;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < px; i += (long long)gridDim.x * blockDim.x) {
            const int y = (int)(i / it.w), x = (int)(i - (long long)y * it.w);
            images[it.img_off + i] = ((bits[it.bits_off + (long long)y * bw + (x >> 5)] >> (x & 31)) & 1u) ? 255 : 0;
        }
    }
}

// ================================================================================================
// host side
// ================================================================================================
struct LmGroups {
    LmStream* s;
    // parameters
    int max_gap, min_times, t_window;
    double min_recall, img_thr;
    int n_frames, n_uniq0;
    // ---- results (host)
    int64_t n_split;
    std::vector<int32_t> uniq_cc;               // [n_uniq] first-seen CC record of every unique (aliases after split)
    std::vector<int64_t> ulist_off;             // [n_uniq+1] CSR of every unique's entries
    std::vector<int32_t> ulist_cc;              // global cc index of each entry (frame and raw label follow from the record)
    std::vector<int32_t> assign;                // [n_cc] unique index per kept CC after the split
    std::vector<int32_t> stable;                // stable unique indices, ascending
    int64_t total_intersections;
    std::vector<int64_t> grp_off; std::vector<int32_t> grp_members;     // cc_groups
    std::vector<int32_t> gid_of_unique;         // [n_uniq] group of a unique or -1
    std::vector<int64_t> ages_off; std::vector<int32_t> ages;           // group_ages
    std::vector<int64_t> gpf_off; std::vector<int32_t> gpf;             // groups_per_frame
    std::vector<int32_t> conf_g1, conf_g2; std::vector<int64_t> conf_matched, conf_unmatched, conf_union; std::vector<double> conf_inter;
    std::vector<int32_t> bounds;                // [n_groups][4] min_x, max_x, min_y, max_y
    std::vector<int64_t> gimg_off;              // [n_items + 1] byte offsets of the segment images (item order: group, segment)
    std::vector<int64_t> gimg_item_off;         // [n_groups + 1] first item of every group
    // materialised from the device tables on first request (lm_group_array): a long lecture has ~10^7 overlapping pairs
    bool have_pairs = false, have_nbr = false;
    std::vector<int32_t> pair_a, pair_b, pair_match;      // bbox-overlapping stable pairs (unique indices, a < b), sorted
    std::vector<int64_t> tov_off; std::vector<int32_t> tov_other; std::vector<double> tov_recall, tov_precision;
    std::vector<int64_t> aov_off; std::vector<int32_t> aov_other, aov_matched, aov_size_other, aov_size_self;
    std::vector<uint8_t> gimg_host;
    // ---- device
    int n_stable = 0;
    long long n_adj = 0, n_aov = 0, n_tov = 0;
    unsigned* d_adj_off = nullptr; int32_t *d_adj = nullptr, *d_arow = nullptr, *d_match = nullptr;
    unsigned *d_aov_off = nullptr, *d_tov_off = nullptr;
    int32_t *d_aov_j = nullptr, *d_aov_other = nullptr, *d_aov_matched = nullptr, *d_aov_size_other = nullptr, *d_aov_size_self = nullptr;
    int32_t* d_tov_other = nullptr; double *d_tov_recall = nullptr, *d_tov_precision = nullptr;
    LmGimgItem* d_items = nullptr;
    int n_items = 0;
    uint32_t* d_gbits = nullptr;                // the segment images as bit rows (renderer input)
    std::vector<int64_t> gbits_off;             // [n_items] word offsets into d_gbits
    long long* d_frame_item_off = nullptr;
    LmRenderItem* d_render_items = nullptr;
    std::vector<void*> d_owned;                 // allocations that did not fit the arena
    char* arena = nullptr;                      // bump arena (the stream's cached one when it was free)
    size_t arena_cap = 0, arena_used = 0, arena_want = 0;
    bool arena_cached = false;
    char* pin = nullptr;                        // pinned host staging (D2H / H2D at PCIe rate instead of the pageable path)
    size_t pin_cap = 0, pin_used = 0;
    std::vector<void*> pin_owned;
    bool pin_cached = false;
    size_t pin_want = 0;
};

// Device memory for one lm_group_run: bump allocation from the arena, hipMalloc only for what does not fit.
static void* lm_galloc(LmGroups* g, size_t bytes)
{
    const size_t need = (bytes + 255) & ~(size_t)255;
    g->arena_want += need;
    if (g->arena && g->arena_used + need <= g->arena_cap) {
        void* p = g->arena + g->arena_used;
        g->arena_used += need;
        return p;
    }
    void* p = nullptr;
    if (hipMalloc(&p, need ? need : 256) != hipSuccess) { lm_set_error("lm_group_run: hipMalloc(%zu) failed", need); return nullptr; }
    g->d_owned.push_back(p);
    return p;
}

// pinned host memory for one run (bump; what does not fit the first block is allocated on its own)
static void* lm_gpin(LmGroups* g, size_t bytes)
{
    const size_t need = (bytes + 255) & ~(size_t)255;
    g->pin_want += need;
    if (g->pin && g->pin_used + need <= g->pin_cap) {
        void* p = g->pin + g->pin_used;
        g->pin_used += need;
        return p;
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, need ? need : 256) != hipSuccess) { lm_set_error("lm_group_run: hipHostMalloc(%zu) failed", need); return nullptr; }
    g->pin_owned.push_back(p);
    return p;
}

#define LM_G_ARRAYS 40
enum {
    LM_G_UNIQ_CC = 0, LM_G_ULIST_OFF, LM_G_ULIST_CC, LM_G_ASSIGN, LM_G_STABLE, LM_G_PAIR_A, LM_G_PAIR_B, LM_G_PAIR_MATCH,
    LM_G_TOV_OFF, LM_G_TOV_OTHER, LM_G_TOV_RECALL, LM_G_TOV_PRECISION, LM_G_AOV_OFF, LM_G_AOV_OTHER, LM_G_AOV_MATCHED,
    LM_G_AOV_SIZE_OTHER, LM_G_AOV_SIZE_SELF, LM_G_GRP_OFF, LM_G_GRP_MEMBERS, LM_G_GID, LM_G_AGES_OFF, LM_G_AGES, LM_G_GPF_OFF,
    LM_G_GPF, LM_G_CONF_G1, LM_G_CONF_G2, LM_G_CONF_MATCHED, LM_G_CONF_UNMATCHED, LM_G_CONF_UNION, LM_G_CONF_INTER, LM_G_BOUNDS,
    LM_G_GIMG_OFF, LM_G_GIMG_ITEM_OFF, LM_G_GIMG, LM_G_SCALARS
};

// host vector -> device (through pinned staging: the async copy is then a real DMA and the vector may go away at once)
template <class T> static int lm_upload(LmGroups* g, const std::vector<T>& v, T** d, hipStream_t st)
{
    *d = (T*)lm_galloc(g, (v.size() ? v.size() : 1) * sizeof(T));
    if (!*d) return LM_ERR_HIP;
    if (v.empty()) return LM_OK;
    void* stage = lm_gpin(g, v.size() * sizeof(T));
    if (!stage) return LM_ERR_HIP;
    memcpy(stage, v.data(), v.size() * sizeof(T));
    LM_HIP(hipMemcpyAsync(*d, stage, v.size() * sizeof(T), hipMemcpyHostToDevice, st));
    return LM_OK;
}

// device -> host vector (synchronises the stream)
template <class T> static int lm_download(LmGroups* g, const T* d, size_t n, std::vector<T>& v, hipStream_t st)
{
    v.resize(n);
    if (!n) return LM_OK;
    LM_HIP(hipMemcpyAsync(v.data(), d, n * sizeof(T), hipMemcpyDeviceToHost, st));
    LM_HIP(hipStreamSynchronize(st));
    return LM_OK;
}

extern "C" void lm_group_destroy(LmGroups* g)
{
    if (!g) return;
    for (void* p : g->d_owned) (void)hipFree(p);
    for (void* p : g->pin_owned) (void)hipHostFree(p);
    if (g->pin) {
        LmStream* s = g->s;
        if (g->pin_cached) {
            s->gpin_busy = 0;
            if (g->pin_want > s->gpin_bytes) {              // grow for the next run
                (void)hipHostFree(s->gpin);
                s->gpin = nullptr; s->gpin_bytes = 0;
                const size_t want = g->pin_want + g->pin_want / 4;
                if (hipHostMalloc(&s->gpin, want) == hipSuccess) s->gpin_bytes = want;
            }
        } else {
            (void)hipHostFree(g->pin);
        }
    }
    if (g->arena) {
        LmStream* s = g->s;
        if (g->arena_cached) {
            s->garena_busy = 0;
            if (g->arena_want > s->garena_bytes) {          // grow for the next run
                (void)hipFree(s->garena);
                s->garena = nullptr; s->garena_bytes = 0;
                size_t want = g->arena_want + g->arena_want / 4;
                if (hipMalloc(&s->garena, want) == hipSuccess) s->garena_bytes = want;
            }
        } else {
            (void)hipFree(g->arena);
        }
    }
    delete g;
}

// LM_GROUP_TIMING=1 prints the wall time of every phase of lm_group_run to stderr
struct LmPhaseTimer {
    bool on;
    std::chrono::steady_clock::time_point t;
    LmPhaseTimer() : on(getenv("LM_GROUP_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
    void mark(const char* what, hipStream_t st = nullptr, bool sync = false)
    {
        if (!on) return;
        if (sync) (void)hipStreamSynchronize(st);
        auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[lm_group] %-36s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// Host loops of step 03 that write disjoint ranges (entry lists, conflict rows) run on a few threads: with 10,000 frames they
// were 24 + 25 of the 130 ms of lm_group_run.  fn(part, parts) is called once per part; parts == 1 runs inline.
// Reached through the C ABI: nothing may leave as an exception.  A part whose thread cannot be started runs inline; a part that throws
// (std::bad_alloc at worst: the parts index pre-sized vectors) is reported after every started thread has been joined.
template <class F>
static bool lm_host_parts(int parts, F fn)
{
    std::atomic<bool> ok{true};
    auto guarded = [&fn, &ok](int p, int n) { try { fn(p, n); } catch (...) { ok = false; } };
    if (parts <= 1) { guarded(0, 1); return ok; }
    std::vector<std::thread> th;
    std::vector<int> inline_parts;
    try { th.reserve((size_t)parts - 1); } catch (...) {}
    for (int p = 1; p < parts; p++) {
        try { th.emplace_back([&guarded, p, parts] { guarded(p, parts); }); }
        catch (...) { inline_parts.push_back(p); }       // std::system_error (no thread to be had) or bad_alloc
    }
    guarded(0, parts);
    for (int p : inline_parts) guarded(p, parts);
    for (auto& t : th) t.join();
    return ok;
}
#define LM_HOST_PARTS(parts, ...)                                                                  \
    do {                                                                                           \
        if (!lm_host_parts(parts, __VA_ARGS__)) { lm_set_error("lm_group_run: a host worker failed (out of memory?)"); return LM_ERR_STATE; } \
    } while (0)

static int lm_host_threads(long long items)
{
    static const int forced = [] { const char* e = getenv("LM_GROUP_THREADS"); return (e && atoi(e) > 0) ? atoi(e) : 0; }();    // tests: any size
    if (forced) return forced;
    static const int hw = [] {
        // CPUs this process may run on (cgroup cpusets / taskset), not the machine's: hardware_concurrency() ignores both
        cpu_set_t set;
        CPU_ZERO(&set);
        unsigned n = (sched_getaffinity(0, sizeof(set), &set) == 0) ? (unsigned)CPU_COUNT(&set) : std::thread::hardware_concurrency();
        return (int)(n >= 16 ? 8 : (n >= 4 ? n / 2 : 1));
    }();
    return items < (1 << 16) ? 1 : hw;
}

static inline unsigned lm_gblocks(long long items, int per_block, unsigned max_blocks)
{
    if (LM_HIP_EMULATED) return 2;
    long long b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > (long long)max_blocks) b = max_blocks;
    return (unsigned)b;
}

static int lm_group_run_impl(LmGroups* g, int reconstruct_tables, hipStream_t st)
{
    LmStream* s = g->s;
    LmPhaseTimer tm;
    int64_t k[7];
    int rc = lm_stream_counters(s, k, st);
    if (rc) return rc;
    const int F = (int)k[0];
    const long long n_cc = k[1];
    const int nU0 = (int)k[3];
    g->n_frames = F;
    g->n_uniq0 = nU0;
    {   // pinned staging for this run: the assignments down, the tables up
        // (the stream keeps the block between runs: hipHostMalloc / hipHostFree of ~100 MB per run cost milliseconds and, like
        // hipMalloc / hipFree, wait for every queue of the device -- in a pipeline, for the other stream's whole backlog)
        const size_t want = (size_t)std::max<long long>(n_cc, 1) * 16 + ((size_t)F + 1) * 64 + ((size_t)4 << 20);
        if (!s->gpin_busy) {
            if (!s->gpin && hipHostMalloc(&s->gpin, want) == hipSuccess) s->gpin_bytes = want;
            if (s->gpin) { g->pin = (char*)s->gpin; g->pin_cap = s->gpin_bytes; g->pin_cached = true; s->gpin_busy = 1; }
        } else if (hipHostMalloc((void**)&g->pin, want) == hipSuccess) {
            g->pin_cap = want;
        }
    }
    // ---- what the host bookkeeping needs of the records: the unique every CC was assigned to and its frame (the frame
    // follows from the per-frame offsets)
    const size_t ncc1 = (size_t)std::max<long long>(n_cc, 1);
    int32_t* h_assign = (int32_t*)lm_gpin(g, ncc1 * sizeof(int32_t));
    long long* h_foff = (long long*)lm_gpin(g, ((size_t)F + 1) * sizeof(long long));
    if (!h_assign || !h_foff) return LM_ERR_HIP;
    if (n_cc > 0) LM_HIP(hipMemcpyAsync(h_assign, s->assign, (size_t)n_cc * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    LM_HIP(hipMemcpyAsync(h_foff, s->frame_cc_off, ((size_t)F + 1) * sizeof(long long), hipMemcpyDeviceToHost, st));
    LM_HIP(hipStreamSynchronize(st));
    tm.mark("counters + assignments D2H");
    const int32_t* cc_assign = h_assign;

    // ---- per-unique entry lists (CC order == ascending frame, the reference's append order)
    std::vector<int64_t> cnt((size_t)nU0 + 1, 0);
    for (long long c = 0; c < n_cc; c++) cnt[(size_t)cc_assign[(size_t)c] + 1]++;
    for (int u = 0; u < nU0; u++) cnt[(size_t)u + 1] += cnt[u];
    std::vector<int32_t> lst(ncc1), lst_frame(ncc1);       // entries of all uniques back to back, and their frames
    {
        // counting sort by unique; part p owns the uniques whose entries lie in its share of `lst` and scans all CCs for them
        std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
        const int parts = lm_host_threads(n_cc);
        LM_HOST_PARTS(parts, [&](int part, int nparts) {
            const int64_t e0 = (int64_t)n_cc * part / nparts, e1 = (int64_t)n_cc * (part + 1) / nparts;
            const int u0 = (int)(std::lower_bound(cnt.begin(), cnt.begin() + nU0, e0) - cnt.begin());
            const int u1 = (part + 1 == nparts) ? nU0 : (int)(std::lower_bound(cnt.begin(), cnt.begin() + nU0, e1) - cnt.begin());
            if (u0 >= u1) return;
            for (int f = 0; f < F; f++)
                for (long long c = h_foff[f]; c < h_foff[f + 1]; c++) {
                    const int a = cc_assign[(size_t)c];
                    if (a < u0 || a >= u1) continue;
                    const int64_t p = pos[(size_t)a]++;
                    lst[(size_t)p] = (int32_t)c;
                    lst_frame[(size_t)p] = f;
                }
        });
    }
    tm.mark("entry lists");
    // ---- split_stable_cc_by_gaps (:181-228)
    g->assign.assign(cc_assign, cc_assign + n_cc);
    g->uniq_cc.resize(nU0);
    std::vector<std::pair<int64_t, int64_t>> seg((size_t)nU0);      // [begin, end) in lst of every unique
    for (int u = 0; u < nU0; u++) {
        seg[u] = {cnt[u], cnt[(size_t)u + 1]};
        g->uniq_cc[u] = lst[(size_t)cnt[u]];    // every unique has at least its first-seen entry
    }
    g->n_split = 0;
    {
        std::vector<int64_t> cuts;              // starts of later runs
        for (int u = 0; u < nU0; u++) {
            const int64_t b = cnt[u], e = cnt[(size_t)u + 1];
            const int64_t n_local = e - b;
            cuts.clear();
            for (int64_t i = b + 1; i < e; i++)
                if (lst_frame[(size_t)i] - lst_frame[(size_t)i - 1] > g->max_gap) cuts.push_back(i);
            if (cuts.empty() || n_local < g->min_times) continue;
            seg[u].second = cuts[0];
            for (size_t ci = 0; ci < cuts.size(); ci++) {
                const int64_t rb = cuts[ci], re = (ci + 1 < cuts.size()) ? cuts[ci + 1] : e;
                const int new_u = (int)seg.size();
                seg.push_back({rb, re});
                g->uniq_cc.push_back(g->uniq_cc[u]);            // another reference to the original CC (:212)
                for (int64_t i = rb; i < re; i++) g->assign[(size_t)lst[(size_t)i]] = new_u;
            }
            g->n_split++;
        }
    }
    const int nU = (int)seg.size();
    g->ulist_off.assign((size_t)nU + 1, 0);
    std::vector<int32_t> ulist_frame;
    if (g->n_split == 0) {          // the usual case: the entry lists are the CSR
        g->ulist_cc.swap(lst);
        g->ulist_cc.resize((size_t)n_cc);
        ulist_frame.swap(lst_frame);
        ulist_frame.resize((size_t)n_cc);
        for (int u = 0; u < nU; u++) g->ulist_off[(size_t)u + 1] = cnt[(size_t)u + 1];
    } else {
        g->ulist_cc.resize((size_t)n_cc);
        ulist_frame.resize((size_t)n_cc);
        size_t w = 0;
        for (int u = 0; u < nU; u++) {
            const size_t len = (size_t)(seg[u].second - seg[u].first);
            if (len) {
                memcpy(g->ulist_cc.data() + w, lst.data() + seg[u].first, len * sizeof(int32_t));
                memcpy(ulist_frame.data() + w, lst_frame.data() + seg[u].first, len * sizeof(int32_t));
            }
            w += len;
            g->ulist_off[(size_t)u + 1] = (int64_t)w;
        }
        g->ulist_cc.resize(w);
        ulist_frame.resize(w);
    }
    auto first_frame = [&](int u) { return ulist_frame[(size_t)g->ulist_off[u]]; };
    auto last_frame = [&](int u) { return ulist_frame[(size_t)g->ulist_off[(size_t)u + 1] - 1]; };
    tm.mark("split + CSR");

    // ---- stable set (:230-236) and what the device needs to know about it
    g->stable.clear();
    for (int u = 0; u < nU; u++)
        if (g->ulist_off[(size_t)u + 1] - g->ulist_off[u] >= g->min_times) g->stable.push_back(u);
    const int nS = (int)g->stable.size();
    g->n_stable = nS;
    std::vector<int32_t> s_cc((size_t)nS), s_first((size_t)nS), s_last((size_t)nS);
    std::vector<int32_t> sidx_of_unique((size_t)nU, -1);
    for (int i = 0; i < nS; i++) {
        const int u = g->stable[i];
        s_cc[i] = g->uniq_cc[u]; s_first[i] = first_frame(u); s_last[i] = last_frame(u);
        sidx_of_unique[u] = i;
    }
    int32_t *d_scc, *d_sfirst, *d_slast, *d_suid, *d_ssize;
    unsigned long long* d_sbox;
    if (lm_upload(g, s_cc, &d_scc, st) || lm_upload(g, s_first, &d_sfirst, st) || lm_upload(g, s_last, &d_slast, st) ||
        lm_upload(g, g->stable, &d_suid, st))
        return LM_ERR_HIP;
    d_sbox = (unsigned long long*)lm_galloc(g, (size_t)std::max(nS, 1) * sizeof(unsigned long long));
    d_ssize = (int32_t*)lm_galloc(g, (size_t)std::max(nS, 1) * sizeof(int32_t));
    unsigned long long* d_tot = (unsigned long long*)lm_galloc(g, 8 * sizeof(unsigned long long));
    if (!d_sbox || !d_ssize || !d_tot) return LM_ERR_HIP;
    LM_HIP(hipMemsetAsync(d_tot, 0, 8 * sizeof(unsigned long long), st));
    std::vector<unsigned long long> s_box;      // host copies for the group boxes
    std::vector<int32_t> s_size;
    g->total_intersections = 0;
    std::vector<unsigned> str_off((size_t)nS + 1, 0);
    std::vector<int32_t> str_j;
    if (nS > 0) {
        hipLaunchKernelGGL(lm_k_stable_gather, dim3(lm_gblocks(nS, 256, 1024)), dim3(256), 0, st, s->cc, d_scc, nS, d_sbox, d_ssize);
        // ---- adjacency of the stable uniques (:245-306)
        g->d_adj_off = (unsigned*)lm_galloc(g, ((size_t)nS + 1) * sizeof(unsigned));
        if (!g->d_adj_off) return LM_ERR_HIP;
        const unsigned row_blocks = lm_gblocks(nS, 4, 4096);       // one wave per row, four rows per workgroup (lm_k_adj_lists)
        const unsigned join_blocks = lm_gblocks(nS, 4 * LM_ADJ_R, 4096);      // lm_k_adj_rows: LM_ADJ_R rows per wave
        hipLaunchKernelGGL((lm_k_adj_rows<0>), dim3(join_blocks), dim3(256), 0, st, d_sbox, nS, g->d_adj_off, (int32_t*)nullptr, (int32_t*)nullptr);
        hipLaunchKernelGGL(lm_k_scan_u32, dim3(1), dim3(1024), 0, st, g->d_adj_off, g->d_adj_off, nS, d_tot);
        unsigned long long h_tot[8];
        LM_HIP(hipMemcpyAsync(h_tot, d_tot, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        LM_HIP(hipStreamSynchronize(st));
        if (h_tot[0] >= (1ull << 31)) { lm_set_error("lm_group_run: %llu overlapping stable pairs (limit 2^31)", h_tot[0]); return LM_ERR_CAPACITY; }
        const long long ne = (long long)h_tot[0];
        g->n_adj = ne;
        const size_t ne1 = (size_t)std::max<long long>(ne, 1);
        g->d_adj = (int32_t*)lm_galloc(g, ne1 * sizeof(int32_t));
        g->d_arow = (int32_t*)lm_galloc(g, ne1 * sizeof(int32_t));
        g->d_match = (int32_t*)lm_galloc(g, ne1 * sizeof(int32_t));
        g->d_aov_off = (unsigned*)lm_galloc(g, ((size_t)nS + 1) * sizeof(unsigned));
        g->d_tov_off = (unsigned*)lm_galloc(g, ((size_t)nS + 1) * sizeof(unsigned));
        unsigned* d_str_off = (unsigned*)lm_galloc(g, ((size_t)nS + 1) * sizeof(unsigned));
        if (!g->d_adj || !g->d_arow || !g->d_match || !g->d_aov_off || !g->d_tov_off || !d_str_off) return LM_ERR_HIP;
        tm.mark("adjacency: count + scan", st, true);
        hipLaunchKernelGGL((lm_k_adj_rows<1>), dim3(join_blocks), dim3(256), 0, st, d_sbox, nS, g->d_adj_off, g->d_adj, g->d_arow);
        tm.mark("adjacency: fill", st, true);
        if (ne > 0) {
            hipLaunchKernelGGL(lm_k_adj_match, dim3(lm_gblocks(ne, 16, 8192)), dim3(256), 0, st, s->cc, s->crop, d_scc, g->d_adj, g->d_arow, ne, g->d_match);
            tm.mark("adjacency: pixel overlaps", st, true);
            hipLaunchKernelGGL(lm_k_adj_mirror, dim3(lm_gblocks(ne, 256, 8192)), dim3(256), 0, st, g->d_adj_off, g->d_adj, g->d_arow, ne, g->d_match);
        }
        LmAdjTab A;
        A.off = g->d_adj_off; A.adj = g->d_adj; A.match = g->d_match; A.size = d_ssize; A.first = d_sfirst; A.last = d_slast; A.uid = d_suid;
        A.n = nS; A.t_window = g->t_window; A.min_recall = g->min_recall;
        LmAdjOut O;
        memset(&O, 0, sizeof(O));
        O.aov_off = g->d_aov_off; O.tov_off = g->d_tov_off; O.str_off = d_str_off; O.total_intersections = d_tot + 1;
        hipLaunchKernelGGL((lm_k_adj_lists<0>), dim3(row_blocks), dim3(256), 0, st, A, O);
        hipLaunchKernelGGL(lm_k_scan_u32, dim3(1), dim3(1024), 0, st, g->d_aov_off, g->d_aov_off, nS, d_tot + 2);
        hipLaunchKernelGGL(lm_k_scan_u32, dim3(1), dim3(1024), 0, st, g->d_tov_off, g->d_tov_off, nS, d_tot + 3);
        hipLaunchKernelGGL(lm_k_scan_u32, dim3(1), dim3(1024), 0, st, d_str_off, d_str_off, nS, d_tot + 4);
        LM_HIP(hipMemcpyAsync(h_tot, d_tot, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        LM_HIP(hipStreamSynchronize(st));
        g->total_intersections = (int64_t)h_tot[1];
        g->n_aov = (long long)h_tot[2];
        g->n_tov = (long long)h_tot[3];
        const long long n_str = (long long)h_tot[4];
        tm.mark("neighbour lists: kinds + counts");
        const size_t na1 = (size_t)std::max<long long>(g->n_aov, 1), nt1 = (size_t)std::max<long long>(g->n_tov, 1);
        g->d_aov_j = (int32_t*)lm_galloc(g, na1 * sizeof(int32_t));
        g->d_aov_other = (int32_t*)lm_galloc(g, na1 * sizeof(int32_t));
        g->d_aov_matched = (int32_t*)lm_galloc(g, na1 * sizeof(int32_t));
        g->d_aov_size_other = (int32_t*)lm_galloc(g, na1 * sizeof(int32_t));
        g->d_aov_size_self = (int32_t*)lm_galloc(g, na1 * sizeof(int32_t));
        g->d_tov_other = (int32_t*)lm_galloc(g, nt1 * sizeof(int32_t));
        g->d_tov_recall = (double*)lm_galloc(g, nt1 * sizeof(double));
        g->d_tov_precision = (double*)lm_galloc(g, nt1 * sizeof(double));
        int32_t* d_str_j = (int32_t*)lm_galloc(g, (size_t)std::max<long long>(n_str, 1) * sizeof(int32_t));
        if (!g->d_aov_j || !g->d_aov_other || !g->d_aov_matched || !g->d_aov_size_other || !g->d_aov_size_self || !g->d_tov_other ||
            !g->d_tov_recall || !g->d_tov_precision || !d_str_j)
            return LM_ERR_HIP;
        O.aov_j = g->d_aov_j; O.aov_other = g->d_aov_other; O.aov_matched = g->d_aov_matched; O.aov_size_other = g->d_aov_size_other;
        O.aov_size_self = g->d_aov_size_self; O.tov_other = g->d_tov_other; O.tov_recall = g->d_tov_recall; O.tov_precision = g->d_tov_precision;
        O.str_j = d_str_j;
        hipLaunchKernelGGL((lm_k_adj_lists<1>), dim3(row_blocks), dim3(256), 0, st, A, O);
        // what compute_groups follows, and the boxes / sizes of the stable uniques, to the host
        unsigned* h_str_off = (unsigned*)lm_gpin(g, ((size_t)nS + 1) * sizeof(unsigned));
        int32_t* h_str_j = (int32_t*)lm_gpin(g, (size_t)std::max<long long>(n_str, 1) * sizeof(int32_t));
        unsigned long long* h_box = (unsigned long long*)lm_gpin(g, (size_t)nS * sizeof(unsigned long long));
        int32_t* h_size = (int32_t*)lm_gpin(g, (size_t)nS * sizeof(int32_t));
        if (!h_str_off || !h_str_j || !h_box || !h_size) return LM_ERR_HIP;
        LM_HIP(hipMemcpyAsync(h_str_off, d_str_off, ((size_t)nS + 1) * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        if (n_str > 0) LM_HIP(hipMemcpyAsync(h_str_j, d_str_j, (size_t)n_str * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        LM_HIP(hipMemcpyAsync(h_box, d_sbox, (size_t)nS * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        LM_HIP(hipMemcpyAsync(h_size, d_ssize, (size_t)nS * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        LM_HIP(hipStreamSynchronize(st));
        str_off.assign(h_str_off, h_str_off + nS + 1);
        str_j.assign(h_str_j, h_str_j + n_str);
        s_box.assign(h_box, h_box + nS);
        s_size.assign(h_size, h_size + nS);
        tm.mark("neighbour lists: fill + strong edges D2H");
        if (tm.on) fprintf(stderr, "[lm_group]   stable %d adjacency %lld aov %lld tov %lld strong %lld\n", nS, ne, g->n_aov, g->n_tov, n_str);

        // ---- compute_groups (:308-413): sequential, order-dependent; works on stable indices.  A group is a singly linked list
        // of its members in the order the reference's list holds them (appends and whole-list concatenations only), so a merge
        // splices in O(1); group numbers are handed out in creation order.
        std::vector<int32_t> sg((size_t)nS, -1), next_m((size_t)nS, -1), head, tail, alias;
        head.reserve((size_t)nS); tail.reserve((size_t)nS); alias.reserve((size_t)nS);
        // an absorbed group forwards to the group that absorbed it (path halving), so a merge does not walk its members:
        // the reference re-points every member (:376-383), which is quadratic when a long-lived group is absorbed again and again
        auto group_of = [&](int x) {
            int k = sg[x];
            while (alias[k] != k) { alias[k] = alias[alias[k]]; k = alias[k]; }
            return k;
        };
        for (int a = 0; a < nS; a++) {
            int gi;
            if (sg[a] < 0) { gi = (int)head.size(); head.push_back(a); tail.push_back(a); alias.push_back(gi); sg[a] = gi; }
            else gi = group_of(a);
            for (unsigned e = str_off[a]; e < str_off[(size_t)a + 1]; e++) {
                const int b = str_j[e];
                if (sg[b] < 0) {
                    sg[b] = gi; next_m[tail[gi]] = b; tail[gi] = b;
                } else {
                    const int og = group_of(b);
                    if (og == gi) continue;
                    next_m[tail[gi]] = head[og]; tail[gi] = tail[og];
                    head[og] = -1;                      // the absorbed group stays behind, empty (:399-411 drops it)
                    alias[og] = gi;
                }
            }
        }
        g->grp_off.assign(1, 0);
        g->grp_members.clear();
        g->grp_members.reserve((size_t)nS);
        g->gid_of_unique.assign((size_t)nU, -1);
        for (size_t k = 0; k < head.size(); k++) {
            if (head[k] < 0) continue;
            const int ng = (int)g->grp_off.size() - 1;
            for (int m = head[k]; m >= 0; m = next_m[m]) { g->grp_members.push_back(g->stable[m]); g->gid_of_unique[g->stable[m]] = ng; }
            g->grp_off.push_back((int64_t)g->grp_members.size());
        }
    } else {
        g->grp_off.assign(1, 0);
        g->grp_members.clear();
        g->gid_of_unique.assign((size_t)nU, -1);
    }
    const int nG = (int)g->grp_off.size() - 1;
    tm.mark("compute_groups");
    // ---- temporal information (:415-444)
    g->ages_off.assign(1, 0); g->ages.clear();
    std::vector<int32_t> g_from((size_t)nG), g_to((size_t)nG);      // live frame range [from, to) of every group
    std::vector<int64_t> fcnt((size_t)F + 2, 0);
    {
        std::vector<int32_t> a;
        for (int gi = 0; gi < nG; gi++) {
            a.clear();
            for (int64_t i = g->grp_off[gi]; i < g->grp_off[(size_t)gi + 1]; i++) {
                const int u = g->grp_members[(size_t)i];
                a.push_back(first_frame(u));
                a.push_back(last_frame(u));
            }
            std::sort(a.begin(), a.end());
            a.erase(std::unique(a.begin(), a.end()), a.end());
            for (int v : a) g->ages.push_back(v);
            g->ages_off.push_back((int64_t)g->ages.size());
            g_from[gi] = a.front(); g_to[gi] = std::max(a.front(), std::min(a.back() + 1, F));
            if (g_to[gi] > g_from[gi]) { fcnt[(size_t)g_from[gi] + 1]++; fcnt[(size_t)g_to[gi] + 1]--; }
        }
    }
    // per-frame lists in ascending group index: difference array -> counts -> offsets -> fill
    g->gpf_off.assign((size_t)F + 1, 0);
    {
        int64_t live = 0;
        for (int f = 0; f < F; f++) { live += fcnt[(size_t)f + 1]; g->gpf_off[(size_t)f + 1] = g->gpf_off[f] + live; }
        g->gpf.assign((size_t)g->gpf_off[F], 0);
        std::vector<int64_t> cur(g->gpf_off.begin(), g->gpf_off.end() - 1);
        LM_HOST_PARTS(lm_host_threads(g->gpf_off[F]), [&](int part, int nparts) {        // a part owns a range of frames
            const int f0 = (int)((int64_t)F * part / nparts), f1 = (int)((int64_t)F * (part + 1) / nparts);
            for (int gi = 0; gi < nG; gi++) {
                const int a0 = std::max(g_from[gi], f0), a1 = std::min(g_to[gi], f1);
                for (int f = a0; f < a1; f++) g->gpf[(size_t)cur[f]++] = gi;
            }
        });
    }
    tm.mark("ages + groups_per_frame");
    // ---- conflicts (:446-500): aggregated on the device per unordered group pair, ordered here.  The reference's dict of
    // dicts is emitted grouped by g1, inner entries in first-insertion order.
    g->conf_g1.clear(); g->conf_g2.clear(); g->conf_matched.clear(); g->conf_unmatched.clear(); g->conf_union.clear(); g->conf_inter.clear();
    if (nS > 0 && g->n_aov > 0) {
        std::vector<int32_t> sgid((size_t)nS);
        for (int i = 0; i < nS; i++) sgid[i] = g->gid_of_unique[g->stable[i]];
        int32_t* d_sgid;
        if (lm_upload(g, sgid, &d_sgid, st)) return LM_ERR_HIP;
        unsigned cap = 1024;
        while ((long long)cap < std::min<long long>(g->n_aov, 1ll << 22)) cap <<= 1;
        cap <<= 1;
        LmConfRow* h_rows = nullptr;
        unsigned n_rows = 0;
        for (;;) {
            LmConfTable T;
            T.key = (unsigned long long*)lm_galloc(g, (size_t)cap * 6 * sizeof(unsigned long long));
            unsigned* d_cnt = (unsigned*)lm_galloc(g, 256);
            if (!T.key || !d_cnt) return LM_ERR_HIP;
            T.matched = T.key + cap; T.unmatched = T.key + 2 * (size_t)cap; T.uni = T.key + 3 * (size_t)cap; T.inter = T.key + 4 * (size_t)cap;
            T.first = T.key + 5 * (size_t)cap;
            T.cap_mask = cap - 1;
            T.overflow = (int*)(d_cnt + 1);
            LM_HIP(hipMemsetAsync(T.key, 0xff, (size_t)cap * sizeof(unsigned long long), st));
            LM_HIP(hipMemsetAsync(T.matched, 0, (size_t)cap * 4 * sizeof(unsigned long long), st));
            LM_HIP(hipMemsetAsync(T.first, 0xff, (size_t)cap * sizeof(unsigned long long), st));
            LM_HIP(hipMemsetAsync(d_cnt, 0, 256, st));
            hipLaunchKernelGGL(lm_k_conflicts, dim3(lm_gblocks(nS, 4, 4096)), dim3(256), 0, st, g->d_aov_off, g->d_aov_j, g->d_aov_matched,
                               g->d_aov_size_other, g->d_aov_size_self, d_sbox, d_sgid, nS, T);
            LmConfRow* d_rows = (LmConfRow*)lm_galloc(g, (size_t)cap * sizeof(LmConfRow));
            if (!d_rows) return LM_ERR_HIP;
            hipLaunchKernelGGL(lm_k_conf_compact, dim3(lm_gblocks(cap, 256, 2048)), dim3(256), 0, st, T, cap, d_rows, d_cnt, cap);
            unsigned h_cnt[2] = {0, 0};
            LM_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(h_cnt), hipMemcpyDeviceToHost, st));
            LM_HIP(hipStreamSynchronize(st));
            if (h_cnt[1] || h_cnt[0] > cap / 2 + cap / 4) { cap <<= 1; continue; }      // too full: probe chains degrade; redo larger
            n_rows = h_cnt[0];
            if (n_rows) {
                h_rows = (LmConfRow*)lm_gpin(g, (size_t)n_rows * sizeof(LmConfRow));
                if (!h_rows) return LM_ERR_HIP;
                LM_HIP(hipMemcpyAsync(h_rows, d_rows, (size_t)n_rows * sizeof(LmConfRow), hipMemcpyDeviceToHost, st));
                LM_HIP(hipStreamSynchronize(st));
            }
            break;
        }
        // both directions of every pair, bucketed by g1 (counting sort), inner order = first touch
        struct Ent { unsigned long long first; int32_t g2; uint32_t row; };
        std::vector<int64_t> coff((size_t)nG + 1, 0);
        for (unsigned i = 0; i < n_rows; i++) { coff[(size_t)(h_rows[i].key >> 32) + 1]++; coff[(size_t)(h_rows[i].key & 0xffffffffu) + 1]++; }
        for (int x = 0; x < nG; x++) coff[(size_t)x + 1] += coff[x];
        std::vector<Ent> ents((size_t)coff[nG]);
        {
            std::vector<int64_t> cur(coff.begin(), coff.end() - 1);
            for (unsigned i = 0; i < n_rows; i++) {
                const int32_t lo = (int32_t)(h_rows[i].key >> 32), hi = (int32_t)(h_rows[i].key & 0xffffffffu);
                ents[(size_t)cur[lo]++] = {h_rows[i].first, hi, i};
                ents[(size_t)cur[hi]++] = {h_rows[i].first, lo, i};
            }
        }
        const size_t ntot = ents.size();
        g->conf_g1.resize(ntot); g->conf_g2.resize(ntot); g->conf_matched.resize(ntot); g->conf_unmatched.resize(ntot);
        g->conf_union.resize(ntot); g->conf_inter.resize(ntot);
        LM_HOST_PARTS(lm_host_threads((long long)ntot), [&](int part, int nparts) {
            // groups whose rows start inside this part's share of the output
            const int64_t e0 = (int64_t)ntot * part / nparts, e1 = (int64_t)ntot * (part + 1) / nparts;
            const int x0 = (int)(std::lower_bound(coff.begin(), coff.begin() + nG, e0) - coff.begin());
            const int x1 = (part + 1 == nparts) ? nG : (int)(std::lower_bound(coff.begin(), coff.begin() + nG, e1) - coff.begin());
            for (int x = x0; x < x1; x++) {
                if (coff[(size_t)x + 1] - coff[x] > 1)
                    std::sort(ents.begin() + coff[x], ents.begin() + coff[(size_t)x + 1], [](const Ent& a, const Ent& b) { return a.first < b.first; });
                for (int64_t i = coff[x]; i < coff[(size_t)x + 1]; i++) {
                    const Ent& e = ents[(size_t)i];
                    const LmConfRow& r = h_rows[e.row];
                    g->conf_g1[(size_t)i] = x; g->conf_g2[(size_t)i] = e.g2; g->conf_matched[(size_t)i] = (int64_t)r.matched;
                    g->conf_unmatched[(size_t)i] = (int64_t)r.unmatched; g->conf_union[(size_t)i] = (int64_t)r.uni;
                    g->conf_inter[(size_t)i] = (double)(int64_t)r.inter;
                }
            }
        });
    }
    tm.mark("conflicts");
    // ---- group images (:575-636): item = (group, age segment); boxes and offsets here, member / tile tables on the device
    g->bounds.assign((size_t)nG * 4, 0);
    g->gimg_off.assign(1, 0);
    g->gbits_off.clear();
    g->gimg_item_off.assign(1, 0);
    std::vector<LmGimgItem> items;
    std::vector<int64_t> tile_off(1, 0);
    std::vector<int32_t> slot_group(g->grp_members.size()), slot_cc(g->grp_members.size());
    long long bit_words = 0;
    for (int gi = 0; gi < nG; gi++) {
        int x0 = 1 << 30, x1 = -1, y0 = 1 << 30, y1 = -1;
        for (int64_t i = g->grp_off[gi]; i < g->grp_off[(size_t)gi + 1]; i++) {
            const int u = g->grp_members[(size_t)i];
            const unsigned long long b = s_box[(size_t)sidx_of_unique[u]];
            x0 = std::min(x0, (int)(b & 0xffff)); x1 = std::max(x1, (int)((b >> 16) & 0xffff));
            y0 = std::min(y0, (int)((b >> 32) & 0xffff)); y1 = std::max(y1, (int)(b >> 48));
            slot_group[(size_t)i] = gi;
            slot_cc[(size_t)i] = g->uniq_cc[u];
        }
        g->bounds[(size_t)gi * 4 + 0] = x0; g->bounds[(size_t)gi * 4 + 1] = x1; g->bounds[(size_t)gi * 4 + 2] = y0; g->bounds[(size_t)gi * 4 + 3] = y1;
        const int w = x1 - x0 + 1, h = y1 - y0 + 1;
        const int ns = (int)(g->ages_off[(size_t)gi + 1] - g->ages_off[gi]) - 1;
        const long long ntiles = (long long)((w + LM_GT - 1) / LM_GT) * ((h + LM_GT - 1) / LM_GT);
        for (int j = 0; j < ns; j++) {
            LmGimgItem it;
            it.x0 = x0; it.y0 = y0; it.w = w; it.h = h;
            it.img_off = g->gimg_off.back();
            it.bits_off = bit_words;
            g->gbits_off.push_back(bit_words);
            g->gimg_off.push_back(it.img_off + (int64_t)w * h);
            bit_words += (long long)h * ((w + 31) >> 5);
            tile_off.push_back(tile_off.back() + ntiles);
            items.push_back(it);
        }
        g->gimg_item_off.push_back((int64_t)items.size());
    }
    const int n_items = (int)items.size();
    const long long n_tiles = tile_off.back();
    g->n_items = n_items;
    tm.mark("  gimg: items (host)");
    g->d_gbits = (uint32_t*)lm_galloc(g, (size_t)std::max<long long>(bit_words, 1) * sizeof(uint32_t));
    if (!g->d_gbits) return LM_ERR_HIP;
    int64_t* d_gitem_first = nullptr;
    int32_t *d_ages = nullptr;
    if (n_items > 0) {
        LmGimgTab T;
        int32_t *d_slot_group, *d_slot_unique, *d_slot_cc, *d_ulist_frame, *d_bounds_g;
        int64_t *d_ages_off64, *d_ulist_off, *d_tile_off;
        std::vector<int64_t> gitem_first(g->gimg_item_off.begin(), g->gimg_item_off.end() - 1);
        if (lm_upload(g, slot_group, &d_slot_group, st) || lm_upload(g, g->grp_members, &d_slot_unique, st) || lm_upload(g, slot_cc, &d_slot_cc, st) ||
            lm_upload(g, g->ages, &d_ages, st) || lm_upload(g, g->ages_off, &d_ages_off64, st) || lm_upload(g, gitem_first, &d_gitem_first, st) ||
            lm_upload(g, g->ulist_off, &d_ulist_off, st) || lm_upload(g, ulist_frame, &d_ulist_frame, st) || lm_upload(g, tile_off, &d_tile_off, st) ||
            lm_upload(g, items, &g->d_items, st) || lm_upload(g, g->bounds, &d_bounds_g, st))
            return LM_ERR_HIP;
        T.slot_group = d_slot_group; T.slot_unique = d_slot_unique; T.slot_cc = d_slot_cc; T.n_slots = (int)g->grp_members.size();
        T.ages = d_ages; T.ages_off = d_ages_off64; T.gitem_first = d_gitem_first; T.ulist_off = d_ulist_off; T.ulist_frame = d_ulist_frame;
        T.tile_off = d_tile_off; T.bounds = d_bounds_g;
        const size_t nt1 = (size_t)std::max<long long>(n_tiles, 1);
        const int n_chunks = (int)((n_tiles + LM_SCAN_CHUNK - 1) / LM_SCAN_CHUNK);
        unsigned* d_tile_cnt = (unsigned*)lm_galloc(g, (nt1 + 1) * sizeof(unsigned));
        unsigned* d_tile_moff = (unsigned*)lm_galloc(g, (nt1 + 1) * sizeof(unsigned));
        unsigned* d_chunk = (unsigned*)lm_galloc(g, ((size_t)n_chunks + 2) * sizeof(unsigned));
        LmGimgUnit* d_units = (LmGimgUnit*)lm_galloc(g, nt1 * sizeof(LmGimgUnit));
        int32_t* d_max = (int32_t*)lm_galloc(g, (size_t)n_items * sizeof(int32_t));
        unsigned* d_nunits = (unsigned*)lm_galloc(g, 256);
        if (!d_tile_cnt || !d_tile_moff || !d_chunk || !d_units || !d_max || !d_nunits) return LM_ERR_HIP;
        LM_HIP(hipMemsetAsync(d_tile_cnt, 0, (nt1 + 1) * sizeof(unsigned), st));
        LM_HIP(hipMemsetAsync(d_nunits, 0, 256, st));
        LM_HIP(hipMemsetAsync(d_max, 0, (size_t)n_items * sizeof(int32_t), st));
        LM_HIP(hipMemsetAsync(g->d_gbits, 0, (size_t)std::max<long long>(bit_words, 1) * sizeof(uint32_t), st));
        const unsigned sb = lm_gblocks(T.n_slots, 4, 16384);      // a wave per slot
        hipLaunchKernelGGL((lm_k_gimg_members<0>), dim3(sb), dim3(256), 0, st, T, d_tile_cnt, (const unsigned*)nullptr, (LmGimgMember*)nullptr, s->cc);
        hipLaunchKernelGGL(lm_k_scan_chunks, dim3((unsigned)std::max(n_chunks, 1)), dim3(1024), 0, st, d_tile_cnt, d_tile_moff, n_tiles, d_chunk);
        hipLaunchKernelGGL(lm_k_scan_u32, dim3(1), dim3(1024), 0, st, d_chunk, d_chunk, n_chunks, d_tot + 5);
        hipLaunchKernelGGL(lm_k_scan_add, dim3(lm_gblocks(n_tiles + 1, 256, 4096)), dim3(256), 0, st, d_tile_moff, n_tiles, d_chunk, n_chunks);
        unsigned long long h_nmem = 0;
        LM_HIP(hipMemcpyAsync(&h_nmem, d_tot + 5, sizeof(h_nmem), hipMemcpyDeviceToHost, st));
        LM_HIP(hipStreamSynchronize(st));
        if (h_nmem >= (1ull << 32)) { lm_set_error("lm_group_run: %llu group-image tile members (limit 2^32)", h_nmem); return LM_ERR_CAPACITY; }
        LmGimgMember* d_members = (LmGimgMember*)lm_galloc(g, (size_t)std::max<unsigned long long>(h_nmem, 1) * sizeof(LmGimgMember));
        if (!d_members) return LM_ERR_HIP;
        LM_HIP(hipMemsetAsync(d_tile_cnt, 0, (nt1 + 1) * sizeof(unsigned), st));
        hipLaunchKernelGGL((lm_k_gimg_members<1>), dim3(sb), dim3(256), 0, st, T, d_tile_cnt, d_tile_moff, d_members, s->cc);
        // a threshold <= 0 turns every pixel of every box on (0 / max >= thr): then every tile is visited
        hipLaunchKernelGGL(lm_k_gimg_units, dim3(lm_gblocks(n_tiles, 256, 4096)), dim3(256), 0, st, d_tile_moff, n_tiles, d_tile_off, g->d_items, n_items,
                           g->img_thr > 0.0 ? 0 : 1, d_units, d_nunits);
        tm.mark("  gimg: member + tile tables (device)", st, true);
        const unsigned nb = LM_HIP_EMULATED ? 2u : (unsigned)std::min<long long>(std::max<long long>(n_tiles, 1), 8192);
        hipLaunchKernelGGL(lm_k_gimg_max, dim3(nb), dim3(256), 0, st, g->d_items, d_units, d_nunits, d_members, s->cc, s->crop, d_max, g->img_thr, g->d_gbits);
        hipLaunchKernelGGL(lm_k_gimg_write, dim3(nb), dim3(256), 0, st, g->d_items, d_units, d_nunits, d_members, s->cc, s->crop, d_max, g->img_thr,
                           g->d_gbits);
        LM_HIP(hipGetLastError());
        if (tm.on) {
            unsigned nu_ = 0;
            (void)hipMemcpyAsync(&nu_, d_nunits, 4, hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            fprintf(stderr, "[lm_group]   groups %d items %d tile members %llu tiles %lld visited %u conflicts %zu\n", nG, n_items, h_nmem, n_tiles, nu_,
                    g->conf_g1.size());
        }
        tm.mark("  gimg: max + write", st, true);
    }
    // ---- render tables for frames_from_groups (:638-681): one item per (frame, live group), built on the device
    if (reconstruct_tables) {
        std::vector<long long> fio(g->gpf_off.begin(), g->gpf_off.end());
        std::vector<int32_t> ages_off32(g->ages_off.begin(), g->ages_off.end());
        int32_t *d_gpf, *d_ages_off, *d_bounds;
        int64_t* d_gbits_off;
        const size_t n_ritems = g->gpf.size();
        g->d_render_items = (LmRenderItem*)lm_galloc(g, std::max<size_t>(n_ritems, 1) * sizeof(LmRenderItem));
        if (!g->d_render_items) return LM_ERR_HIP;
        if (lm_upload(g, fio, &g->d_frame_item_off, st)) return LM_ERR_HIP;
        if (n_ritems) {
            if (lm_upload(g, g->gpf, &d_gpf, st) || lm_upload(g, ages_off32, &d_ages_off, st) || lm_upload(g, g->bounds, &d_bounds, st) ||
                lm_upload(g, g->gbits_off, &d_gbits_off, st))
                return LM_ERR_HIP;
            hipLaunchKernelGGL(lm_k_render_items, dim3(lm_gblocks(F, 1, 16384)), dim3(256), 0, st, g->d_frame_item_off, F, d_gpf,
                               d_ages, d_ages_off, d_bounds, d_gitem_first, d_gbits_off, g->d_render_items);
        }
    }
    LM_HIP(hipGetLastError());
    LM_HIP(hipStreamSynchronize(st));
    tm.mark("render tables + final sync");
    return LM_OK;
}

// PAIR_* / TOV_* / AOV_* arrays of lm_group_array: copied down and re-indexed by unique on first request
static int lm_group_materialize_pairs(LmGroups* g)
{
    if (g->have_pairs) return LM_OK;
    std::vector<int32_t> adj, arow, match;
    hipStream_t st = nullptr;
    if (g->n_adj > 0 && (lm_download(g, g->d_adj, (size_t)g->n_adj, adj, st) || lm_download(g, g->d_arow, (size_t)g->n_adj, arow, st) ||
                         lm_download(g, g->d_match, (size_t)g->n_adj, match, st)))
        return LM_ERR_HIP;
    g->pair_a.clear(); g->pair_b.clear(); g->pair_match.clear();
    for (long long e = 0; e < g->n_adj; e++)
        if (arow[(size_t)e] < adj[(size_t)e]) {       // rows ascending, partners ascending: already sorted by (a, b)
            g->pair_a.push_back(g->stable[arow[(size_t)e]]); g->pair_b.push_back(g->stable[adj[(size_t)e]]); g->pair_match.push_back(match[(size_t)e]);
        }
    g->have_pairs = true;
    return LM_OK;
}

static int lm_group_materialize_neighbours(LmGroups* g)
{
    if (g->have_nbr) return LM_OK;
    hipStream_t st = nullptr;
    const size_t nU = g->uniq_cc.size();
    const int nS = g->n_stable;
    std::vector<unsigned> aoff, toff;
    g->tov_off.assign(nU + 1, 0); g->aov_off.assign(nU + 1, 0);
    if (nS > 0) {
        if (lm_download(g, g->d_aov_off, (size_t)nS + 1, aoff, st) || lm_download(g, g->d_tov_off, (size_t)nS + 1, toff, st)) return LM_ERR_HIP;
        if (lm_download(g, g->d_aov_other, (size_t)g->n_aov, g->aov_other, st) || lm_download(g, g->d_aov_matched, (size_t)g->n_aov, g->aov_matched, st) ||
            lm_download(g, g->d_aov_size_other, (size_t)g->n_aov, g->aov_size_other, st) ||
            lm_download(g, g->d_aov_size_self, (size_t)g->n_aov, g->aov_size_self, st) || lm_download(g, g->d_tov_other, (size_t)g->n_tov, g->tov_other, st) ||
            lm_download(g, g->d_tov_recall, (size_t)g->n_tov, g->tov_recall, st) || lm_download(g, g->d_tov_precision, (size_t)g->n_tov, g->tov_precision, st))
            return LM_ERR_HIP;
        // rows of the device tables are the stable uniques in ascending unique index: the flat arrays are already in
        // per-unique order, only the offsets have to be spread over all uniques
        int si = 0;
        for (size_t u = 0; u < nU; u++) {
            if (si < nS && g->stable[si] == (int32_t)u) {
                g->aov_off[u + 1] = g->aov_off[u] + (int64_t)(aoff[(size_t)si + 1] - aoff[si]);
                g->tov_off[u + 1] = g->tov_off[u] + (int64_t)(toff[(size_t)si + 1] - toff[si]);
                si++;
            } else {
                g->aov_off[u + 1] = g->aov_off[u];
                g->tov_off[u + 1] = g->tov_off[u];
            }
        }
    }
    g->have_nbr = true;
    return LM_OK;
}

extern "C" LmGroups* lm_group_run(LmStream* s, int max_gap, int min_times, int t_window, double min_recall, double img_threshold,
                                  int reconstruct_tables, void* stream)
{
    if (!s) { lm_set_error("lm_group_run: null stream"); return nullptr; }
    LmGroups* g = new LmGroups();
    g->s = s;
    if (!s->garena_busy) {
        if (!s->garena) {
            // first guess from the size of the stream (the group images as bit rows dominate: about a bit per pixel of the stream -- 3,966 MB for the 10,000-frame 1080p bench stream, 1.5 bits per pixel --, plus the
            // adjacency lists); what does not fit is allocated piecewise and the arena re-sized after the run.  A fixed 64 MB made the first run
            // of a long stream a series of hipMalloc / hipFree calls, each a device-wide wait.
            const LmGeom gm = s->ctx->g;
            size_t first = ((size_t)64 << 20) + (size_t)s->frames_pushed * (size_t)gm.W * (size_t)gm.H / 4;
            if (first > ((size_t)16 << 30)) first = (size_t)16 << 30;
            if (hipMalloc(&s->garena, first) == hipSuccess) s->garena_bytes = first;
            else {
                (void)hipGetLastError();
                first = (size_t)64 << 20;
                if (hipMalloc(&s->garena, first) == hipSuccess) s->garena_bytes = first;
            }
        }
        if (s->garena) { g->arena = (char*)s->garena; g->arena_cap = s->garena_bytes; g->arena_cached = true; s->garena_busy = 1; }
    }
    g->max_gap = max_gap; g->min_times = min_times; g->t_window = t_window;
    g->min_recall = min_recall; g->img_thr = img_threshold;
    if (lm_group_run_impl(g, reconstruct_tables, (hipStream_t)stream) != LM_OK) {
        lm_group_destroy(g);
        return nullptr;
    }
    if (getenv("LM_GROUP_TIMING"))
        fprintf(stderr, "[lm_group] device arena: %zu MB wanted, %zu MB cached; pinned staging: %zu MB wanted, %zu MB cached\n", g->arena_want >> 20,
                g->arena_cap >> 20, g->pin_want >> 20, g->pin_cap >> 20);
    return g;
}

// Renders frames [first, first + n) of the reconstructed clean binary stream into d_out ([n][H][W] uint8, device).
extern "C" int lm_group_render(LmGroups* g, int first, int n, uint8_t* d_out, void* stream)
{
    if (!g || !g->d_frame_item_off || first < 0 || n <= 0 || first + n > g->n_frames || !d_out) {
        lm_set_error("lm_group_render: bad arguments (or lm_group_run was called with reconstruct_tables = 0)");
        return LM_ERR_ARG;
    }
    const LmGeom gm = g->s->ctx->g;
    hipLaunchKernelGGL(lm_k_render_frames, dim3((gm.W + LM_RT_COLS - 1) / LM_RT_COLS, (gm.H + LM_RT_ROWS - 1) / LM_RT_ROWS, n), dim3(256), 0,
                       (hipStream_t)stream, g->d_frame_item_off, g->d_render_items, g->d_gbits, first, gm.W, gm.H, d_out);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// Generic accessor: pointer to a host array owned by `g` (valid until lm_group_destroy) and its element count.
// Element types: see the LM_G_* table in include/lecturemath_amd.h.
extern "C" int lm_frame_sums(const uint8_t* d_frames, int n_frames, int64_t pixels_per_frame, uint64_t* d_sums, void* stream)
{
    if (!d_frames || !d_sums || n_frames <= 0 || pixels_per_frame <= 0) { lm_set_error("lm_frame_sums: bad arguments"); return LM_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    LM_HIP(hipMemsetAsync(d_sums, 0, (size_t)n_frames * sizeof(uint64_t), st));
    long long bx = (pixels_per_frame / 16 + 255) / 256;
    if (bx > 64) bx = 64;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(lm_k_frame_sums, dim3((unsigned)(LM_HIP_EMULATED ? 2 : bx), n_frames), dim3(256), 0, st, d_frames, (long long)pixels_per_frame,
                       (unsigned long long*)d_sums);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// Pairs (i < j) of images that share an ink pixel (see G6).  h_boxes [n][4] = min_x, max_x, min_y, max_y (inclusive), h_img_off
// [n + 1] byte offsets into h_images (image k is (max_y - min_y + 1) x (max_x - min_x + 1) uint8, non-zero = ink).  Writes at
// most cap pairs, sorted by (i, j), to h_pairs [cap][2]; *n_pairs receives the number found (LM_ERR_CAPACITY when > cap).
extern "C" int lm_image_pairs_overlap(const int32_t* h_boxes, const uint8_t* h_images, const int64_t* h_img_off, int n, int32_t* h_pairs,
                                      int64_t cap, int64_t* n_pairs, void* stream)
{
    if (!n_pairs || n < 0 || (n > 0 && (!h_boxes || !h_images || !h_img_off)) || cap < 0 || (cap > 0 && !h_pairs)) {
        lm_set_error("lm_image_pairs_overlap: bad arguments");
        return LM_ERR_ARG;
    }
    *n_pairs = 0;
    if (n < 2) return LM_OK;
    hipStream_t st = (hipStream_t)stream;
    std::vector<LmBitImage> items((size_t)n);
    std::vector<unsigned long long> hbox((size_t)n);
    long long words = 0;
    for (int k = 0; k < n; k++) {
        const int32_t* bx = h_boxes + (size_t)k * 4;
        LmBitImage& im = items[(size_t)k];
        im.x0 = bx[0]; im.y0 = bx[2]; im.w = bx[1] - bx[0] + 1; im.h = bx[3] - bx[2] + 1;
        if (im.w <= 0 || im.h <= 0 || bx[0] < 0 || bx[2] < 0 || bx[1] > 32767 || bx[3] > 32767 ||
            h_img_off[k + 1] - h_img_off[k] != (int64_t)im.w * im.h) {
            lm_set_error("lm_image_pairs_overlap: image %d: box / size mismatch", k);
            return LM_ERR_ARG;
        }
        im.src_off = h_img_off[k];
        im.bits_off = words;
        words += (long long)im.h * ((im.w + 31) >> 5);
        hbox[(size_t)k] = (unsigned long long)(unsigned short)bx[0] | ((unsigned long long)(unsigned short)bx[1] << 16) |
                          ((unsigned long long)(unsigned short)bx[2] << 32) | ((unsigned long long)(unsigned short)bx[3] << 48);
    }
    const size_t img_bytes = (size_t)h_img_off[n];
    LmBitImage* d_items = nullptr; unsigned long long* d_box = nullptr; uint8_t* d_src = nullptr; uint32_t* d_bits = nullptr;
    int* d_np = nullptr; int2* d_pairs = nullptr; int32_t* d_hit = nullptr;
    int rc = LM_OK;
    auto done = [&](int code) {
        void* ptrs[] = {d_items, d_box, d_src, d_bits, d_np, d_pairs, d_hit};
        for (void* q : ptrs)
            if (q) (void)hipFree(q);
        return code;
    };
#define LM_PO(x) do { if ((x) != hipSuccess) { lm_set_error("lm_image_pairs_overlap: HIP error at %s", #x); return done(LM_ERR_HIP); } } while (0)
    LM_PO(hipMalloc((void**)&d_items, items.size() * sizeof(LmBitImage)));
    LM_PO(hipMalloc((void**)&d_box, hbox.size() * sizeof(unsigned long long)));
    LM_PO(hipMalloc((void**)&d_src, std::max<size_t>(img_bytes, 1)));
    LM_PO(hipMalloc((void**)&d_bits, (size_t)std::max<long long>(words, 1) * sizeof(uint32_t)));
    LM_PO(hipMalloc((void**)&d_np, sizeof(int)));
    LM_PO(hipMemcpyAsync(d_items, items.data(), items.size() * sizeof(LmBitImage), hipMemcpyHostToDevice, st));
    LM_PO(hipMemcpyAsync(d_box, hbox.data(), hbox.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    LM_PO(hipMemcpyAsync(d_src, h_images, img_bytes, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(lm_k_img_pack, dim3(LM_HIP_EMULATED ? 1 : 4, (unsigned)std::min(n, LM_HIP_EMULATED ? 2 : 4096)), dim3(256), 0, st, d_items, n,
                       d_src, d_bits);
    int cap_pairs = 1 << 16, np = 0;
    for (;;) {
        LM_PO(hipMalloc((void**)&d_pairs, (size_t)cap_pairs * sizeof(int2)));
        LM_PO(hipMemsetAsync(d_np, 0, sizeof(int), st));
        const int gx = std::min((n + 255) / 256, LM_HIP_EMULATED ? 2 : 64), gy = std::min((n + LM_SJ_TILE - 1) / LM_SJ_TILE, LM_HIP_EMULATED ? 2 : 64);
        hipLaunchKernelGGL(lm_k_selfjoin, dim3(gx, gy), dim3(256), 0, st, d_box, n, d_np, d_pairs, cap_pairs);
        LM_PO(hipMemcpyAsync(&np, d_np, sizeof(int), hipMemcpyDeviceToHost, st));
        LM_PO(hipStreamSynchronize(st));
        if (np <= cap_pairs) break;
        (void)hipFree(d_pairs); d_pairs = nullptr;
        cap_pairs = np + (np >> 3);
    }
    std::vector<std::pair<int32_t, int32_t>> found;
    if (np > 0) {
        LM_PO(hipMalloc((void**)&d_hit, (size_t)np * sizeof(int32_t)));
        hipLaunchKernelGGL(lm_k_bitimg_pair_any, dim3(LM_HIP_EMULATED ? 2 : 1024), dim3(256), 0, st, d_items, d_bits, d_pairs, np, d_hit);
        std::vector<int2> hp((size_t)np);
        std::vector<int32_t> hh((size_t)np);
        LM_PO(hipMemcpyAsync(hp.data(), d_pairs, (size_t)np * sizeof(int2), hipMemcpyDeviceToHost, st));
        LM_PO(hipMemcpyAsync(hh.data(), d_hit, (size_t)np * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        LM_PO(hipStreamSynchronize(st));
        for (int i = 0; i < np; i++)
            if (hh[(size_t)i]) found.push_back({hp[(size_t)i].x, hp[(size_t)i].y});
        std::sort(found.begin(), found.end());
    }
#undef LM_PO
    *n_pairs = (int64_t)found.size();
    if ((int64_t)found.size() > cap) {
        lm_set_error("lm_image_pairs_overlap: %zu pairs, room for %lld", found.size(), (long long)cap);
        return done(LM_ERR_CAPACITY);
    }
    for (size_t i = 0; i < found.size(); i++) { h_pairs[i * 2] = found[i].first; h_pairs[i * 2 + 1] = found[i].second; }
    return done(rc);
}

extern "C" int lm_group_array(LmGroups* g, int which, const void** ptr, int64_t* count)
{
    if (!g || !ptr || !count) { lm_set_error("lm_group_array: bad arguments"); return LM_ERR_ARG; }
#define LM_GA(id, vec) case id: *ptr = (vec).data(); *count = (int64_t)(vec).size(); return LM_OK;
    static thread_local int64_t scalars[8];
    if (which >= LM_G_PAIR_A && which <= LM_G_PAIR_MATCH && lm_group_materialize_pairs(g)) return LM_ERR_HIP;
    if (which >= LM_G_TOV_OFF && which <= LM_G_AOV_SIZE_SELF && lm_group_materialize_neighbours(g)) return LM_ERR_HIP;
    switch (which) {
        LM_GA(LM_G_UNIQ_CC, g->uniq_cc) LM_GA(LM_G_ULIST_OFF, g->ulist_off) LM_GA(LM_G_ULIST_CC, g->ulist_cc) LM_GA(LM_G_ASSIGN, g->assign)
        LM_GA(LM_G_STABLE, g->stable) LM_GA(LM_G_PAIR_A, g->pair_a) LM_GA(LM_G_PAIR_B, g->pair_b) LM_GA(LM_G_PAIR_MATCH, g->pair_match)
        LM_GA(LM_G_TOV_OFF, g->tov_off) LM_GA(LM_G_TOV_OTHER, g->tov_other) LM_GA(LM_G_TOV_RECALL, g->tov_recall)
        LM_GA(LM_G_TOV_PRECISION, g->tov_precision) LM_GA(LM_G_AOV_OFF, g->aov_off) LM_GA(LM_G_AOV_OTHER, g->aov_other)
        LM_GA(LM_G_AOV_MATCHED, g->aov_matched) LM_GA(LM_G_AOV_SIZE_OTHER, g->aov_size_other) LM_GA(LM_G_AOV_SIZE_SELF, g->aov_size_self)
        LM_GA(LM_G_GRP_OFF, g->grp_off) LM_GA(LM_G_GRP_MEMBERS, g->grp_members) LM_GA(LM_G_GID, g->gid_of_unique)
        LM_GA(LM_G_AGES_OFF, g->ages_off) LM_GA(LM_G_AGES, g->ages) LM_GA(LM_G_GPF_OFF, g->gpf_off) LM_GA(LM_G_GPF, g->gpf)
        LM_GA(LM_G_CONF_G1, g->conf_g1) LM_GA(LM_G_CONF_G2, g->conf_g2) LM_GA(LM_G_CONF_MATCHED, g->conf_matched)
        LM_GA(LM_G_CONF_UNMATCHED, g->conf_unmatched) LM_GA(LM_G_CONF_UNION, g->conf_union) LM_GA(LM_G_CONF_INTER, g->conf_inter)
        LM_GA(LM_G_BOUNDS, g->bounds) LM_GA(LM_G_GIMG_OFF, g->gimg_off) LM_GA(LM_G_GIMG_ITEM_OFF, g->gimg_item_off)
        case LM_G_GIMG: {
            // the reference's uint8 segment images (compute_group_images :630), expanded from the bit rows on demand
            if (g->gimg_host.size() != (size_t)g->gimg_off.back()) {
                g->gimg_host.resize((size_t)g->gimg_off.back());
                if (!g->gimg_host.empty()) {
                    uint8_t* d_img = nullptr;
                    LM_HIP(hipMalloc((void**)&d_img, g->gimg_host.size()));
                    hipLaunchKernelGGL(lm_k_gimg_expand, dim3(LM_HIP_EMULATED ? 1 : 8, (unsigned)std::min(g->n_items, LM_HIP_EMULATED ? 2 : 4096)), dim3(256), 0,
                                       (hipStream_t) nullptr, g->d_items, g->n_items, g->d_gbits, d_img);
                    const hipError_t e = hipMemcpy(g->gimg_host.data(), d_img, g->gimg_host.size(), hipMemcpyDeviceToHost);
                    (void)hipFree(d_img);
                    LM_HIP(e);
                }
            }
            *ptr = g->gimg_host.data(); *count = (int64_t)g->gimg_host.size();
            return LM_OK;
        }
        case LM_G_SCALARS:
            scalars[0] = g->n_split; scalars[1] = g->total_intersections; scalars[2] = (int64_t)g->grp_off.size() - 1;
            scalars[3] = (int64_t)g->uniq_cc.size(); scalars[4] = g->n_frames; scalars[5] = g->gimg_off.back();
            *ptr = scalars; *count = 6;
            return LM_OK;
        default: break;
    }
#undef LM_GA
    lm_set_error("lm_group_array: unknown array id %d", which);
    return LM_ERR_ARG;
}
