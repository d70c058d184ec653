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

#include <algorithm>
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
//   pass A (lm_k_gimg_max)    per-item maximum of the accumulated mask
//   pass B (lm_k_gimg_write)  recompute and write  ((double)v / (double)max >= thr) ? 255 : 0
// ------------------------------------------------------------------------------------------------
struct LmGimgItem {
    int32_t x0, y0, w, h;        // group box origin and size
    int32_t mem_off, mem_cnt;    // slice of the member list
    long long img_off;           // byte offset of the item's (h x w) uint8 image
    long long bits_off;          // word offset of the same image as bit rows (ceil(w / 32) words per row), what the renderer reads
};
struct LmGimgMember { int32_t cc; int32_t count; };
struct LmGimgUnit { int32_t item; int16_t tx, ty; };

#define LM_GT 64   // tile side

LM_DEV void lm_gimg_accumulate(const LmGimgItem& it, int tx, int ty, const LmGimgMember* __restrict__ members,
                               const LmCcRec* __restrict__ cc, const uint32_t* __restrict__ crop, int* s_mask)
{
    // tile covers x in [X0, X0+64), y in [Y0, Y0+64) of the frame
    const int X0 = it.x0 + tx * LM_GT, Y0 = it.y0 + ty * LM_GT;
    for (int i = threadIdx.x; i < LM_GT * LM_GT; i += blockDim.x) s_mask[i] = 0;
    __syncthreads();
    const int wave = (int)(threadIdx.x >> 6), nwaves = (int)(blockDim.x >> 6), lane = lm_lane();
    for (int m = wave; m < it.mem_cnt; m += nwaves) {
        const LmGimgMember mem = members[it.mem_off + m];
        const LmCcRec r = cc[mem.cc];
        if (r.max_x < X0 || r.min_x >= X0 + LM_GT || r.max_y < Y0 || r.min_y >= Y0 + LM_GT) continue;
        const int wx0 = r.min_x >> 5, nw = (r.max_x >> 5) - wx0 + 1;
        const int ya = r.min_y > Y0 ? r.min_y : Y0;
        const int yb = r.max_y < Y0 + LM_GT - 1 ? r.max_y : Y0 + LM_GT - 1;
        const int total = nw * (yb - ya + 1);
        for (int idx = lane; idx < total; idx += 64) {
            const int rr = idx / nw, j = idx - rr * nw;
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

__global__ void __launch_bounds__(256) lm_k_gimg_max(const LmGimgItem* __restrict__ items, const LmGimgUnit* __restrict__ units,
                                                     int n_units, const LmGimgMember* __restrict__ members,
                                                     const LmCcRec* __restrict__ cc, const uint32_t* __restrict__ crop,
                                                     int32_t* __restrict__ item_max)
{
    __shared__ int s_mask[LM_GT * LM_GT];
    __shared__ int s_max;
    for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
        const LmGimgUnit un = units[u];
        const LmGimgItem it = items[un.item];
        if (threadIdx.x == 0) s_max = 0;
        lm_gimg_accumulate(it, un.tx, un.ty, members, cc, crop, s_mask);
        int mx = 0;
        for (int i = threadIdx.x; i < LM_GT * LM_GT; i += blockDim.x) mx = s_mask[i] > mx ? s_mask[i] : mx;
        if (mx) atomicMax(&s_max, mx);
        __syncthreads();
        if (threadIdx.x == 0 && s_max) atomicMax(&item_max[un.item], s_max);
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) lm_k_gimg_write(const LmGimgItem* __restrict__ items, const LmGimgUnit* __restrict__ units,
                                                       int n_units, const LmGimgMember* __restrict__ members,
                                                       const LmCcRec* __restrict__ cc, const uint32_t* __restrict__ crop,
                                                       const int32_t* __restrict__ item_max, double thr, uint8_t* __restrict__ images,
                                                       uint32_t* __restrict__ bits)
{
    __shared__ int s_mask[LM_GT * LM_GT];
    for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
        const LmGimgUnit un = units[u];
        const LmGimgItem it = items[un.item];
        lm_gimg_accumulate(it, un.tx, un.ty, members, cc, crop, s_mask);
        const double mx = (double)item_max[un.item];
        const int tw = (it.w - un.tx * LM_GT < LM_GT) ? it.w - un.tx * LM_GT : LM_GT;
        const int th = (it.h - un.ty * LM_GT < LM_GT) ? it.h - un.ty * LM_GT : LM_GT;
        for (int i = threadIdx.x; i < tw * th; i += blockDim.x) {
            const int yy = i / tw, xx = i - yy * tw;
            const double v = (double)s_mask[yy * LM_GT + xx] / mx;      // float64 like numpy (:630); max >= 1 by construction
            images[it.img_off + (long long)(un.ty * LM_GT + yy) * it.w + (un.tx * LM_GT + xx)] = (v >= thr) ? 255 : 0;
        }
        // the same decisions as bit rows: a 64-px tile row is two whole words of the item's row
        const int bw = (it.w + 31) >> 5;
        for (int i = threadIdx.x; i < th * 2; i += blockDim.x) {
            const int yy = i >> 1, hf = i & 1;
            if (un.tx * 2 + hf >= bw) continue;
            unsigned word = 0;
            for (int b = 0; b < 32; b++) {
                const int xx = hf * 32 + b;
                if (xx < tw && (double)s_mask[yy * LM_GT + xx] / mx >= thr) word |= 1u << b;
            }
            bits[it.bits_off + (long long)(un.ty * LM_GT + yy) * bw + un.tx * 2 + hf] = word;
        }
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
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= frame_item_off[F]) return;
    int lo = 0, hi = F;         // frame of item i: largest f with frame_item_off[f] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (frame_item_off[mid] <= i) lo = mid; else hi = mid;
    }
    const int f = lo, gi = gpf[i];
    const int32_t* a = ages + ages_off[gi];
    const int na = ages_off[gi + 1] - ages_off[gi];
    LmRenderItem ri;
    ri.x0 = bounds[gi * 4 + 0]; ri.y0 = bounds[gi * 4 + 2];
    ri.w = bounds[gi * 4 + 1] - ri.x0 + 1; ri.h = bounds[gi * 4 + 3] - ri.y0 + 1;
    ri.bits_off = 0;
    if (na < 2) {               // no segment image (the reference would raise IndexError at :650 here): an item that hits no tile
        ri.w = 0; ri.h = 0;
    } else {
        int sidx = 0;
        while (sidx + 1 < na - 1 && a[sidx + 1] < f) sidx++;
        ri.bits_off = gbits_off[gitem_first[gi] + sidx];
    }
    out[i] = ri;
}

__global__ void __launch_bounds__(256) lm_k_render_frames(const long long* __restrict__ frame_item_off,
                                                          const LmRenderItem* __restrict__ items, const uint32_t* __restrict__ bits,
                                                          int first_frame, int W, int H, uint8_t* __restrict__ out)
{
    __shared__ unsigned s_pl[8 * LM_RT_PLANE];      // bit-sliced k per pixel (see lm_render_paint)
    __shared__ LmRenderItem s_hit[LM_RT_MAXHIT];
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
    const int wave = (int)(threadIdx.x >> 6), nwaves = (int)(blockDim.x >> 6), lane = lm_lane();
    for (int h = wave; h < nhit; h += nwaves) lm_render_paint(s_hit[h], X0, Y0, bits, s_pl, lane, 64);
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
    std::vector<int32_t> pair_a, pair_b, pair_match;      // bbox-overlapping stable pairs (unique indices, a < b), sorted
    int64_t total_intersections;
    std::vector<int64_t> tov_off; std::vector<int32_t> tov_other; std::vector<double> tov_recall, tov_precision;
    std::vector<int64_t> aov_off; std::vector<int32_t> aov_other, aov_matched, aov_size_other, aov_size_self;
    std::vector<int64_t> grp_off; std::vector<int32_t> grp_members;     // cc_groups
    std::vector<int32_t> gid_of_unique;         // [n_uniq] group of a unique or -1
    std::vector<int64_t> ages_off; std::vector<int32_t> ages;           // group_ages
    std::vector<int64_t> gpf_off; std::vector<int32_t> gpf;             // groups_per_frame
    std::vector<int32_t> conf_g1, conf_g2; std::vector<int64_t> conf_matched, conf_unmatched, conf_union; std::vector<double> conf_inter;
    std::vector<int32_t> bounds;                // [n_groups][4] min_x, max_x, min_y, max_y
    std::vector<int64_t> gimg_off;              // [n_items + 1] byte offsets of the segment images (item order: group, segment)
    std::vector<int64_t> gimg_item_off;         // [n_groups + 1] first item of every group
    std::vector<uint8_t> gimg_host;             // filled on demand
    // ---- device
    uint8_t* d_images = nullptr;
    uint32_t* d_gbits = nullptr;                // the same images as bit rows (renderer input)
    std::vector<int64_t> gbits_off;             // [n_items] word offsets into d_gbits
    long long* d_frame_item_off = nullptr;
    LmRenderItem* d_render_items = nullptr;
    std::vector<void*> d_owned;                 // allocations that did not fit the arena
    char* arena = nullptr;                      // bump arena (the stream's cached one when it was free)
    size_t arena_cap = 0, arena_used = 0, arena_want = 0;
    bool arena_cached = false;
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

#define LM_G_ARRAYS 40
enum {
    LM_G_UNIQ_CC = 0, LM_G_ULIST_OFF, LM_G_ULIST_CC, LM_G_ASSIGN, LM_G_STABLE, LM_G_PAIR_A, LM_G_PAIR_B, LM_G_PAIR_MATCH,
    LM_G_TOV_OFF, LM_G_TOV_OTHER, LM_G_TOV_RECALL, LM_G_TOV_PRECISION, LM_G_AOV_OFF, LM_G_AOV_OTHER, LM_G_AOV_MATCHED,
    LM_G_AOV_SIZE_OTHER, LM_G_AOV_SIZE_SELF, LM_G_GRP_OFF, LM_G_GRP_MEMBERS, LM_G_GID, LM_G_AGES_OFF, LM_G_AGES, LM_G_GPF_OFF,
    LM_G_GPF, LM_G_CONF_G1, LM_G_CONF_G2, LM_G_CONF_MATCHED, LM_G_CONF_UNMATCHED, LM_G_CONF_UNION, LM_G_CONF_INTER, LM_G_BOUNDS,
    LM_G_GIMG_OFF, LM_G_GIMG_ITEM_OFF, LM_G_GIMG, LM_G_SCALARS
};

template <class T> static int lm_upload(LmGroups* g, const std::vector<T>& v, T** d, hipStream_t st)
{
    *d = (T*)lm_galloc(g, (v.size() ? v.size() : 1) * sizeof(T));
    if (!*d) return LM_ERR_HIP;
    // the host vectors outlive the copy: they are members of g or locals that live until the final stream sync
    if (!v.empty()) LM_HIP(hipMemcpyAsync(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
    return LM_OK;
}

extern "C" void lm_group_destroy(LmGroups* g)
{
    if (!g) return;
    for (void* p : g->d_owned) (void)hipFree(p);
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

#define LM_GROUP_THREADS 4     // host threads tabulating group images

// LM_GROUP_TIMING=1 prints the wall time of every phase of lm_group_run to stderr
struct LmPhaseTimer {
    bool on;
    std::chrono::steady_clock::time_point t;
    LmPhaseTimer() : on(getenv("LM_GROUP_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
    void mark(const char* what)
    {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[lm_group] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

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
    // ---- records to the host
    std::vector<int32_t> rec((size_t)std::max<long long>(n_cc, 1) * 8);
    std::vector<int64_t> foff((size_t)F + 1);
    rc = lm_stream_read(s, rec.data(), foff.data(), nullptr, nullptr, nullptr, st);
    if (rc) return rc;
    tm.mark("counters + records D2H");
    auto R = [&](long long c, int field) { return rec[(size_t)c * 8 + field]; };   // 0 cc_id 1 min_x 2 max_x 3 min_y 4 max_y 5 size 6 frame 7 assign
    // the two fields every per-CC loop below needs, as dense arrays (one sequential pass over the records)
    std::vector<int32_t> cc_frame((size_t)std::max<long long>(n_cc, 1)), cc_assign((size_t)std::max<long long>(n_cc, 1));
    for (long long c = 0; c < n_cc; c++) { cc_frame[(size_t)c] = rec[(size_t)c * 8 + 6]; cc_assign[(size_t)c] = rec[(size_t)c * 8 + 7]; }

    // ---- per-unique entry lists (CC order == ascending frame, the reference's append order)
    std::vector<int64_t> cnt((size_t)nU0 + 1, 0);
    for (long long c = 0; c < n_cc; c++) cnt[(size_t)cc_assign[(size_t)c] + 1]++;
    for (int u = 0; u < nU0; u++) cnt[(size_t)u + 1] += cnt[u];
    std::vector<int32_t> lst((size_t)std::max<long long>(n_cc, 1));
    {
        std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
        for (long long c = 0; c < n_cc; c++) lst[(size_t)pos[cc_assign[(size_t)c]]++] = (int32_t)c;
    }
    tm.mark("entry lists");
    // ---- split_stable_cc_by_gaps (:181-228)
    g->assign = cc_assign;
    g->uniq_cc.resize(nU0);
    std::vector<std::pair<int64_t, int64_t>> seg((size_t)nU0);      // [begin, end) in lst of every unique
    for (int u = 0; u < nU0; u++) {
        seg[u] = {cnt[u], cnt[(size_t)u + 1]};
        g->uniq_cc[u] = lst[(size_t)cnt[u]];    // every unique has at least its first-seen entry
    }
    g->n_split = 0;
    for (int u = 0; u < nU0; u++) {
        const int64_t b = cnt[u], e = cnt[(size_t)u + 1];
        const int64_t n_local = e - b;
        std::vector<int64_t> cuts;              // starts of later runs
        for (int64_t i = b + 1; i < e; i++)
            if (cc_frame[(size_t)lst[(size_t)i]] - cc_frame[(size_t)lst[(size_t)i - 1]] > g->max_gap) cuts.push_back(i);
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
    const int nU = (int)seg.size();
    g->ulist_off.assign((size_t)nU + 1, 0);
    g->ulist_cc.clear();
    g->ulist_cc.reserve((size_t)n_cc);
    for (int u = 0; u < nU; u++) {
        for (int64_t i = seg[u].first; i < seg[u].second; i++) g->ulist_cc.push_back(lst[(size_t)i]);
        g->ulist_off[(size_t)u + 1] = (int64_t)g->ulist_cc.size();
    }
    auto first_frame = [&](int u) { return cc_frame[(size_t)g->ulist_cc[(size_t)g->ulist_off[u]]]; };
    auto last_frame = [&](int u) { return cc_frame[(size_t)g->ulist_cc[(size_t)g->ulist_off[(size_t)u + 1] - 1]]; };
    auto usize = [&](int u) { return R(g->uniq_cc[u], 5); };
    auto ubox = [&](int u, int i) { return R(g->uniq_cc[u], 1 + i); };     // min_x max_x min_y max_y

    tm.mark("split + CSR");
    // ---- stable set (:230-236)
    g->stable.clear();
    for (int u = 0; u < nU; u++)
        if (g->ulist_off[(size_t)u + 1] - g->ulist_off[u] >= g->min_times) g->stable.push_back(u);
    const int nS = (int)g->stable.size();

    // ---- overlapping stable CCs (:245-306): box self-join + pixel overlaps on the device
    g->pair_a.clear(); g->pair_b.clear(); g->pair_match.clear();
    if (nS > 1) {
        std::vector<unsigned long long> hbox((size_t)nS);
        std::vector<int32_t> hcc((size_t)nS);
        for (int i = 0; i < nS; i++) {
            const int u = g->stable[i];
            hbox[i] = (unsigned long long)(unsigned short)ubox(u, 0) | ((unsigned long long)(unsigned short)ubox(u, 1) << 16) |
                      ((unsigned long long)(unsigned short)ubox(u, 2) << 32) | ((unsigned long long)(unsigned short)ubox(u, 3) << 48);
            hcc[i] = g->uniq_cc[u];
        }
        unsigned long long* d_box; int32_t* d_cc; int* d_np;
        if (lm_upload(g, hbox, &d_box, st) || lm_upload(g, hcc, &d_cc, st)) return LM_ERR_HIP;
        d_np = (int*)lm_galloc(g, 64);
        if (!d_np) return LM_ERR_HIP;
        int cap_pairs = 1 << 20;
        int2* d_pairs = nullptr;
        int np = 0;
        for (;;) {
            d_pairs = (int2*)lm_galloc(g, (size_t)cap_pairs * sizeof(int2));
            if (!d_pairs) return LM_ERR_HIP;
            LM_HIP(hipMemsetAsync(d_np, 0, sizeof(int), st));
            const int gx = std::min((nS + 255) / 256, LM_HIP_EMULATED ? 2 : 64), gy = std::min((nS + LM_SJ_TILE - 1) / LM_SJ_TILE, LM_HIP_EMULATED ? 2 : 64);
            hipLaunchKernelGGL(lm_k_selfjoin, dim3(gx, gy), dim3(256), 0, st, d_box, nS, d_np, d_pairs, cap_pairs);
            LM_HIP(hipMemcpyAsync(&np, d_np, sizeof(int), hipMemcpyDeviceToHost, st));
            LM_HIP(hipStreamSynchronize(st));
            if (np <= cap_pairs) break;
            cap_pairs = np + (np >> 3);     // grow and redo (the join is cheap; the small buffer stays in the arena)
        }
        if (np > 0) {
            int32_t* d_match = (int32_t*)lm_galloc(g, (size_t)np * sizeof(int32_t));
            if (!d_match) return LM_ERR_HIP;
            hipLaunchKernelGGL(lm_k_pair_overlap, dim3(LM_HIP_EMULATED ? 2 : 1024), dim3(256), 0, st, s->cc, s->crop, d_cc, d_pairs, np, d_match);
            std::vector<int2> hp((size_t)np);
            std::vector<int32_t> hm((size_t)np);
            LM_HIP(hipMemcpyAsync(hp.data(), d_pairs, (size_t)np * sizeof(int2), hipMemcpyDeviceToHost, st));
            LM_HIP(hipMemcpyAsync(hm.data(), d_match, (size_t)np * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            LM_HIP(hipStreamSynchronize(st));
            std::vector<int32_t> order((size_t)np);
            for (int i = 0; i < np; i++) order[i] = i;
            std::sort(order.begin(), order.end(), [&](int x, int y) { return hp[x].x != hp[y].x ? hp[x].x < hp[y].x : hp[x].y < hp[y].y; });
            g->pair_a.resize(np); g->pair_b.resize(np); g->pair_match.resize(np);
            for (int i = 0; i < np; i++) {
                const int o = order[i];
                g->pair_a[i] = g->stable[hp[o].x]; g->pair_b[i] = g->stable[hp[o].y]; g->pair_match[i] = hm[o];
            }
        }
    }
    tm.mark("self-join + overlaps (device)");
    // per-unique neighbour lists, filled in sorted pair order (== the reference's iteration order)
    struct Tov { int32_t other; double recall, precision; };
    struct Aov { int32_t other, matched, size_other, size_self; };
    const size_t npairs = g->pair_a.size();
    std::vector<uint8_t> pkind(npairs, 0);              // 1: all-overlap only, 2: also inside the time window
    std::vector<double> precall(npairs), pprec(npairs);
    std::vector<int64_t> tov_cnt((size_t)nU + 1, 0), aov_cnt((size_t)nU + 1, 0);
    g->total_intersections = 0;
    for (size_t i = 0; i < npairs; i++) {
        const int a = g->pair_a[i], b = g->pair_b[i];
        const int match = g->pair_match[i];
        const double recall = (double)match / (double)usize(a);         // connected_component.py:239
        const double precision = (double)match / (double)usize(b);      // :240
        precall[i] = recall; pprec[i] = precision;
        if (recall > 0.0 || precision > 0.0) {
            pkind[i] = 1;
            aov_cnt[(size_t)a + 1]++; aov_cnt[(size_t)b + 1]++;
            if (last_frame(a) + g->t_window >= first_frame(b) && last_frame(b) >= first_frame(a) - g->t_window) {
                pkind[i] = 2;
                tov_cnt[(size_t)a + 1]++; tov_cnt[(size_t)b + 1]++;
                g->total_intersections++;
            }
        }
    }
    for (int u = 0; u < nU; u++) { tov_cnt[(size_t)u + 1] += tov_cnt[u]; aov_cnt[(size_t)u + 1] += aov_cnt[u]; }
    std::vector<Tov> tov_flat((size_t)tov_cnt[nU]);
    std::vector<Aov> aov_flat((size_t)aov_cnt[nU]);
    {
        std::vector<int64_t> tp(tov_cnt.begin(), tov_cnt.end() - 1), ap(aov_cnt.begin(), aov_cnt.end() - 1);
        for (size_t i = 0; i < npairs; i++) {
            if (!pkind[i]) continue;
            const int a = g->pair_a[i], b = g->pair_b[i];
            const int sa = usize(a), sb = usize(b);
            const int matched_pixels = (int)((double)sa * precall[i]);   // float64 round trip, can be match-1 (:294)
            aov_flat[(size_t)ap[a]++] = {b, matched_pixels, sb, sa};
            aov_flat[(size_t)ap[b]++] = {a, matched_pixels, sa, sb};
            if (pkind[i] == 2) {
                tov_flat[(size_t)tp[a]++] = {b, precall[i], pprec[i]};
                tov_flat[(size_t)tp[b]++] = {a, pprec[i], precall[i]};
            }
        }
    }
    struct TovRange { const Tov *b, *e; const Tov* begin() const { return b; } const Tov* end() const { return e; } };
    struct AovRange { const Aov *b, *e; const Aov* begin() const { return b; } const Aov* end() const { return e; } };
    auto tov = [&](int u) { return TovRange{tov_flat.data() + tov_cnt[u], tov_flat.data() + tov_cnt[(size_t)u + 1]}; };
    auto aov = [&](int u) { return AovRange{aov_flat.data() + aov_cnt[u], aov_flat.data() + aov_cnt[(size_t)u + 1]}; };
    tm.mark("neighbour lists");
    // ---- compute_groups (:308-413): sequential, order-dependent
    std::vector<std::vector<int32_t>> groups;
    std::vector<int32_t> gid((size_t)nU, -1);
    for (int a : g->stable) {
        int gi;
        if (gid[a] >= 0) gi = gid[a];
        else { gi = (int)groups.size(); groups.push_back({a}); gid[a] = gi; }
        for (const Tov& t : tov(a)) {
            if (t.recall < g->min_recall) continue;
            const int b = t.other;
            if (gid[b] < 0) { gid[b] = gi; groups[gi].push_back(b); }
            else if (gid[b] != gi) {
                const int og = gid[b];
                for (int m : groups[og]) { gid[m] = gi; groups[gi].push_back(m); }
                groups[og].clear();
            }
        }
    }
    g->grp_off.assign(1, 0);
    g->grp_members.clear();
    g->gid_of_unique.assign((size_t)nU, -1);
    for (auto& grp : groups) {
        if (grp.empty()) continue;
        const int ng = (int)g->grp_off.size() - 1;
        for (int m : grp) { g->grp_members.push_back(m); g->gid_of_unique[m] = ng; }
        g->grp_off.push_back((int64_t)g->grp_members.size());
    }
    const int nG = (int)g->grp_off.size() - 1;
    tm.mark("compute_groups");
    // ---- temporal information (:415-444)
    g->ages_off.assign(1, 0); g->ages.clear();
    std::vector<int32_t> g_from((size_t)nG), g_to((size_t)nG);      // live frame range [from, to) of every group
    std::vector<int64_t> fcnt((size_t)F + 2, 0);
    for (int gi = 0; gi < nG; gi++) {
        std::vector<int32_t> a;
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
    // per-frame lists in ascending group index: difference array -> counts -> offsets -> fill
    g->gpf_off.assign((size_t)F + 1, 0);
    {
        int64_t live = 0;
        for (int f = 0; f < F; f++) { live += fcnt[(size_t)f + 1]; g->gpf_off[(size_t)f + 1] = g->gpf_off[f] + live; }
        g->gpf.assign((size_t)g->gpf_off[F], 0);
        std::vector<int64_t> cur(g->gpf_off.begin(), g->gpf_off.end() - 1);
        for (int gi = 0; gi < nG; gi++)
            for (int f = g_from[gi]; f < g_to[gi]; f++) g->gpf[(size_t)cur[f]++] = gi;
    }
    tm.mark("ages + groups_per_frame");
    // ---- conflicts (:446-500); emitted grouped by g1 in first-insertion order of g2
    {
        struct Acc { int64_t matched = 0, unmatched = 0, area_union = 0; double inter = 0; };
        std::vector<std::vector<std::pair<int32_t, Acc>>> conf((size_t)nG);
        std::vector<std::unordered_map<int32_t, int32_t>> idx((size_t)nG);
        auto add = [&](int x, int y, int64_t m, int64_t um, int64_t au, double ai) {
            auto it = idx[x].find(y);
            if (it == idx[x].end()) { idx[x][y] = (int32_t)conf[x].size(); conf[x].push_back({y, Acc()}); it = idx[x].find(y); }
            Acc& a = conf[x][it->second].second;
            a.matched += m; a.unmatched += um; a.area_union += au; a.inter += ai;
        };
        auto area = [&](int u) { return (int64_t)(ubox(u, 1) - ubox(u, 0) + 1) * (ubox(u, 3) - ubox(u, 2) + 1); };
        for (int a : g->stable)
            for (const Aov& t : aov(a)) {
                const int b = t.other;
                if (!(a < b)) continue;
                const int64_t unmatched = (int64_t)t.size_self + t.size_other - (int64_t)t.matched * 2;
                int64_t inter = 0;
                if (ubox(a, 0) <= ubox(b, 1) && ubox(b, 0) <= ubox(a, 1) && ubox(a, 2) <= ubox(b, 3) && ubox(b, 2) <= ubox(a, 3))
                    inter = (int64_t)(std::min(ubox(a, 1), ubox(b, 1)) - std::max(ubox(a, 0), ubox(b, 0)) + 1) *
                            (std::min(ubox(a, 3), ubox(b, 3)) - std::max(ubox(a, 2), ubox(b, 2)) + 1);
                const int64_t uni = area(a) + area(b) - inter;
                const int ga = g->gid_of_unique[a], gb = g->gid_of_unique[b];
                if (ga == gb) continue;
                add(ga, gb, t.matched, unmatched, uni, (double)inter);
                add(gb, ga, t.matched, unmatched, uni, (double)inter);
            }
        g->conf_g1.clear(); g->conf_g2.clear(); g->conf_matched.clear(); g->conf_unmatched.clear(); g->conf_union.clear(); g->conf_inter.clear();
        for (int x = 0; x < nG; x++)
            for (auto& e : conf[x]) {
                g->conf_g1.push_back(x); g->conf_g2.push_back(e.first); g->conf_matched.push_back(e.second.matched);
                g->conf_unmatched.push_back(e.second.unmatched); g->conf_union.push_back(e.second.area_union); g->conf_inter.push_back(e.second.inter);
            }
    }
    tm.mark("conflicts");
    // flatten the neighbour lists
    g->tov_off.assign(1, 0); g->tov_other.clear(); g->tov_recall.clear(); g->tov_precision.clear();
    g->aov_off.assign(1, 0); g->aov_other.clear(); g->aov_matched.clear(); g->aov_size_other.clear(); g->aov_size_self.clear();
    for (int u = 0; u < nU; u++) {
        for (const Tov& t : tov(u)) { g->tov_other.push_back(t.other); g->tov_recall.push_back(t.recall); g->tov_precision.push_back(t.precision); }
        g->tov_off.push_back((int64_t)g->tov_other.size());
        for (const Aov& t : aov(u)) { g->aov_other.push_back(t.other); g->aov_matched.push_back(t.matched); g->aov_size_other.push_back(t.size_other); g->aov_size_self.push_back(t.size_self); }
        g->aov_off.push_back((int64_t)g->aov_other.size());
    }
    tm.mark("flatten neighbour lists");
    // ---- group images (:575-636)
    std::vector<int32_t> ulist_frame(g->ulist_cc.size());
    for (size_t e = 0; e < g->ulist_cc.size(); e++) ulist_frame[e] = cc_frame[(size_t)g->ulist_cc[e]];
    tm.mark("  gimg: entry frames");
    g->bounds.assign((size_t)nG * 4, 0);
    std::vector<LmGimgItem> items;
    std::vector<LmGimgMember> members;
    std::vector<LmGimgUnit> units;
    g->gimg_off.assign(1, 0);
    g->gbits_off.clear();
    long long bit_words = 0;
    g->gimg_item_off.assign(1, 0);
    // Groups are independent here: contiguous chunks of groups are tabulated by a few host threads into local tables
    // (offsets relative to the chunk), then concatenated in group order.
    struct Chunk {
        std::vector<LmGimgItem> items; std::vector<LmGimgMember> members; std::vector<LmGimgUnit> units;
        std::vector<int64_t> item_end;          // items of the chunk up to and including each group
        long long img_bytes = 0, bit_words = 0;
    };
    const int n_chunks = std::max(1, std::min(LM_GROUP_THREADS, nG / 64));
    std::vector<Chunk> chunks((size_t)n_chunks);
    auto tabulate = [&](int ci) {
        Chunk& ck = chunks[(size_t)ci];
        std::vector<int32_t> seg_cnt;
        const int g_lo = (int)((long long)nG * ci / n_chunks), g_hi = (int)((long long)nG * (ci + 1) / n_chunks);
        for (int gi = g_lo; gi < g_hi; gi++) {
            int x0 = 1 << 30, x1 = -1, y0 = 1 << 30, y1 = -1;
            for (int64_t i = g->grp_off[gi]; i < g->grp_off[(size_t)gi + 1]; i++) {
                const int u = g->grp_members[(size_t)i];
                x0 = std::min(x0, (int)ubox(u, 0)); x1 = std::max(x1, (int)ubox(u, 1));
                y0 = std::min(y0, (int)ubox(u, 2)); y1 = std::max(y1, (int)ubox(u, 3));
            }
            g->bounds[(size_t)gi * 4 + 0] = x0; g->bounds[(size_t)gi * 4 + 1] = x1; g->bounds[(size_t)gi * 4 + 2] = y0; g->bounds[(size_t)gi * 4 + 3] = y1;
            const int w = x1 - x0 + 1, h = y1 - y0 + 1;
            // entries of every member inside every segment [ages[j], ages[j+1]] (both ends included, duplicates counted, :619):
            // a unique's entries and the ages are ascending, so one two-pointer walk per member serves all segments
            const int nm = (int)(g->grp_off[(size_t)gi + 1] - g->grp_off[gi]);
            const int ns = (int)(g->ages_off[(size_t)gi + 1] - g->ages_off[gi]) - 1;
            const int32_t* ag = g->ages.data() + g->ages_off[gi];
            seg_cnt.assign((size_t)std::max(ns, 0) * nm, 0);
            for (int mi = 0; mi < nm && ns > 0; mi++) {
                const int u = g->grp_members[(size_t)g->grp_off[gi] + mi];
                const int32_t* fe = ulist_frame.data() + g->ulist_off[(size_t)u + 1];
                const int32_t *lo = ulist_frame.data() + g->ulist_off[u], *hi = lo;
                for (int j = 0; j < ns; j++) {
                    while (lo < fe && *lo < ag[j]) lo++;
                    if (hi < lo) hi = lo;
                    while (hi < fe && *hi <= ag[j + 1]) hi++;
                    seg_cnt[(size_t)j * nm + mi] = (int32_t)(hi - lo);
                }
            }
            for (int j = 0; j < ns; j++) {
                LmGimgItem it;
                it.x0 = x0; it.y0 = y0; it.w = w; it.h = h;
                it.mem_off = (int32_t)ck.members.size();
                for (int mi = 0; mi < nm; mi++) {
                    const int count = seg_cnt[(size_t)j * nm + mi];
                    if (count) ck.members.push_back({g->uniq_cc[g->grp_members[(size_t)g->grp_off[gi] + mi]], count});
                }
                it.mem_cnt = (int32_t)ck.members.size() - it.mem_off;
                it.img_off = ck.img_bytes;
                it.bits_off = ck.bit_words;
                ck.img_bytes += (long long)w * h;
                ck.bit_words += (long long)h * ((w + 31) >> 5);
                const int item_idx = (int)ck.items.size();
                ck.items.push_back(it);
                for (int ty = 0; ty * LM_GT < h; ty++)
                    for (int tx = 0; tx * LM_GT < w; tx++) ck.units.push_back({item_idx, (int16_t)tx, (int16_t)ty});
            }
            ck.item_end.push_back((int64_t)ck.items.size());
        }
    };
    tm.mark("  gimg: setup");
    {
        std::vector<std::thread> workers;
        for (int ci = 1; ci < n_chunks; ci++) workers.emplace_back(tabulate, ci);
        tabulate(0);
        for (auto& t : workers) t.join();
    }
    tm.mark("  gimg: threads");
    for (const Chunk& ck : chunks) {
        const int64_t item0 = (int64_t)items.size();
        const int32_t mem0 = (int32_t)members.size();
        const long long img0 = g->gimg_off.back();
        for (LmGimgItem it : ck.items) {
            it.mem_off += mem0; it.img_off += img0; it.bits_off += bit_words;
            g->gbits_off.push_back(it.bits_off);
            g->gimg_off.push_back(it.img_off + (int64_t)it.w * it.h);
            items.push_back(it);
        }
        members.insert(members.end(), ck.members.begin(), ck.members.end());
        for (LmGimgUnit un : ck.units) { un.item += (int32_t)item0; units.push_back(un); }
        for (int64_t e : ck.item_end) g->gimg_item_off.push_back(item0 + e);
        bit_words += ck.bit_words;
    }
    if (tm.on) fprintf(stderr, "[lm_group]   groups %d items %zu members %zu units %zu chunks %d pairs %zu\n", nG, items.size(), members.size(), units.size(), n_chunks, g->pair_a.size());
    tm.mark("  gimg: item tables");
    const long long img_bytes = g->gimg_off.back();
    g->d_images = (uint8_t*)lm_galloc(g, (size_t)std::max<long long>(img_bytes, 1));
    g->d_gbits = (uint32_t*)lm_galloc(g, (size_t)std::max<long long>(bit_words, 1) * sizeof(uint32_t));
    if (!g->d_images || !g->d_gbits) return LM_ERR_HIP;
    if (!items.empty()) {
        LmGimgItem* d_items; LmGimgMember* d_members; LmGimgUnit* d_units; int32_t* d_max;
        if (lm_upload(g, items, &d_items, st) || lm_upload(g, members, &d_members, st) || lm_upload(g, units, &d_units, st)) return LM_ERR_HIP;
        tm.mark("  gimg: uploads");
        d_max = (int32_t*)lm_galloc(g, items.size() * sizeof(int32_t));
        if (!d_max) return LM_ERR_HIP;
        LM_HIP(hipMemsetAsync(d_max, 0, items.size() * sizeof(int32_t), st));
        const int nb = (int)std::min<size_t>(units.size(), LM_HIP_EMULATED ? 2 : 4096);
        hipLaunchKernelGGL(lm_k_gimg_max, dim3(nb), dim3(256), 0, st, d_items, d_units, (int)units.size(), d_members, s->cc, s->crop, d_max);
        hipLaunchKernelGGL(lm_k_gimg_write, dim3(nb), dim3(256), 0, st, d_items, d_units, (int)units.size(), d_members, s->cc, s->crop,
                           d_max, g->img_thr, g->d_images, g->d_gbits);
        LM_HIP(hipGetLastError());
    }
    tm.mark("group images (host tables + device)");
    // ---- render tables for frames_from_groups (:638-681): one item per (frame, live group), built on the device
    if (reconstruct_tables) {
        std::vector<long long> fio(g->gpf_off.begin(), g->gpf_off.end());
        std::vector<int32_t> ages_off32(g->ages_off.begin(), g->ages_off.end());
        std::vector<int64_t> gitem_first(g->gimg_item_off.begin(), g->gimg_item_off.end() - 1);
        int32_t *d_gpf, *d_ages, *d_ages_off, *d_bounds;
        int64_t *d_gitem_first, *d_gbits_off;
        const size_t n_ritems = g->gpf.size();
        g->d_render_items = (LmRenderItem*)lm_galloc(g, std::max<size_t>(n_ritems, 1) * sizeof(LmRenderItem));
        if (!g->d_render_items) return LM_ERR_HIP;
        if (lm_upload(g, fio, &g->d_frame_item_off, st)) return LM_ERR_HIP;
        if (n_ritems) {
            if (lm_upload(g, g->gpf, &d_gpf, st) || lm_upload(g, g->ages, &d_ages, st) || lm_upload(g, ages_off32, &d_ages_off, st) ||
                lm_upload(g, g->bounds, &d_bounds, st) || lm_upload(g, gitem_first, &d_gitem_first, st) ||
                lm_upload(g, g->gbits_off, &d_gbits_off, st))
                return LM_ERR_HIP;
            hipLaunchKernelGGL(lm_k_render_items, dim3((unsigned)((n_ritems + 255) / 256)), dim3(256), 0, st, g->d_frame_item_off, F, d_gpf,
                               d_ages, d_ages_off, d_bounds, d_gitem_first, d_gbits_off, g->d_render_items);
        }
        LM_HIP(hipStreamSynchronize(st));      // the uploaded vectors are locals: the async copies must finish before they go away
    }
    LM_HIP(hipStreamSynchronize(st));
    tm.mark("render tables + final sync");
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
            const size_t first = (size_t)64 << 20;
            if (hipMalloc(&s->garena, first) == hipSuccess) s->garena_bytes = first;
        }
        if (s->garena) { g->arena = (char*)s->garena; g->arena_cap = s->garena_bytes; g->arena_cached = true; s->garena_busy = 1; }
    }
    g->max_gap = max_gap; g->min_times = min_times; g->t_window = t_window;
    g->min_recall = min_recall; g->img_thr = img_threshold;
    if (lm_group_run_impl(g, reconstruct_tables, (hipStream_t)stream) != LM_OK) {
        lm_group_destroy(g);
        return nullptr;
    }
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
            if (g->gimg_host.size() != (size_t)g->gimg_off.back()) {
                g->gimg_host.resize((size_t)g->gimg_off.back());
                if (!g->gimg_host.empty())
                    LM_HIP(hipMemcpy(g->gimg_host.data(), g->d_images, g->gimg_host.size(), hipMemcpyDeviceToHost));
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
