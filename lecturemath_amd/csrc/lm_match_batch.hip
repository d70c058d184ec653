// lm_match_batch.hip -- temporal CC matching for a whole batch of frames in a constant number of launches.
//
// Replaces (paths relative to /root/reference/ACCESS2021_release) CCStabilityEstimator.add_frame,
// content/cc_stability_estimator.py:41-155, for frames f0 .. f0+B-1 at once.  What makes that legal:
//   * whether a current CC c and a unique u "match" (getOverlapFMeasure + the recall/precision thresholds, :92-99) is a
//     function of the two bit crops alone (the unique keeps its FIRST-seen CC, :110-123) -- no stream state involved;
//   * the state only decides WHICH uniques are candidates at frame f: those that exist (created in an earlier frame) and
//     have not been retired (:126-145), and the first accepted one in ascending unique index wins (:84-108).
//   * a CC that is a bit-identical TWIN (box, size, crop) of an earlier CC of the batch (its root), at most max_gap frames
//     before it, is assigned the unique its root was assigned: every smaller-index candidate it accepts was dead or rejected
//     at the root's frame already (a dead unique never comes back), and the root's unique was matched there, so it is alive
//     (identical crops are accepted: recall = precision = 1, thresholds <= 1).  Static video: almost every CC is a twin;
//   * tempo_count (:85), the number of (cur, candidate unique) box pairs, does not depend on the order of events: u is a
//     candidate at frame f iff it was born before f and alive(f, last_u at the END of the batch) -- a unique that is matched
//     later was alive all along, a dead one is never matched again.
// So the expensive part (box join + AND/popcount on the crops) runs in wide launches over the NON-TWIN CCs of all B frames,
// and only a cheap replay of the decisions -- list lookups, no pixels -- is sequential over the frames, in ONE workgroup:
//   T  lm_k_mb_twin_*     hash table (key -> smallest CC index of the batch), candidates verified word by word; a missed twin
//                         only costs time
//   N  lm_k_mb_nt         drop the actives retired before f0 (order kept), compact the non-twin CCs (NT), tile tables (64 NT
//                         CCs per tile, tiles never straddle a frame), reset per-batch tables
//   A  lm_k_mb_join<0,*>  (NT tile) x (uniques active at f0): count pass, then fill pass at exact offsets; the sources of a
//                         tile are shared by gridDim.y workgroups, one counter update / slot reservation per workgroup
//   E  lm_k_mb_eval<0>    every pair, 8 lanes each: overlap + float64 thresholds -> accepted bit; a CC with an accepted
//                         candidate that stays alive whatever happens in the batch is "surely matched"
//   S  lm_k_mb_sources    the other NT CCs are the only ones that CAN become new uniques: compact them (S)
//   B  lm_k_mb_join<1,*> + lm_k_mb_eval<1>: the same join against S, restricted to sources of EARLIER frames
//   C  lm_k_mb_resolve    frames in order: smallest accepted active position per NT CC among the uniques that exist and are
//                         alive (ascending unique index == the reference's first match) or a new unique, numbered in CC
//                         order; a twin copies its root's position; `last` and the active list are updated as the
//                         reference does.  In-batch uniques get positions behind all earlier ones in creation order, so
//                         "smallest position" is still "smallest unique index"
//   Q  lm_k_mb_tempo      streaming box join of ALL CCs against the final active list with the tempo_count rule above
#include "lm_stream.h"

#define LM_MB_TILE 64
#define LM_MB_JY 8              // workgroups that share the sources of one tile in the join launches (gridDim.y, at most)
#define LM_MB_TTAB (1 << 18)     // twin table slots; batches with more than LM_MB_TTAB / 2 CCs skip twin detection
#ifndef LM_MB_CHUNK
#define LM_MB_CHUNK 1024        // source boxes a join workgroup filters per round (LDS survivors list: 12 KB).  A workgroup's share of the sources is
                                // 1 / LM_MB_JY of the list (a few hundred); round 2's 4096 (48 KB) kept the joins' workgroups waiting for LDS beside the
                                // other kernels of the pipeline
#endif
#define LM_MB_MAX_FRAMES 64     // frames per batch (per-frame tables of the replay kernel live in LDS)
#define LM_MB_BIGPAIR 2048       // words of a box intersection above which a pair is evaluated by a whole workgroup
#ifndef LM_MB_CH
#define LM_MB_CH 2048           // non-twin CCs of one frame resolved per LDS pass
#endif
#define LM_MB_LA 16384          // active positions whose last-matched frame is cached in LDS by the replay kernel
#define LM_MB_LS 16384          // sources whose active position is cached in LDS by the replay kernel
#ifndef LM_MB_PFA
#define LM_MB_PFA 2             // pairs per thread prefetched one frame ahead: against earlier uniques ...
#endif
#ifndef LM_MB_PFB
#define LM_MB_PFB 1             // ... and against in-batch sources
#endif
#define LM_MB_RESOLVE_SMEM ((size_t)LM_MB_CH * 8 + (size_t)LM_MB_LA * 4 + (size_t)LM_MB_LS * 4)

struct LmMatchBatch {
    int32_t* ftile;             // [cap_frames + 2] first tile (64 NON-TWIN CCs) of every frame of the batch; ftile[B] = number of tiles
    int32_t* ftile_all;         // [cap_frames + 2] the same over ALL CCs (lm_k_mb_tempo)
    int32_t* nt_foff;           // [cap_frames + 2] index in nt_list of the first non-twin CC of every frame; nt_foff[B] = count
    int32_t* nt_list;           // [cap_cc] global cc index of non-twin k (ascending)
    int32_t* cls;               // [cap_cc] per global cc: k >= 0 own index in nt_list; -1 - k: twin whose root is non-twin k
    int32_t* troot;             // [cap_cc] per global cc: global cc index of the twin's root (valid where twin[] is set)
    int32_t* rootpos;           // [cap_cc] scratch per non-twin k: sources before k (lm_k_mb_sources -> s_prefix)
    int32_t* nt_cnt;            // [LM_MB_MAX_FRAMES] non-twins per frame of the batch (zeroed by lm_k_mb_twin_insert, counted by lm_k_mb_twin_find)
    int32_t* nt_src;            // [cap_cc] per non-twin k: its index in the source list, or -1 (copy of sidx in nt order)
    int32_t* tlast;             // [cap_cc] per non-twin k: last frame of the batch in which one of its twins appears, or -1
    int32_t* s_prefix;          // [cap_frames + 2] number of sources that belong to frames before frame b of the batch
    uint32_t* tcount[2];        // [cap_tiles + 1]  pairs per tile (A: vs actives, B: vs in-batch sources)
    uint32_t* toff[2];          // [cap_tiles + 1]  exclusive prefix, toff[nt] = total
    uint32_t* tcur[2];          // [(cap_tiles + 1) * LM_MB_JY]  hits of every (tile, blockIdx.y) workgroup of the count launch
    uint2* pairs[2];            // x = index in nt_list | accepted << 31, y = active position (A) / source index (B)
    int32_t* pair_u[2];         // cc index of the unique's first-seen CC (what the pair is evaluated against)
    int32_t* sidx;              // [cap_cc] per global cc: -2 undecided, -1 surely matched; after lm_k_mb_sources: k >= 0 index into
                                // the source list for the sources, still negative for everybody else
    int32_t* s_list;            // [cap_cc] cc index of source k (ascending)
    unsigned long long* s_box;  // packed box of source k
    int32_t* newpos;            // active position source k was given when it became a unique, else -1
    int32_t* n_src;             // [1]
    unsigned long long* ttab;   // [LM_MB_TTAB] twin table: key32 << 32 | smallest batch-relative cc index with that key
    uint32_t* tkey;             // [cap_cc] per global cc: key32 of (box, size, crop) (lm_k_mb_twin_insert); then the twin-table slot of that key (lm_k_mb_twin_probe)
    uint8_t* twin;              // [cap_cc] per global cc: 1 = exact twin of an earlier CC of the batch
    uint32_t* big[2];           // [cap_big] pairs whose pixel intersection is left to a whole workgroup (lm_k_mb_eval_big)
    unsigned* n_big;            // [2] their number (zeroed by lm_k_mb_twin_insert / lm_k_mb_nt)
    uint32_t cap_big;
    uint32_t cap_pairs;
    int cap_tiles;
};

LM_DEV bool lm_mb_alive(int f, int last, int max_gap) { return f <= 1 || (f - 1) - last < max_gap; }

// Box test of the join inner loops on packed 16-bit lanes.  A CC's box goes in as (x0, -x1 | y0, -y1), a source's as (x1, -x0 | y1, -y0):
// the boxes intersect iff all four 16-bit differences source - cc are >= 0 (x1' - x0 >= 0, x1 - x0' >= 0, the same in y) -- two packed
// subtractions, an OR and a sign test instead of unpacking eight fields (the loops were VALU bound: 28 of lm_k_mb_tempo's 54 us).
// Coordinates are int16 >= 0, so no difference overflows.
typedef short lm_s16x2 __attribute__((ext_vector_type(2)));
struct alignas(16) lm_u64x2 { unsigned long long x, y; };
LM_DEV unsigned long long lm_box_as_cc(unsigned long long b)
{
    const unsigned x0 = (unsigned)(b & 0xffff), x1 = (unsigned)((b >> 16) & 0xffff), y0 = (unsigned)((b >> 32) & 0xffff), y1 = (unsigned)(b >> 48);
    return (unsigned long long)(x0 | ((0u - x1) << 16)) | ((unsigned long long)(y0 | ((0u - y1) << 16)) << 32);
}
LM_DEV unsigned long long lm_box_as_src(unsigned long long b)
{
    const unsigned x0 = (unsigned)(b & 0xffff), x1 = (unsigned)((b >> 16) & 0xffff), y0 = (unsigned)((b >> 32) & 0xffff), y1 = (unsigned)(b >> 48);
    return (unsigned long long)(x1 | ((0u - x0) << 16)) | ((unsigned long long)(y1 | ((0u - y0) << 16)) << 32);
}
LM_DEV bool lm_box_hit_pk(unsigned long long c, unsigned long long u)
{
    const lm_s16x2 dx = __builtin_bit_cast(lm_s16x2, (unsigned)u) - __builtin_bit_cast(lm_s16x2, (unsigned)c);
    const lm_s16x2 dy = __builtin_bit_cast(lm_s16x2, (unsigned)(u >> 32)) - __builtin_bit_cast(lm_s16x2, (unsigned)(c >> 32));
    return ((__builtin_bit_cast(unsigned, dx) | __builtin_bit_cast(unsigned, dy)) & 0x80008000u) == 0u;
}
// a source entry no CC box intersects (x0 = 32767 is right of every box narrower than that)
#define LM_BOX_SRC_NEVER ((unsigned long long)(0u | ((0u - 32767u) << 16)) | ((unsigned long long)(0u | ((0u - 32767u) << 16)) << 32))

// ------------------------------------------------------------------------------------------------
// P: first part of lm_k_mb_nt (one block).
// ------------------------------------------------------------------------------------------------
LM_DEV void lm_mb_prologue(const long long* __restrict__ frame_cc_off, int f0, int B,
                                                         int32_t* __restrict__ active, int32_t* __restrict__ active_cc,
                                                         unsigned long long* __restrict__ active_box, int32_t* __restrict__ active_last,
                                                         LmCounters* __restrict__ cnt, LmMatchBatch mb, int max_gap)
{
    if (cnt->error) return;
    // ---- drop actives that cannot be candidates at f0 (order preserved)
    int nA = cnt->n_active;
    if (f0 > 1) {
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
                    keep[k] = lm_mb_alive(f0, ul[k], max_gap) ? 1u : 0u;
                }
                mine += keep[k];
            }
            unsigned tot;
            unsigned ex = lm_block_excl_scan<1024>(mine, &tot);   // barriers inside: this pass's reads precede its writes
            unsigned o = kept + ex;
#pragma unroll
            for (int k = 0; k < LM_UPD_ITEMS; k++)
                if (keep[k]) { active[o] = u[k]; active_cc[o] = uc[k]; active_box[o] = ub[k]; active_last[o] = ul[k]; o++; }
            kept += tot;
        }
        nA = (int)kept;
    }
    if (threadIdx.x == 0) {
        *mb.n_src = 0;
        cnt->n_active = nA;
    }
}

// ------------------------------------------------------------------------------------------------
// N: compact the non-twin CCs (the only ones that go through joins, evaluation and pair replay) and build the tile tables.
// B + 1 workgroups: workgroup b < B compacts frame b -- where its non-twins and tiles start follows from the per-frame
// non-twin counts lm_k_mb_twin_find left in nt_cnt (twins == 0: no twin detection ran, every CC is a non-twin; the counters
// cost lm_k_mb_twin_find +11 us, counting the flags of the earlier frames here was measured at +17 us) -- and the last
// workgroup compacts the active list (lm_mb_prologue).
// ------------------------------------------------------------------------------------------------
#define LM_MB_NT_R 4

__global__ void __launch_bounds__(1024) lm_k_mb_nt(const long long* __restrict__ frame_cc_off, int f0, int B, int32_t* __restrict__ active,
                                                   int32_t* __restrict__ active_cc, unsigned long long* __restrict__ active_box,
                                                   int32_t* __restrict__ active_last, LmCounters* __restrict__ cnt, LmMatchBatch mb, int max_gap,
                                                   int twins)
{
    __shared__ unsigned s_tab[LM_MB_NT_R * 16];
    __shared__ unsigned s_tot;
    __shared__ int s_head[6];
    if (cnt->error) return;
    if ((int)blockIdx.x == B) {
        if (threadIdx.x < 2) mb.n_big[threadIdx.x] = 0;        // filled by lm_k_mb_eval
        lm_mb_prologue(frame_cc_off, f0, B, active, active_cc, active_box, active_last, cnt, mb, max_gap);
        return;
    }
    const int b = (int)blockIdx.x;
    const int lane = lm_lane(), wid = (int)(threadIdx.x >> 6);
    if (wid == 0) {     // frames before this one (B <= 64: one per lane): non-twins, tiles of non-twins, tiles of all CCs
        unsigned n_nt = 0, n_all = 0;
        if (lane < B) {
            n_all = (unsigned)(frame_cc_off[f0 + lane + 1] - frame_cc_off[f0 + lane]);
            n_nt = twins ? (unsigned)mb.nt_cnt[lane] : n_all;
        }
        const unsigned t_nt = (n_nt + LM_MB_TILE - 1) / LM_MB_TILE, t_all = (n_all + LM_MB_TILE - 1) / LM_MB_TILE;
        const unsigned i_nt = lm_wave_incl_scan(n_nt), i_tnt = lm_wave_incl_scan(t_nt), i_tall = lm_wave_incl_scan(t_all);
        if (lane == b) {
            s_head[0] = (int)(i_nt - n_nt); s_head[1] = (int)(i_tnt - t_nt); s_head[2] = (int)i_tnt;
            mb.nt_foff[b] = (int32_t)(i_nt - n_nt);
            mb.ftile[b] = (int32_t)(i_tnt - t_nt);
            mb.ftile_all[b] = (int32_t)(i_tall - t_all);
            if (b == B - 1) {
                mb.nt_foff[B] = (int32_t)i_nt;
                mb.ftile[B] = (int32_t)i_tnt;
                mb.ftile_all[B] = (int32_t)i_tall;
                if ((int)i_tnt > mb.cap_tiles || (int)i_tall > mb.cap_tiles) cnt->error = LM_ERR_CAPACITY;
            }
        }
    }
    __syncthreads();
    const int t0 = s_head[1], t1 = s_head[2] + (b == B - 1 ? 1 : 0);      // this frame's tiles (the last frame also clears entry [n_tiles])
    for (int t = t0 + (int)threadIdx.x; t < t1 && t <= mb.cap_tiles; t += 1024) { mb.tcount[0][t] = 0; mb.tcount[1][t] = 0; }
    const long long C0 = frame_cc_off[f0 + b], C1 = frame_cc_off[f0 + b + 1];
    unsigned carry = (unsigned)s_head[0];
    for (long long base = C0; base < C1; base += 1024 * LM_MB_NT_R) {
        unsigned fm = 0;
#pragma unroll
        for (int k = 0; k < LM_MB_NT_R; k++) {
            const long long i = base + (long long)k * 1024 + threadIdx.x;
            const int flag = (i < C1) ? (mb.twin[i] == 0) : 0;
            fm |= (unsigned)flag << k;
            const unsigned long long bal = __ballot(flag);
            if (lane == 0) s_tab[k * 16 + wid] = (unsigned)__popcll(bal);
        }
        __syncthreads();
        if (wid == 0) {     // exclusive scan of the (row, wave) counts: one per lane
            const unsigned v = s_tab[lane];
            const unsigned incl = lm_wave_incl_scan(v);
            s_tab[lane] = incl - v;
            if (lane == 63) s_tot = incl;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LM_MB_NT_R; k++) {
            const int flag = (int)((fm >> k) & 1u);
            const unsigned long long bal = __ballot(flag);
            const long long i = base + (long long)k * 1024 + threadIdx.x;
            if (flag) {
                const unsigned o = carry + s_tab[k * 16 + wid] + (unsigned)__popcll(bal & lm_lowmask_excl(lane));
                mb.nt_list[o] = (int32_t)i; mb.cls[i] = (int32_t)o; mb.tlast[o] = -1; mb.sidx[i] = -2;
            }
        }
        carry += s_tot;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// A / B: box join of one tile of CCs against a source list.  SRC 0: actives; SRC 1: in-batch sources of earlier frames
// (a prefix of the ascending source list).  256 threads = 64 CCs x 4 quarters of the surviving sources.  Sources are first
// filtered against the tile's union box (CCs are in raster order of their first pixel, so 64 consecutive ones cover a
// thin strip of the frame).
// ------------------------------------------------------------------------------------------------
// largest b < B with ftile[b] <= t (frames without CCs own no tile and are skipped).  B <= 64: one entry per lane and a ballot -- one
// memory latency instead of the six of a binary search (every wave of the workgroup calls; ftile[0] == 0 <= t)
LM_DEV int lm_mb_tile_frame(const int32_t* __restrict__ ftile, int B, int t)
{
    const int lane = lm_lane();
    const unsigned long long le = __ballot(lane < B && ftile[lane] <= t);
    return 63 - __clzll((long long)le);
}

template <int SRC, int FILL>
__global__ void __launch_bounds__(256) lm_k_mb_join(const LmCcRec* __restrict__ cc, const long long* __restrict__ frame_cc_off, int f0,
                                                    int B, const unsigned long long* __restrict__ active_box,
                                                    const int32_t* __restrict__ active_cc, LmCounters* __restrict__ cnt, LmMatchBatch mb)
{
    __shared__ __attribute__((aligned(16))) unsigned long long s_sbox[LM_MB_CHUNK + 4];
    __shared__ int s_spos[LM_MB_CHUNK];
    __shared__ int s_ub[4];
    __shared__ int s_nsurv;
    __shared__ unsigned s_count, s_base;
    if (cnt->error) return;
    const int nt = mb.ftile[B];
    const unsigned long long* src_box = SRC == 0 ? active_box : mb.s_box;
    const int32_t* src_cc = SRC == 0 ? active_cc : mb.s_list;
    const int ccl = (int)(threadIdx.x & 63), q = (int)(threadIdx.x >> 6);
    for (int t = blockIdx.x; t < nt; t += gridDim.x) {
        const int b = lm_mb_tile_frame(mb.ftile, B, t);
        const int n_all = SRC == 0 ? cnt->n_active : mb.s_prefix[b];
        // this workgroup's share of the sources: tiles of non-twins are few and scattered over the frame, so the work of one
        // tile is spread over gridDim.y workgroups (one counter update / one slot reservation per workgroup)
        const int s_lo = (int)((long long)n_all * blockIdx.y / gridDim.y), s_hi = (int)((long long)n_all * (blockIdx.y + 1) / gridDim.y);
        const int k_first = mb.nt_foff[b] + (t - mb.ftile[b]) * LM_MB_TILE;          // first entry of nt_list in this tile
        const int ncc = (mb.nt_foff[b + 1] - k_first < LM_MB_TILE) ? mb.nt_foff[b + 1] - k_first : LM_MB_TILE;
        __syncthreads();        // previous tile done with the shared scalars
        if (threadIdx.x == 0) { s_ub[0] = 0x7fff; s_ub[1] = -1; s_ub[2] = 0x7fff; s_ub[3] = -1; s_count = 0; }
        unsigned off = 0;
        if (FILL) {
            // exclusive prefix of the tile counts, recomputed by every owner (a few hundred tiles per batch at most)
            unsigned part = 0;
            for (int i = threadIdx.x; i < t; i += 256) part += mb.tcount[SRC][i];
            unsigned tot;
            (void)lm_block_excl_scan<256>(part, &tot);
            off = tot;
        }
        __syncthreads();
        unsigned long long mybox = 0;
        const bool have = ccl < ncc;
        const long long my_cc = have ? (long long)mb.nt_list[k_first + ccl] : 0;
        if (have) {
            const LmCcRec r = cc[my_cc];
            mybox = lm_box_as_cc(lm_pack_box(r));
            if (q == 0) {
                atomicMin(&s_ub[0], (int)r.min_x); atomicMax(&s_ub[1], (int)r.max_x);
                atomicMin(&s_ub[2], (int)r.min_y); atomicMax(&s_ub[3], (int)r.max_y);
            }
        }
        __syncthreads();
        const unsigned long long ubox = (unsigned long long)(unsigned short)s_ub[0] | ((unsigned long long)(unsigned short)s_ub[1] << 16) |
                                        ((unsigned long long)(unsigned short)s_ub[2] << 32) | ((unsigned long long)(unsigned short)s_ub[3] << 48);
        const unsigned tc = FILL ? mb.tcount[SRC][t] : 0u;
        if (FILL && threadIdx.x == 0 && blockIdx.y == 0) {
            if ((unsigned long long)off + tc > mb.cap_pairs) cnt->error = LM_ERR_CAPACITY;
            mb.toff[SRC][t] = off;
            if (t == nt - 1) mb.toff[SRC][nt] = off + tc;
        }
        const bool room = !FILL || (unsigned long long)off + tc <= mb.cap_pairs;
        // where this workgroup's pairs go inside the tile's range: behind those of the workgroups with a smaller blockIdx.y, whose hit
        // counts the count launch left in tcur[] (round 2 counted them again here to reserve the slots: the hit loop ran twice)
        if (FILL && threadIdx.x == 0) {
            unsigned before = 0;
            for (unsigned yy = 0; yy < blockIdx.y; yy++) before += mb.tcur[SRC][(size_t)t * LM_MB_JY + yy];
            s_base = before;
        }
        {
            const int pass = 1;
            unsigned mycount = 0;
            for (int base = s_lo; base < s_hi; base += LM_MB_CHUNK) {
                __syncthreads();
                if (threadIdx.x == 0) s_nsurv = 0;
                __syncthreads();
#pragma unroll 4
                for (int k = 0; k < LM_MB_CHUNK / 256; k++) {
                    const int i = base + k * 256 + (int)threadIdx.x;
                    if (i < s_hi) {
                        const unsigned long long sb = src_box[i];
                        if (lm_box_hit_packed(ubox, sb)) {
                            const int slot = atomicAdd(&s_nsurv, 1);
                            s_sbox[slot] = lm_box_as_src(sb);
                            s_spos[slot] = i;
                        }
                    }
                }
                __syncthreads();
                const int ns = s_nsurv;
                if (!FILL) {        // counting: four entries per trip, the list padded with entries nothing intersects (as lm_k_mb_tempo)
                    if (threadIdx.x < 4) s_sbox[ns + (int)threadIdx.x] = LM_BOX_SRC_NEVER;
                    __syncthreads();
                    if (have)
                        for (int j = q * 4; j < ns; j += 16) {
                            const lm_u64x2 e01 = *(const lm_u64x2*)&s_sbox[j], e23 = *(const lm_u64x2*)&s_sbox[j + 2];
                            mycount += (lm_box_hit_pk(mybox, e01.x) ? 1u : 0u) + (lm_box_hit_pk(mybox, e01.y) ? 1u : 0u) +
                                       (lm_box_hit_pk(mybox, e23.x) ? 1u : 0u) + (lm_box_hit_pk(mybox, e23.y) ? 1u : 0u);
                        }
                    continue;
                }
                // the wave's trip count is uniform (q and ns are), so the hits of one trip can share one LDS slot allocation
                for (int j = q; j < ns; j += 4) {
                    const bool hit = have && lm_box_hit_pk(mybox, s_sbox[j]);
                    if (FILL && pass == 1) {
                        const unsigned long long bal = __ballot(hit);
                        if (bal) {
                            unsigned basep = 0;
                            if (lm_lane() == 0) basep = atomicAdd(&s_count, (unsigned)__popcll(bal));
                            basep = (unsigned)__shfl((int)basep, 0);
                            if (hit && room) {
                                const unsigned slot = off + s_base + basep + (unsigned)__popcll(bal & lm_lowmask_excl(lm_lane()));
                                const int pos = s_spos[j];
                                mb.pairs[SRC][slot] = make_uint2((unsigned)(k_first + ccl), (unsigned)pos);
                                mb.pair_u[SRC][slot] = src_cc[pos];
                            }
                        }
                    } else {
                        mycount += hit ? 1u : 0u;
                    }
                }
            }
            if (!FILL) {
                if (mycount) atomicAdd(&s_count, mycount);
                __syncthreads();
                if (threadIdx.x == 0) {
                    if (s_count) atomicAdd(&mb.tcount[SRC][t], s_count);
                    mb.tcur[SRC][(size_t)t * LM_MB_JY + blockIdx.y] = s_count;
                }
                __syncthreads();
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// E: evaluate every pair (8 lanes each).
// ------------------------------------------------------------------------------------------------
template <int SRC>
__global__ void __launch_bounds__(256) lm_k_mb_eval(const LmCcRec* __restrict__ cc, const uint32_t* __restrict__ crop,
                                                    const long long* __restrict__ frame_cc_off, int f0, int B,
                                                    const int32_t* __restrict__ active_last, LmCounters* __restrict__ cnt,
                                                    LmMatchBatch mb, double min_recall, double min_precision, int max_gap)
{
    if (cnt->error) return;
    const int nt = mb.ftile[B];
    const unsigned total = nt > 0 ? mb.toff[SRC][nt] : 0u;
    const long long C0 = frame_cc_off[f0];
    if (SRC == 0) {
        // side job of the first evaluation pass (a wide launch that does not read cls): twins refer to their root's index in
        // nt_list (roots are non-twins, their cls was written by lm_k_mb_nt)
        const long long C1 = frame_cc_off[f0 + B];
        for (long long i = C0 + (long long)blockIdx.x * 256 + threadIdx.x; i < C1; i += (long long)gridDim.x * 256)
            if (mb.twin[i]) {
                const int rk = mb.cls[mb.troot[i]];
                mb.cls[i] = -1 - rk;
                atomicMax(&mb.tlast[rk], cc[i].frame);          // the root's unique is matched again at every twin's frame
            }
    }
    const int sub = (int)(threadIdx.x & 7);
    const unsigned grp = (blockIdx.x * 256u + threadIdx.x) >> 3, ngrp = (gridDim.x * 256u) >> 3;
    const unsigned rounds = (total + ngrp - 1) / ngrp;      // whole waves stay in the loop for the shuffles
    for (unsigned it = 0; it < rounds; it++) {
        const unsigned p = it * ngrp + grp;
        const bool live = p < total;
        int m = 0;
        LmCcRec rec, urec;
        uint2 pr = make_uint2(0u, 0u);
        long long ci = 0;
        bool possible = false, big = false;
        LmIsect is;
        if (live) {
            pr = mb.pairs[SRC][p];
            const int ui = mb.pair_u[SRC][p];
            ci = (long long)mb.nt_list[pr.x];
            rec = cc[ci];
            urec = cc[ui];
            is = lm_isect(rec, urec);
            // the common pixels cannot outnumber the smaller CC: most box neighbours of a large component are glyph-sized and
            // fail the thresholds on the sizes alone
            possible = lm_accept(rec.size < urec.size ? rec.size : urec.size, rec.size, urec.size, min_recall, min_precision);
            big = possible && is.nwc * (is.y1 - is.y0 + 1) > LM_MB_BIGPAIR;
        }
        // a large intersection goes to the list of lm_k_mb_eval_big (a whole workgroup per pair) when there is room
        unsigned slot = 0xffffffffu;
        if (big && sub == 0) slot = atomicAdd(&mb.n_big[SRC], 1u);
        slot = (unsigned)__shfl((int)slot, 0, 8);
        const bool deferred = big && slot < mb.cap_big;
        if (deferred && sub == 0) mb.big[SRC][slot] = p;
        if (live && possible && !deferred) {
            // lane `sub` takes rows sub, sub + 8, ... of the intersection (no integer divisions)
            for (int y = is.y0 + sub; y <= is.y1; y += 8) {
                const unsigned long long ra = rec.crop_off + (unsigned long long)((y - rec.min_y) * is.anw + (is.wc0 - is.awx0));
                const unsigned long long ru = urec.crop_off + (unsigned long long)((y - urec.min_y) * is.unw + (is.wc0 - is.uwx0));
                for (int j = 0; j < is.nwc; j++) m += __popc(crop[ra + j] & crop[ru + j]);
            }
        }
#pragma unroll
        for (int d = 4; d >= 1; d >>= 1) m += __shfl_xor(m, d, 8);
        if (live && possible && !deferred && sub == 0 && lm_accept(m, rec.size, urec.size, min_recall, min_precision)) {
            mb.pairs[SRC][p].x = pr.x | 0x80000000u;
            // alive at rec.frame even if the unique is never matched again: c cannot become a new unique
            if (SRC == 0 && lm_mb_alive(rec.frame, active_last[pr.y], max_gap)) mb.sidx[ci] = -1;
        }
    }
}

// the pairs lm_k_mb_eval set aside: one workgroup per pair, a thread per word of the box intersection
template <int SRC>
__global__ void __launch_bounds__(256) lm_k_mb_eval_big(const LmCcRec* __restrict__ cc, const uint32_t* __restrict__ crop,
                                                        const int32_t* __restrict__ active_last, LmCounters* __restrict__ cnt, LmMatchBatch mb,
                                                        double min_recall, double min_precision, int max_gap)
{
    __shared__ int s_sum;
    if (cnt->error) return;
    const unsigned n = mb.n_big[SRC] < mb.cap_big ? mb.n_big[SRC] : mb.cap_big;
    for (unsigned b = blockIdx.x; b < n; b += gridDim.x) {
        const unsigned p = mb.big[SRC][b];
        const uint2 pr = mb.pairs[SRC][p];
        const long long ci = (long long)mb.nt_list[pr.x];
        const LmCcRec rec = cc[ci], urec = cc[mb.pair_u[SRC][p]];
        const LmIsect is = lm_isect(rec, urec);
        if (threadIdx.x == 0) s_sum = 0;
        __syncthreads();
        int m = lm_overlap_words(rec, urec, is, crop, (int)threadIdx.x, 256);
        m = lm_wave_sum(m);
        if (lm_lane() == 0 && m) atomicAdd(&s_sum, m);
        __syncthreads();
        if (threadIdx.x == 0 && lm_accept(s_sum, rec.size, urec.size, min_recall, min_precision)) {
            mb.pairs[SRC][p].x = pr.x | 0x80000000u;
            if (SRC == 0 && lm_mb_alive(rec.frame, active_last[pr.y], max_gap)) mb.sidx[ci] = -1;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// T: exact twins.  The hash of a CC's crop comes with its record (lm_k_emit / lm_k_crop_hash leave it in chash[]); candidates
// are verified word by word with the work spread over the crop WORDS of the batch (a lecture's large components have crops of
// 10^4..10^5 words: a few lanes per CC would leave the comparison to a handful of waves).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) lm_k_mb_twin_insert(const LmCcRec* __restrict__ cc, const uint32_t* __restrict__ chash,
                                                           const long long* __restrict__ frame_cc_off, int f0, int B,
                                                           LmCounters* __restrict__ cnt, LmMatchBatch mb)
{
    if (cnt->error) return;
    const long long C0 = frame_cc_off[f0], C1 = frame_cc_off[f0 + B];
    const long long n = C1 - C0;
    if (blockIdx.x == 0 && threadIdx.x < LM_MB_MAX_FRAMES) mb.nt_cnt[threadIdx.x] = 0;      // counted by lm_k_mb_twin_final
    if (n * 2 > LM_MB_TTAB) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const LmCcRec r = cc[C0 + i];
        const unsigned h = chash[C0 + i];
        const unsigned long long box = lm_pack_box(r);
        unsigned key = lm_mix32(h ^ lm_mix32((unsigned)box) ^ lm_mix32((unsigned)(box >> 32) + 0x85ebca6bu) ^ lm_mix32((unsigned)r.size + 0xc2b2ae35u));
        if (key == 0xffffffffu) key = 0;
        mb.tkey[C0 + i] = key;
        const unsigned long long val = ((unsigned long long)key << 32) | (unsigned long long)(unsigned)i;
        for (unsigned probe = 0; probe < LM_MB_TTAB; probe++) {
            const unsigned slot = (key + probe) & (LM_MB_TTAB - 1);
            const unsigned long long old = atomicCAS(&mb.ttab[slot], ~0ull, val);
            if (old == ~0ull) break;
            if ((unsigned)(old >> 32) == key) { atomicMin(&mb.ttab[slot], val); break; }
        }
    }
}

// candidate root of every CC: the smallest CC of the batch with the same key, if it has the same box and size and lies at most
// max_gap frames earlier.  twin[] = 1 for the candidates; lm_k_mb_twin_cmp clears it where the crops differ.
__global__ void __launch_bounds__(256) lm_k_mb_twin_probe(const LmCcRec* __restrict__ cc, const long long* __restrict__ frame_cc_off, int f0, int B,
                                                          LmCounters* __restrict__ cnt, LmMatchBatch mb, int max_gap)
{
    if (cnt->error) return;
    const long long C0 = frame_cc_off[f0], C1 = frame_cc_off[f0 + B];
    const long long n = C1 - C0;
    const bool enabled = n * 2 <= LM_MB_TTAB;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        long long root = -1;
        if (enabled) {
            const unsigned key = mb.tkey[C0 + i];
            for (unsigned probe = 0; probe < LM_MB_TTAB; probe++) {
                const unsigned slot = (key + probe) & (LM_MB_TTAB - 1);
                const unsigned long long e = mb.ttab[slot];
                if (e == ~0ull) break;
                if ((unsigned)(e >> 32) == key) {
                    root = (long long)(unsigned)e;
                    mb.tkey[C0 + i] = slot;         // the entry this CC's key lives in: emptied again by lm_k_mb_twin_final
                    break;
                }
            }
            if (root >= 0 && root < i) {
                const LmCcRec r = cc[C0 + i], q = cc[C0 + root];
                if (!(r.size == q.size && r.min_x == q.min_x && r.max_x == q.max_x && r.min_y == q.min_y && r.max_y == q.max_y &&
                      q.frame < r.frame && r.frame - q.frame <= max_gap))
                    root = -1;
            } else {
                root = -1;
            }
        }
        mb.twin[C0 + i] = root >= 0 ? 1 : 0;
        mb.troot[C0 + i] = root >= 0 ? (int32_t)(C0 + root) : -1;
    }
}

// index (relative to C0) of the record whose crop holds word w; the records' crop offsets ascend with the record index.
// `start` = a record at or before the one wanted.
LM_DEV long long lm_cc_of_word(const LmCcRec* __restrict__ cc, long long C0, long long n, unsigned long long w, long long start)
{
    long long k = start;
    while (k + 1 < n && cc[C0 + k + 1].crop_off <= w) k++;
    return k;
}

LM_DEV long long lm_cc_of_word_search(const LmCcRec* __restrict__ cc, long long C0, long long n, unsigned long long w)
{
    long long lo = 0, hi = n;               // largest k with crop_off <= w
    while (hi - lo > 1) {
        const long long mid = (lo + hi) >> 1;
        if (cc[C0 + mid].crop_off <= w) lo = mid; else hi = mid;
    }
    return lo;
}

// one lane per crop word of the batch: a candidate whose word differs from its root's is not a twin.
// Round 3: a workgroup takes a contiguous range of the batch's crop words, finds the record of its first word with a 256-way
// search over the ascending crop offsets (three rounds at most) and stages offsets, candidate roots and the roots' crop offsets
// of the next LM_TW_WIN records in LDS; before, every wave ran a 17-step binary search over the records in L2 and three more
// dependent loads per word (184 vector loads per wave, 79 us per 64 dense frames).
#define LM_TW_WIN 256
#define LM_TW_UNR 4

__global__ void __launch_bounds__(256) lm_k_mb_twin_cmp(const LmCcRec* __restrict__ cc, const uint32_t* __restrict__ crop,
                                                        const long long* __restrict__ frame_cc_off, int f0, int B,
                                                        LmCounters* __restrict__ cnt, LmMatchBatch mb)
{
    __shared__ unsigned long long s_off[LM_TW_WIN + 1], s_roff[LM_TW_WIN];
    __shared__ int s_root[LM_TW_WIN];
    __shared__ int s_cnt4[4], s_any;
    if (cnt->error) return;
    const long long C0 = frame_cc_off[f0], C1 = frame_cc_off[f0 + B];
    const long long n = C1 - C0;
    if (n <= 0 || n * 2 > LM_MB_TTAB) return;
    const unsigned long long W0 = cc[C0].crop_off;
    // End of the batch's crop words from the batch's OWN last record (lm_k_select's layout: box rows x 32-px column words).  The
    // stream counters and cc[C1] belong to the NEXT batch, whose emission may already be running on the other queue.
    const LmCcRec last = cc[C1 - 1];
    const unsigned long long W1 = last.crop_off + (unsigned long long)((last.max_x >> 5) - (last.min_x >> 5) + 1) *
                                                      (unsigned long long)(last.max_y - last.min_y + 1);
    constexpr unsigned long long STEP = 256ull * LM_TW_UNR;
    const unsigned long long per = (((W1 - W0 + gridDim.x - 1) / gridDim.x + STEP - 1) / STEP) * STEP;
    const unsigned long long wb0 = W0 + (unsigned long long)blockIdx.x * per;
    if (wb0 >= W1) return;              // uniform in the workgroup
    const unsigned long long wb1 = (W1 - wb0 < per) ? W1 : wb0 + per;
    const int tid = (int)threadIdx.x, lane = lm_lane();
    // largest k with crop_off <= wb0
    long long klo = 0;
    for (long long len = n; len > 1;) {
        const long long stride = (len + 255) >> 8;
        const bool in = (long long)tid * stride < len;
        const unsigned long long bal = __ballot(in && cc[C0 + klo + tid * stride].crop_off <= wb0);
        __syncthreads();
        if (lane == 0) s_cnt4[tid >> 6] = (int)__popcll(bal);
        __syncthreads();
        const int c = s_cnt4[0] + s_cnt4[1] + s_cnt4[2] + s_cnt4[3];       // >= 1
        klo += (c - 1) * stride;
        len = (len - (c - 1) * stride < stride) ? len - (c - 1) * stride : stride;
    }
    for (unsigned long long cur = wb0; cur < wb1;) {
        const int nwin = (n - klo < LM_TW_WIN) ? (int)(n - klo) : LM_TW_WIN;
        __syncthreads();        // the previous window has been read
        if (tid == 0) s_any = 0;
        __syncthreads();
        if (tid < nwin) {
            const int32_t root = mb.troot[C0 + klo + tid];
            s_off[tid] = cc[C0 + klo + tid].crop_off;
            s_root[tid] = root;
            s_roff[tid] = root >= 0 ? cc[root].crop_off : 0ull;
            if (root >= 0) s_any = 1;
        }
        if (tid == 0) s_off[nwin] = (klo + nwin < n) ? cc[C0 + klo + nwin].crop_off : W1;
        __syncthreads();
        const unsigned long long wend = s_off[nwin] < wb1 ? s_off[nwin] : wb1;
        if (s_any) {
            for (unsigned long long w0 = cur + (unsigned long long)(tid & ~63) * LM_TW_UNR; w0 < wend; w0 += STEP) {
                int jj[LM_TW_UNR];
                unsigned mine[LM_TW_UNR], theirs[LM_TW_UNR];
#pragma unroll
                for (int u = 0; u < LM_TW_UNR; u++) {
                    const unsigned long long w = w0 + (unsigned long long)(u * 64 + lane);
                    jj[u] = -1; mine[u] = 0; theirs[u] = 0;
                    if (w < wend) {
                        int lo = 0, hi = nwin;          // largest j with s_off[j] <= w
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (s_off[mid] <= w) lo = mid; else hi = mid;
                        }
                        if (s_root[lo] >= 0) {
                            jj[u] = lo;
                            mine[u] = crop[w];
                            theirs[u] = crop[s_roff[lo] + (w - s_off[lo])];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < LM_TW_UNR; u++)
                    if (jj[u] >= 0 && mine[u] != theirs[u]) mb.twin[C0 + klo + jj[u]] = 0;
            }
        }
        cur = wend;
        klo += nwin;
    }
}

__global__ void __launch_bounds__(256) lm_k_mb_twin_final(const LmCcRec* __restrict__ cc, const long long* __restrict__ frame_cc_off, int f0, int B,
                                                          LmCounters* __restrict__ cnt, LmMatchBatch mb)
{
    if (cnt->error) return;
    const long long C0 = frame_cc_off[f0], C1 = frame_cc_off[f0 + B];
    const int lane = lm_lane();
    // The twin table is left empty for the next batch: every CC clears the entry of its key (lm_k_mb_twin_probe left the slot in
    // tkey[]; every inserted key was found there).  A 2 MB memset per batch cost 41 us -- 6.5 ms per 10,000-frame stream.
    if ((C1 - C0) * 2 <= LM_MB_TTAB)
        for (long long i = C0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < C1; i += (long long)gridDim.x * blockDim.x)
            mb.ttab[mb.tkey[i] & (LM_MB_TTAB - 1)] = ~0ull;
    // The counters of a batch share two cache lines: one atomic per CC serialised in one L2 channel (106 us per 64 dense
    // frames).  Records are in frame order, so the 64 CCs of a wave belong to one or two frames: one atomic per (wave, frame).
    for (long long base = C0 + ((long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63u)); base < C1; base += (long long)gridDim.x * blockDim.x) {
        const long long i = base + lane;
        const bool nt = i < C1 && !mb.twin[i];
        int fb = -1;
        if (nt) { mb.troot[i] = -1; fb = cc[i].frame - f0; }
        unsigned long long todo = __ballot(nt);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int fl = __shfl(fb, leader);
            const unsigned long long same = __ballot(nt && fb == fl);
            if (lane == leader) atomicAdd(&mb.nt_cnt[fl], (int)__popcll(same));
            todo &= ~same;
        }
    }
}

// crop hashes of records that did not come through lm_k_emit (lm_stream_import / lm_stream_append_packed): chash[c0 .. c0 + n)
// must be zero on entry
__global__ void __launch_bounds__(256) lm_k_crop_hash(const LmCcRec* __restrict__ cc, const uint32_t* __restrict__ crop, long long c0, long long n,
                                                      unsigned long long W1, uint32_t* __restrict__ chash)
{
    if (n <= 0) return;
    const unsigned long long W0 = cc[c0].crop_off;
    const int lane = lm_lane();
    const unsigned long long wave = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    for (unsigned long long w0 = W0 + wave * 64ull; w0 < W1; w0 += nwaves * 64ull) {
        const long long k0 = lm_cc_of_word_search(cc, c0, n, w0);
        const unsigned long long w = w0 + (unsigned long long)lane;
        const bool live = w < W1;
        long long k = k0;
        unsigned contrib = 0;
        if (live) {
            k = lm_cc_of_word(cc, c0, n, w, k0);
            contrib = lm_mix32(crop[w] + 0x9e3779b9u * (unsigned)(w - cc[c0 + k].crop_off));
        }
        const bool same = live && k == k0;
        const unsigned wsum = lm_wave_sum(same ? contrib : 0u);
        if (lane == 0 && live) atomicAdd(&chash[c0 + k0], wsum);
        if (live && !same) atomicAdd(&chash[c0 + k], contrib);
    }
}

// ------------------------------------------------------------------------------------------------
// S: compact the non-twin CCs that are not surely matched.  One block over nt_list; LM_MB_SRC_R entries per thread and pass,
// all loads of a pass in flight together, positions from per-wave ballots.
// ------------------------------------------------------------------------------------------------
#define LM_MB_SRC_R 8

__global__ void __launch_bounds__(1024) lm_k_mb_sources(const LmCcRec* __restrict__ cc, const long long* __restrict__ frame_cc_off,
                                                        int f0, int B, LmCounters* __restrict__ cnt, LmMatchBatch mb)
{
    __shared__ unsigned s_tab[LM_MB_SRC_R * 16];
    __shared__ unsigned s_tot;
    if (cnt->error) return;
    const int nNT = mb.nt_foff[B];
    const int lane = lm_lane(), wid = (int)(threadIdx.x >> 6);
    unsigned carry = 0;
    for (int base = 0; base < nNT; base += 1024 * LM_MB_SRC_R) {
        int ci[LM_MB_SRC_R], v[LM_MB_SRC_R];
#pragma unroll
        for (int k = 0; k < LM_MB_SRC_R; k++) {
            const int e = base + k * 1024 + (int)threadIdx.x;
            ci[k] = (e < nNT) ? mb.nt_list[e] : -1;
        }
#pragma unroll
        for (int k = 0; k < LM_MB_SRC_R; k++) v[k] = (ci[k] >= 0) ? mb.sidx[ci[k]] : 0;
        unsigned fm = 0;
#pragma unroll
        for (int k = 0; k < LM_MB_SRC_R; k++) {
            const int flag = v[k] == -2;
            fm |= (unsigned)flag << k;
            const unsigned long long bal = __ballot(flag);
            if (lane == 0) s_tab[k * 16 + wid] = (unsigned)__popcll(bal);
        }
        __syncthreads();
        if (wid == 0) {     // exclusive scan of the (row, wave) counts: LM_MB_SRC_R * 16 / 64 per lane
            constexpr int PER = LM_MB_SRC_R * 16 / 64;
            unsigned loc[PER], sum = 0;
#pragma unroll
            for (int j = 0; j < PER; j++) { loc[j] = s_tab[lane * PER + j]; sum += loc[j]; }
            const unsigned incl = lm_wave_incl_scan(sum);
            unsigned run = incl - sum;
#pragma unroll
            for (int j = 0; j < PER; j++) { s_tab[lane * PER + j] = run; run += loc[j]; }
            if (lane == 63) s_tot = incl;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LM_MB_SRC_R; k++) {
            const int flag = (int)((fm >> k) & 1u);
            const unsigned long long bal = __ballot(flag);
            const int e = base + k * 1024 + (int)threadIdx.x;
            if (e < nNT) {
                const unsigned o = carry + s_tab[k * 16 + wid] + (unsigned)__popcll(bal & lm_lowmask_excl(lane));
                mb.rootpos[e] = (int32_t)o;         // sources before non-twin e
                mb.nt_src[e] = flag ? (int32_t)o : -1;
                if (flag) {
                    mb.sidx[ci[k]] = (int32_t)o;
                    mb.s_list[o] = ci[k];
                    mb.s_box[o] = lm_pack_box(cc[ci[k]]);
                    mb.newpos[o] = -1;
                }
            }
        }
        carry += s_tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *mb.n_src = (int)carry;
    // sources that belong to frames before frame b: the count in front of the frame's first non-twin
    if ((int)threadIdx.x <= B) {
        const int e = mb.nt_foff[threadIdx.x];
        mb.s_prefix[threadIdx.x] = (e < nNT) ? mb.rootpos[e] : (int)carry;
    }
}

// ------------------------------------------------------------------------------------------------
// C: replay the frames in order.  One block.  Everything the replay re-reads lives in LDS: the last-matched frame per
// active position, the active position of every source, the per-frame pair ranges; a frame's pair lists (and the source
// indices of its CCs) are fetched into registers while the previous frame is being decided.  Global state is written
// through as the replay goes (positions / sources beyond the LDS tables fall back to it).
// ------------------------------------------------------------------------------------------------
#ifndef LM_MB_RT
#define LM_MB_RT 1024           // threads of the replay workgroup (512 was measured slower: 144 vs 113 us per batch)
#endif
#define LM_MB_ITEMS (LM_MB_CH / LM_MB_RT)
#define LM_MB_SCAN_DIRECT 512   // chunks of at most this many CCs rank their new uniques without the scan barrier

// what the replay needs of frame b, fetched one frame ahead: its pair lists and, per non-twin CC, the cc index, the source index
// and the last frame one of its twins appears in
LM_DEV void lm_mb_prefetch(const LmMatchBatch& mb, unsigned pA0, unsigned pA1, unsigned pB0, unsigned pB1, int n0, int n,
                           uint2 (&ra)[LM_MB_PFA], uint2 (&rb)[LM_MB_PFB], int (&rs)[LM_MB_ITEMS], int (&rc)[LM_MB_ITEMS], int (&rt)[LM_MB_ITEMS])
{
#pragma unroll
    for (int k = 0; k < LM_MB_PFA; k++) {
        const unsigned pa = pA0 + (unsigned)k * (unsigned)LM_MB_RT + threadIdx.x;
        ra[k] = (pa < pA1) ? mb.pairs[0][pa] : make_uint2(0u, 0xffffffffu);
    }
#pragma unroll
    for (int k = 0; k < LM_MB_PFB; k++) {
        const unsigned pb = pB0 + (unsigned)k * (unsigned)LM_MB_RT + threadIdx.x;
        rb[k] = (pb < pB1) ? mb.pairs[1][pb] : make_uint2(0u, 0xffffffffu);
    }
#pragma unroll
    for (int k = 0; k < LM_MB_ITEMS; k++) {
        const int i = (int)threadIdx.x * LM_MB_ITEMS + k;
        const bool on = i < n && n <= LM_MB_CH;
        rs[k] = on ? mb.nt_src[n0 + i] : -1;
        rc[k] = on ? mb.nt_list[n0 + i] : 0;
        rt[k] = on ? mb.tlast[n0 + i] : -1;
    }
}

// Only accepted pairs matter to the replay (tempo_count is computed by lm_k_mb_tempo).  pr.x = accepted << 31 | index of the
// CC in nt_list; idx >= nch: a CC of another LDS pass of this frame (frames with more than LM_MB_CH non-twins scan their pair
// lists once per pass).
LM_DEV void lm_mb_pair_old(uint2 pr, int f, int max_gap, unsigned rel0, unsigned nch, unsigned* s_best, const int* s_last,
                           const int32_t* __restrict__ active_last)
{
    if (!(pr.x >> 31)) return;
    const unsigned idx = (pr.x & 0x7fffffffu) - rel0;
    if (idx >= nch) return;
    const int last = pr.y < LM_MB_LA ? s_last[pr.y] : active_last[pr.y];
    if (lm_mb_alive(f, last, max_gap)) atomicMin(&s_best[idx], pr.y);
}

LM_DEV void lm_mb_pair_new(uint2 pr, int f, int max_gap, unsigned rel0, unsigned nch, unsigned* s_best, const int* s_last, const int* s_newpos,
                           const int32_t* __restrict__ active_last, const int32_t* __restrict__ newpos)
{
    if (!(pr.x >> 31)) return;
    const unsigned idx = (pr.x & 0x7fffffffu) - rel0;
    if (idx >= nch) return;
    const int pos = pr.y < LM_MB_LS ? s_newpos[pr.y] : newpos[pr.y];
    if (pos < 0) return;        // that source never became a unique
    const int last = pos < LM_MB_LA ? s_last[pos] : active_last[pos];
    if (lm_mb_alive(f, last, max_gap)) atomicMin(&s_best[idx], (unsigned)pos);
}

// C: replay.  Only the non-twin CCs take part: a twin is assigned its root's unique after the replay, and what it does to the
// stream state -- the unique is matched again at the twin's frame -- is folded into the root's decision: the unique's last
// match becomes max(f, last frame with a twin of the root).  That is equivalent for every aliveness test: all twins lie within
// max_gap frames of the root, so the unique is alive at every frame up to the last twin's whatever else happens, and from
// there on the value is the true one.
__global__ void __launch_bounds__(LM_MB_RT) lm_k_mb_resolve(const LmCcRec* __restrict__ cc, const long long* __restrict__ frame_cc_off,
                                                            int f0, int B, int32_t* __restrict__ active, int32_t* __restrict__ active_cc,
                                                            unsigned long long* __restrict__ active_box, int32_t* __restrict__ active_last,
                                                            LmCounters* __restrict__ cnt, int32_t* __restrict__ assign, LmMatchBatch mb,
                                                            int max_gap, int cap_uniq)
{
    LM_DYN_SMEM(smem);
    unsigned* s_best2 = (unsigned*)smem;                // [2][LM_MB_CH], double-buffered: the next step's buffer is reset while this one is decided
    int* s_last = (int*)(smem + (size_t)LM_MB_CH * 8);  // [LM_MB_LA]
    int* s_newpos = s_last + LM_MB_LA;                  // [LM_MB_LS]
    __shared__ int s_n0[LM_MB_MAX_FRAMES + 1];
    __shared__ unsigned s_tA[LM_MB_MAX_FRAMES + 1], s_tB[LM_MB_MAX_FRAMES + 1];
    __shared__ int s_fail;
    __shared__ unsigned s_wsum[LM_MB_RT / 64];
    if (cnt->error) return;
    const int nt = mb.ftile[B];
    const int nA0 = cnt->n_active;
    const int nS = *mb.n_src;
    int nA = nA0;
    int nU = cnt->n_uniq;
    // every position / source the replay can touch is covered by the LDS tables: the loop then never re-reads global memory it
    // wrote, and its barriers only need to order LDS (prefetch loads and write-through stores stay in flight)
    const bool lds_only = (long long)nA0 + nS <= LM_MB_LA && nS <= LM_MB_LS;
    if (threadIdx.x == 0) s_fail = 0;
    if ((int)threadIdx.x <= B) {
        s_n0[threadIdx.x] = mb.nt_foff[threadIdx.x];
        const int t = mb.ftile[threadIdx.x];
        s_tA[threadIdx.x] = nt > 0 ? mb.toff[0][t] : 0u;
        s_tB[threadIdx.x] = nt > 0 ? mb.toff[1][t] : 0u;
    }
    for (int i = threadIdx.x; i < nA0 && i < LM_MB_LA; i += LM_MB_RT) s_last[i] = active_last[i];
    for (int k = threadIdx.x; k < nS && k < LM_MB_LS; k += LM_MB_RT) s_newpos[k] = -1;
    for (int i = threadIdx.x; i < 2 * LM_MB_CH; i += LM_MB_RT) s_best2[i] = 0xffffffffu;
    int bb = 0;
    __syncthreads();
    uint2 ra[LM_MB_PFA], rb[LM_MB_PFB];
    int rs[LM_MB_ITEMS], rc[LM_MB_ITEMS], rt[LM_MB_ITEMS];
    lm_mb_prefetch(mb, s_tA[0], s_tA[1], s_tB[0], s_tB[1], s_n0[0], s_n0[1] - s_n0[0], ra, rb, rs, rc, rt);

    for (int b = 0; b < B; b++) {
        const int f = f0 + b;
        const int n0 = s_n0[b], n = s_n0[b + 1] - n0;       // the frame's non-twin CCs
        if (n == 0) {       // nothing to decide in this frame (uniform): no barriers
            if (b + 1 < B) lm_mb_prefetch(mb, s_tA[b + 1], s_tA[b + 2], s_tB[b + 1], s_tB[b + 2], s_n0[b + 1], s_n0[b + 2] - s_n0[b + 1], ra, rb, rs, rc, rt);
            continue;
        }
        const bool single = n <= LM_MB_CH;
        const unsigned pA0 = s_tA[b], pA1 = s_tA[b + 1], pB0 = s_tB[b], pB1 = s_tB[b + 1];     // the frame's pair lists
        for (int cb = 0; cb < n; cb += LM_MB_CH) {
            const int nch = (n - cb < LM_MB_CH) ? n - cb : LM_MB_CH;
            const unsigned rel0 = (unsigned)(n0 + cb);
            unsigned* s_best = s_best2 + bb * LM_MB_CH;
            // ---- accepted pairs against uniques that existed before the batch / were born inside it
            if (single) {
#pragma unroll
                for (int k = 0; k < LM_MB_PFA; k++)
                    if (ra[k].y != 0xffffffffu) lm_mb_pair_old(ra[k], f, max_gap, rel0, (unsigned)nch, s_best, s_last, active_last);
#pragma unroll
                for (int k = 0; k < LM_MB_PFB; k++)
                    if (rb[k].y != 0xffffffffu) lm_mb_pair_new(rb[k], f, max_gap, rel0, (unsigned)nch, s_best, s_last, s_newpos, active_last, mb.newpos);
            }
            for (unsigned p = pA0 + (single ? (unsigned)LM_MB_PFA * (unsigned)LM_MB_RT : 0u) + threadIdx.x; p < pA1; p += LM_MB_RT)
                lm_mb_pair_old(mb.pairs[0][p], f, max_gap, rel0, (unsigned)nch, s_best, s_last, active_last);
            for (unsigned p = pB0 + (single ? (unsigned)LM_MB_PFB * (unsigned)LM_MB_RT : 0u) + threadIdx.x; p < pB1; p += LM_MB_RT)
                lm_mb_pair_new(mb.pairs[1][p], f, max_gap, rel0, (unsigned)nch, s_best, s_last, s_newpos, active_last, mb.newpos);
            int cur_s[LM_MB_ITEMS], cur_c[LM_MB_ITEMS], cur_t[LM_MB_ITEMS];
#pragma unroll
            for (int k = 0; k < LM_MB_ITEMS; k++) {
                const int i = (int)threadIdx.x * LM_MB_ITEMS + k;
                const bool on = i < nch;
                cur_s[k] = single ? rs[k] : (on ? mb.nt_src[n0 + cb + i] : -1);
                cur_c[k] = single ? rc[k] : (on ? mb.nt_list[n0 + cb + i] : 0);
                cur_t[k] = single ? rt[k] : (on ? mb.tlast[n0 + cb + i] : -1);
            }
            if (cb + LM_MB_CH >= n && b + 1 < B)        // next frame's lists: in flight while this one is decided
                lm_mb_prefetch(mb, s_tA[b + 1], s_tA[b + 2], s_tB[b + 1], s_tB[b + 2], s_n0[b + 1], s_n0[b + 2] - s_n0[b + 1], ra, rb, rs, rc, rt);
            if (lds_only) lm_lds_barrier(); else __syncthreads();
            for (int i = threadIdx.x; i < LM_MB_CH; i += LM_MB_RT) s_best2[(bb ^ 1) * LM_MB_CH + i] = 0xffffffffu;     // for the next step
            // ---- decisions; matched CCs keep the active POSITION for now (-2 - pos), translated after the replay
            unsigned isnew[LM_MB_ITEMS], mine = 0;
#pragma unroll
            for (int k = 0; k < LM_MB_ITEMS; k++) {
                const int i = (int)threadIdx.x * LM_MB_ITEMS + k;
                isnew[k] = 0;
                if (i < nch) {
                    const unsigned best = s_best[i];
                    if (best != 0xffffffffu) {
                        const int upto = cur_t[k] > f ? cur_t[k] : f;       // matched now, and again at every twin's frame
                        assign[cur_c[k]] = -2 - (int32_t)best;
                        if (best < LM_MB_LA) atomicMax(&s_last[best], upto);
                        atomicMax(&active_last[best], upto);
                    } else {
                        isnew[k] = 1;
                    }
                }
                mine += isnew[k];
            }
            // exclusive scan of the new-unique flags.  Few CCs (the usual frame): no barrier -- a CC is new iff its s_best entry
            // is still empty, so every wave counts the new ones of the waves before it (and all of them) itself.  Many: ONE
            // barrier -- wave totals to LDS, every thread sums the waves before it (s_wsum is rewritten only after the barrier
            // that ends this chunk).
            const unsigned incl = lm_wave_incl_scan(mine);
            unsigned tot = 0, before = 0;
            if (nch <= LM_MB_SCAN_DIRECT) {
                const int w0 = (int)(threadIdx.x >> 6) * 64 * LM_MB_ITEMS;      // first CC of this wave
                for (int base = 0; base < nch; base += 64) {
                    const int i = base + lm_lane();
                    const unsigned c = (unsigned)__popcll(__ballot(i < nch && s_best[i] == 0xffffffffu));
                    tot += c;
                    before += (base < w0) ? c : 0u;
                }
            } else {
                if (lm_lane() == 63) s_wsum[threadIdx.x >> 6] = incl;
                if (lds_only) lm_lds_barrier(); else __syncthreads();
#pragma unroll
                for (int w = 0; w < LM_MB_RT / 64; w++) {
                    const unsigned t = s_wsum[w];
                    before += (w < (int)(threadIdx.x >> 6)) ? t : 0u;
                    tot += t;
                }
            }
            unsigned o = before + incl - mine;
#pragma unroll
            for (int k = 0; k < LM_MB_ITEMS; k++) {
                if (!isnew[k]) continue;
                const long long idx = (long long)nU + o;
                const int src = cur_s[k];
                if (idx < cap_uniq && src >= 0) {
                    const int pos = nA + (int)o;
                    const int upto = cur_t[k] > f ? cur_t[k] : f;
                    assign[cur_c[k]] = (int32_t)idx;
                    active[pos] = (int32_t)idx;
                    active_cc[pos] = cur_c[k];
                    active_last[pos] = upto;
                    if (pos < LM_MB_LA) s_last[pos] = upto;
                    mb.newpos[src] = pos;
                    if (src < LM_MB_LS) s_newpos[src] = pos;
                } else {
                    // STATE: a surely matched CC found no match (a bug, not an input condition)
                    s_fail = idx < cap_uniq ? LM_ERR_STATE : LM_ERR_CAPACITY;
                }
                o++;
            }
            nA += (int)tot;
            nU += (int)tot;
            bb ^= 1;
            if (lds_only) lm_lds_barrier(); else __syncthreads();
        }
    }
    __syncthreads();        // everything the replay wrote is visible to the whole block
    // ---- after the replay: boxes of the new actives (assignments are finished by lm_k_mb_tempo)
    if (!s_fail) {
        for (int pos = nA0 + (int)threadIdx.x; pos < nA; pos += LM_MB_RT) active_box[pos] = lm_pack_box(cc[active_cc[pos]]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_fail) {
            cnt->error = s_fail;
        } else {
            cnt->n_uniq = nU;
            cnt->n_active = nA;
            cnt->n_matched = f0 + B;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Q: tempo_count (:85) of the batch, after the replay.  (cur CC of frame f, unique u) counts iff their boxes intersect, u
// was born before f, and alive(f, last_u now): box join of ALL CCs (tiles of 64) against the active list as the replay
// left it -- entries that died inside the batch are still there, entries that were dead before it were dropped by the
// prologue.  One 64-bit atomic per workgroup.
// ------------------------------------------------------------------------------------------------
// sources filtered per round by lm_k_mb_tempo: the survivors' boxes are all the inner loop needs (the birth / last-match tests are
// made when a source is listed), so 4096 candidates per round are 32 KB of LDS and the usual active list (2-3 thousand) is ONE
// round: a header and two memory latencies per tile instead of three rounds of three (45 us alone, 108 us beside the record
// emission, per 64 dense frames).  The loads of a round are issued together, LM_MB_TUN per thread.
#ifndef LM_MB_TCHUNK
#define LM_MB_TCHUNK 4096
#endif
#define LM_MB_TUN 8

__global__ void __launch_bounds__(256) lm_k_mb_tempo(const LmCcRec* __restrict__ cc, const long long* __restrict__ frame_cc_off, int f0, int B,
                                                     const unsigned long long* __restrict__ active_box, const int32_t* __restrict__ active_cc,
                                                     const int32_t* __restrict__ active_last, LmCounters* __restrict__ cnt, LmMatchBatch mb,
                                                     int max_gap, const int32_t* __restrict__ active, int32_t* __restrict__ assign)
{
    __shared__ __attribute__((aligned(16))) unsigned long long s_sbox[LM_MB_TCHUNK + 4];
    __shared__ int s_ub[4];
    __shared__ int s_nsurv;
    __shared__ unsigned long long s_sum;
    if (cnt->error) return;
    // C' (round 2: lm_k_mb_finish, a launch of its own): the replay left active POSITIONS (-2 - pos) in the matched non-twins'
    // assignments and nothing in the twins': positions become unique indices and every twin takes its root's unique.  A twin's thread
    // may read its root's entry before or after the root's own thread translated it; both forms are understood.  Like the count
    // below it reads the active list as the replay left it.
    {
        const long long C0 = frame_cc_off[f0], C1 = frame_cc_off[f0 + B];
        for (long long i = C0 + (long long)blockIdx.x * 256 + threadIdx.x; i < C1; i += (long long)gridDim.x * 256) {
            int32_t v = assign[mb.twin[i] ? (long long)mb.troot[i] : i];
            if (v <= -2) v = active[-2 - v];
            assign[i] = v;
        }
    }
    const int nt = mb.ftile_all[B];
    const int nA = cnt->n_active;
    const int ccl = (int)(threadIdx.x & 63), q = (int)(threadIdx.x >> 6);
    unsigned long long mine = 0;
    for (int t = blockIdx.x; t < nt; t += gridDim.x) {
        const int b = lm_mb_tile_frame(mb.ftile_all, B, t);
        const int f = f0 + b;
        const long long cf1 = frame_cc_off[f + 1];
        const long long c_first = frame_cc_off[f] + (long long)(t - mb.ftile_all[b]) * LM_MB_TILE;
        const int ncc = (cf1 - c_first < LM_MB_TILE) ? (int)(cf1 - c_first) : LM_MB_TILE;
        __syncthreads();
        if (threadIdx.x == 0) { s_ub[0] = 0x7fff; s_ub[1] = -1; s_ub[2] = 0x7fff; s_ub[3] = -1; }
        __syncthreads();
        unsigned long long mybox = 0;
        const bool have = ccl < ncc;
        if (have) {
            const LmCcRec r = cc[c_first + ccl];
            mybox = lm_box_as_cc(lm_pack_box(r));
            if (q == 0) {
                atomicMin(&s_ub[0], (int)r.min_x); atomicMax(&s_ub[1], (int)r.max_x);
                atomicMin(&s_ub[2], (int)r.min_y); atomicMax(&s_ub[3], (int)r.max_y);
            }
        }
        __syncthreads();
        const unsigned long long ubox = (unsigned long long)(unsigned short)s_ub[0] | ((unsigned long long)(unsigned short)s_ub[1] << 16) |
                                        ((unsigned long long)(unsigned short)s_ub[2] << 32) | ((unsigned long long)(unsigned short)s_ub[3] << 48);
#if defined(LM_TEMPO_CUT) && LM_TEMPO_CUT == 1
        mine += ubox & 1ull; continue;
#endif
        for (int base = 0; base < nA; base += LM_MB_TCHUNK) {
            __syncthreads();
            if (threadIdx.x == 0) s_nsurv = 0;
            __syncthreads();
            for (int k0 = 0; k0 < LM_MB_TCHUNK / 256; k0 += LM_MB_TUN) {
                if (base + k0 * 256 >= nA) break;           // uniform
                unsigned long long sb[LM_MB_TUN];
                int last[LM_MB_TUN], ucc[LM_MB_TUN], born[LM_MB_TUN];
                bool cand[LM_MB_TUN];
#pragma unroll
                for (int u = 0; u < LM_MB_TUN; u++) {
                    const int i = base + (k0 + u) * 256 + (int)threadIdx.x;
                    const bool in = i < nA;
                    sb[u] = in ? active_box[i] : 0ull;
                    last[u] = in ? active_last[i] : 0;
                    ucc[u] = in ? active_cc[i] : 0;
                    cand[u] = in;
                }
#pragma unroll
                for (int u = 0; u < LM_MB_TUN; u++) {
                    cand[u] = cand[u] && lm_box_hit_packed(ubox, sb[u]) && lm_mb_alive(f, last[u], max_gap);
                    born[u] = cand[u] ? cc[ucc[u]].frame : 0x7fffffff;
                }
#pragma unroll
                for (int u = 0; u < LM_MB_TUN; u++)
                    if (cand[u] && born[u] < f) s_sbox[atomicAdd(&s_nsurv, 1)] = lm_box_as_src(sb[u]);        // born / alive: the same for every CC of the tile (one frame)
            }
            __syncthreads();
            const int ns = s_nsurv;
#if defined(LM_TEMPO_CUT) && (LM_TEMPO_CUT == 2 || LM_TEMPO_CUT == 3)
            mine += (unsigned long long)ns; continue;
#endif
            // the list is padded to a multiple of four; a wave takes four consecutive entries per trip (two 16-byte LDS reads in
            // flight together -- one entry per trip left every test behind an LDS round trip)
            if (threadIdx.x < 4) s_sbox[ns + (int)threadIdx.x] = LM_BOX_SRC_NEVER;
            __syncthreads();
            unsigned hits = 0;
            if (have)
                for (int j = q * 4; j < ns; j += 16) {
                    const lm_u64x2 e01 = *(const lm_u64x2*)&s_sbox[j], e23 = *(const lm_u64x2*)&s_sbox[j + 2];
                    hits += (lm_box_hit_pk(mybox, e01.x) ? 1u : 0u) + (lm_box_hit_pk(mybox, e01.y) ? 1u : 0u) +
                            (lm_box_hit_pk(mybox, e23.x) ? 1u : 0u) + (lm_box_hit_pk(mybox, e23.y) ? 1u : 0u);
                }
            mine += hits;
        }
    }
#if defined(LM_TEMPO_CUT) && LM_TEMPO_CUT == 2
    if (mine == 0x123456789ull) cnt->tempo_count = 1;
    return;
#endif
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    if (mine) atomicAdd(&s_sum, mine);
    __syncthreads();
#if defined(LM_TEMPO_CUT) && LM_TEMPO_CUT == 4
    if (threadIdx.x == 0 && s_sum == 0x123456789ull) cnt->tempo_count = 1;
    return;
#endif
    if (threadIdx.x == 0 && s_sum) atomicAdd(&cnt->tempo_count, s_sum);
}
