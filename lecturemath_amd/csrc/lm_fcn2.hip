// lm_fcn2.hip -- FCN-LectureNet inference, second engine: planar f16 activations + one gather-GEMM kernel on v_mfma_f32_16x16x32_f16.
//
// Same network and reference lines as lm_fcn.hip (FCN_lecturenet.py:260-323 encode_decode, :364-403 forward, :607-618 prepare_image);
// this engine runs the shipped topology (3x3 encoder / decoder, 7x7 pixel branch, every width a multiple of 16) with a per-layer
// OPERAND FORMAT: 1 = f16 x f16 (the layers below full resolution); 3 = f16 hi + lo split of both operands (hi.hi + lo.hi + hi.lo);
// 4 ("w2") = weights hi + lo x activations hi -- the input tensor then needs no lo planes; 2 ("a2") = activations hi + lo x weights hi.
// lecturemath_amd/fcn.py holds the shipped assignment and profiles/r04_fcn_formats.* the measurement behind it (logit error AND binary flips).
//
// Activations in HBM ("planar octets"): a tensor of C channels is C/8 planes of [Hp][Wp] slots of 16 bytes = 8 consecutive channels
// of one pixel as f16 (the hi parts), followed -- when a consumer runs a format that splits the activations -- by C/8 planes of the lo
// parts (lo = f16(x - hi), so hi + lo carries ~22 bits).  Hp x Wp = the image rounded up to whole tiles (16 rows, 32 columns) plus a zero
// halo as wide as the largest padding of any consumer: a convolution's input patch is then a plain rectangle of every plane, no bounds
// checks.  A producer converts ONCE per value in its epilogue (lm_fcn.hip converted fp32 -> f16 hi / lo in every consuming workgroup:
// 12-24 times per value in the deep layers) and every load of the engine is a 16-byte LDS-DMA (buffer_load_dwordx4 ... lds).
//
// The kernel (lm_k_g2) is a gather-GEMM  D[channel][pixel] = sum_k A[channel][k] * B[k][pixel]  on 16 x 16 x 32 MFMA tiles:
//   * a workgroup = a tile of 16 rows x 16 * NC pixels x MT tiles of 16 output channels, computed by 4 waves: wave w owns the pixel
//     rows 4w..4w+3 (4 * NC column fragments of 16 consecutive pixels) and all MT channel tiles: 4 * NC * MT accumulators of 4 registers.
//     NC = 2 (16 x 32 tiles) halves the weight bytes fetched and the A-fragment reads per pixel; LOADER = 1 adds a fifth wave that
//     issues every LDS-DMA of the workgroup (the compute waves then issue no vector-memory instruction in their loop);
//   * K is walked in SLICES of 32: the four 8-wide k-groups of a slice (lane >> 4) are four (plane, tap) pairs chosen by the HOST --
//     four channel octets of one tap, or two octets of two taps, or taps of a "pair plane" (below) -- so a layer's K needs no padding
//     beyond its last slice.  The B fragment of k-group g is ONE ds_read_b128 at (pair's slot of the lane's pixel); the four pairs'
//     patch offsets of every slice sit in an LDS table (16 bytes per slice, copied once per workgroup): the slice loop issues LDS
//     operations only, which return in order, so its waits are counted;
//   * weights are the A operand, packed by the host in fragment order per (channel block, weight group, slice, tile, hi | lo);
//   * staging: the chunk's planes (double-buffered when a layer has several chunks and LDS allows) and the weight groups (a ring of
//     two or three LDS buffers) are fetched by LDS-DMA while the current group's MFMAs run; one counted vmcnt + barrier per group;
//   * epilogues: bias + GELU + f16 hi (+ lo) as whole-octet stores (+ the 2x2 max-pooled copy), the four parities of a transposed
//     convolution, fp32 rows of a head's row convolution, or (EPI_V) the head's row convolution AND its vertical sums in one kernel.
// Pair planes: a 3-channel input (the RGB frame, the diff of the pixel branch) is stored as ONE plane whose slot x holds
// {c0 c1 c2 0 of pixel x | c0 c1 c2 0 of pixel x + 1}: one k-group covers two horizontal taps, so a 7-tap kernel row of the 3-channel
// part costs 4 k-groups instead of 7 x a zero-padded 16-channel chunk (lm_fcn.hip), and conv_down_1's K = 27 fits two slices.
// What round 4 measured about this kernel (stamps, timing-only cut builds, variants) is in DESIGN.md section 4.5 and profiles/r04_*.
#include "lm_common.h"

#if LM_HIP_EMULATED
lm_f32x4 hipemu_mfma_16x16x32f16(lm_h8 a, lm_h8 b, lm_f32x4 c);
#define LM_MFMA16(a, b, c) hipemu_mfma_16x16x32f16(a, b, c)
// the LDS destination is wave-uniform base + lane * 16, the source address is base + wave-uniform soff + per-lane voff
typedef const char* LmRsrc;
#define LM_MAKE_RSRC(base) ((const char*)(base))
#define LM_DMA16(rsrc, voff, soff, lds_base) memcpy((char*)(lds_base) + lm_lane() * 16, (rsrc) + (unsigned)(voff) + (unsigned)(soff), 16)
#define LM_VMWAIT0() ((void)0)
#else
#define LM_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
// LDS-DMA as `buffer_load_dwordx4 ... offen lds` (a raw buffer over the whole allocation, 32-bit offsets), NOT `global_load_lds_dwordx4`:
// the compiler's wait-count pass treats a pending global_load_lds as a FLAT access that may return out of order with LDS operations and
// then waits lgkmcnt(0) before EVERY use of an LDS read -- with weights and patches in flight all through the slice loop, no fragment
// read was ever left outstanding across an MFMA block (r04: 620 cycles per 128-cycle slice in the deep layers).  The MUBUF form counts
// on vmcnt only and the loop's LDS waits come out counted.
typedef __amdgpu_buffer_rsrc_t LmRsrc;
#define LM_MAKE_RSRC(base) __builtin_amdgcn_make_buffer_rsrc((void*)(base), 0, -1, 0x00020000)
#define LM_DMA16(rsrc, voff, soff, lds_base) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds_base), 16, (int)(voff), (int)(soff), 0, 0)
// LDS-DMA is a pending LDS write on the vector-memory counter: the issuing wave waits for its own, the barrier after it covers the
// other waves' (MI355X_MICROARCH.md, "Two waves per SIMD" item 7)
#define LM_VMWAIT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

// waits until at most n (wave-uniform) of the wave's vector-memory operations are outstanding; they retire in issue order, so
// this leaves the n youngest -- the LDS-DMA of the weight group after next -- in flight across the barrier
LM_DEV void lm_vmwait(int n)
{
#if !LM_HIP_EMULATED
#define LM_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n < 32 ? n : 32) {      // a smaller count than asked for only waits longer
        LM_VMW(0) LM_VMW(1) LM_VMW(2) LM_VMW(3) LM_VMW(4) LM_VMW(5) LM_VMW(6) LM_VMW(7) LM_VMW(8) LM_VMW(9) LM_VMW(10) LM_VMW(11) LM_VMW(12) LM_VMW(13)
        LM_VMW(14) LM_VMW(15) LM_VMW(16) LM_VMW(17) LM_VMW(18) LM_VMW(19) LM_VMW(20) LM_VMW(21) LM_VMW(22) LM_VMW(23) LM_VMW(24) LM_VMW(25) LM_VMW(26)
        LM_VMW(27) LM_VMW(28) LM_VMW(29) LM_VMW(30) LM_VMW(31)
        default: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    }
#undef LM_VMW
#endif
}

// Diagnostic build only (-DLM_G2_STAMPS, tools/variants): per-wave cycle stamps of ONE layer (LM_G2_STAMP_LAYER) into a device buffer that no
// kernel reads -- where a compute wave's lifetime goes (prologue, slice loops, group barriers, chunk switches, epilogue).
#ifndef LM_G2_STAMPS
#define LM_G2_STAMPS 0
#endif
#ifndef LM_G2_CUT           // timing-only diagnostic builds (wrong results): 1 = no weight fetches after the first groups, 2 = no activation, 3 = no stores, 4 = no epilogue, 6 = no patch fetches after the first chunk, 7 = a quarter of the pixel-fragment LDS reads
#define LM_G2_CUT 0
#endif
#if LM_G2_STAMPS && !LM_HIP_EMULATED
#define LM_G2_NSTAMP 12
#define LM_G2_STAMP_WAVES (8192 * 4 * 2)
__device__ unsigned long long lm_g2_stamp_buf[LM_G2_STAMP_WAVES * LM_G2_NSTAMP];
#define LM_STAMP_NOW() __builtin_amdgcn_s_memtime()
#define LM_STAMP_REAL() __builtin_amdgcn_s_memrealtime()
#else
#define LM_STAMP_NOW() 0ull
#define LM_STAMP_REAL() 0ull
#endif

// nn.GELU (erf form) as lm_gelu (lm_fcn.hip: erfc by Abramowitz & Stegun 7.1.26, |error| <= 3.3e-7), arranged without fmaxf / fabsf calls:
// gelu(x) = (x + |x| (1 - q)) / 2 with q = erfc(|x| / sqrt 2): for x >= 0 that is x - x q / 2, for x < 0 it is x q / 2.  (fmaxf compiles to
// two v_max_f32 per value -- one only canonicalises a NaN; with 32-128 values per lane the epilogues are VALU bound.)
LM_DEV float lm_gelu_fast(float x)
{
    const float ax = __builtin_fabsf(x), z = ax * 0.70710678118654752440f;
#if LM_HIP_EMULATED
    const float t = 1.0f / fmaf(0.3275911f, z, 1.0f);
    const float e = exp2f(-(z * z) * 1.4426950408889634f);
#else
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float e = __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);
#endif
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    return 0.5f * fmaf(ax, 1.0f - p * t * e, x);
}

#define LM_G2_EPI_PO 0      // planar-octet output (+ optional lo planes, + optional 2x2 max-pooled copy), GELU or none
#define LM_G2_EPI_T 1       // fp32 [pixel][TS] rows of the head row convolutions (lm_k_vsum2_*), no activation
#define LM_G2_EPI_TC 2      // transposed 2x2 / stride 2, one (dy, dx) parity per workgroup: output pixel (2y + dy, 2x + dx), planar octets
#define LM_G2_EPI_V 4       // head row convolution + the vertical sum of its rows in ONE kernel (kernel template EPI_V, LmG2Args::vmode: 1 = text mask +
                            // reconstruction + diff pair plane, 2 = output logit): tiles of 16 computed rows give 10 finished rows; the fp32
                            // row buffer [pixel][16] (133 + 66 MB written and read back per 1080p frame) stays in LDS
#define LM_G2_V_ROWS 10
#define LM_G2_EPI_TC2 3     // the same with BOTH dx of a 32-channel block in one workgroup (tile pair 0 = dx 0, pair 1 = dx 1): a lane writes the
                            // two neighbouring output slots (32 contiguous bytes) instead of 16 bytes at a 32-byte stride; kernel template EPI_TC

struct LmG2Args {
    const char* arena;              // base of the engine's activation arena
    const long long* psrc;          // [nchunks][npc][2] byte offsets of the chunk's planes in the arena: hi, lo
    const char* wpk;                // packed weights: [channel block][wblock_bytes]
    const int4* groups;             // [ngroups] {first slice, slices, chunk, byte offset of the group's weights inside a channel block}
    const int4* t4;                 // [slices + 2] LDS byte offsets of the four k-groups' pairs inside the patch buffer (two entries of look-ahead)
    const float* bias;              // [Cout] in the kernel's channel order (= natural order)
    long long wblock_bytes;
    int nchunks, npc, ngroups, nslices;
    int wbuf_bytes;                 // LDS bytes of one weight buffer (largest group)
    int wring;                      // weight buffers (2 or 3): wring - 1 groups are in flight or in use ahead of the one being read
    int pdouble;                    // 1: two patch buffers (next chunk fetched under the current one)
    int Wp_in;                      // slots per padded input row
    int org_in;                     // slot offset of tile (0, 0)'s patch origin in a plane
    int tiles_x, tiles_y, cblocks;  // pixel tiles, channel blocks of MT * 16 outputs
    int H, W;                       // output grid of this launch (bounds of the stores; the input grid for EPI_TC)
    int act;                        // LM_ACT_GELU or LM_ACT_NONE
    int tc_merged;                  // EPI_TC: 1 = LM_G2_EPI_TC2 (parity = dy, tile pair q = dx)
    int stamp;                      // diagnostic builds: this launch writes its stamps
    // EPI_PO / EPI_TC output tensor
    char* out_hi; char* out_lo;     // plane 0 of the hi / lo parts (lo may be null)
    long long out_plane;            // bytes per plane
    int Wp_out, halo_out;
    char* pool_hi; char* pool_lo; long long pool_plane; int Wp_pool, halo_pool;     // EPI_PO: 2x2 max-pooled copy, null when absent
    // EPI_T
    float* tout; int ts, tn;        // [pixel][ts] floats, the first tn of the 16 rows are stored
    int tile_rows;                  // image rows from one tile row to the next: 16, LM_G2_V_ROWS for EPI_V
    // EPI_V
    int vmode;
    const float* vbias;             // vmode 1: [0] text, [1..3] reconstruction; vmode 2: [0] the output logit
    const char* x0_hi; const char* x0_lo; char* dp_hi; char* dp_lo; int Wp_v, halo_v;      // the network input and the diff, pair planes
    float* v_text; float* v_rec4; float* v_out;      // v_rec4: [3][H][W] (the fused head writes the caller's image directly), may be null
};

LM_DEV void lm_pair_store(char* hi_plane, char* lo_plane, int Wp, int halo, int y, int x, float v0, float v1, float v2);

// one tile of 16 rows x 16 * NC columns x MT channel tiles; see the header comment.
//   NC      column tiles of 16 pixels per wave row group: the wave's 4 rows x NC x 16 pixels share every weight fragment (2 = a 16 x 32
//           tile: half the weight bytes fetched and half the A-fragment LDS reads per pixel)
//   LOADER  1: waves 0..3 compute (LDS reads + MFMAs, no vector-memory instruction in their loop) and wave 4 issues every LDS-DMA of the
//           workgroup and waits for them; 0: four waves, each issues its quarter of the fetches in front of its slice loop.
//           r04 stamps (profiles/r04_fcn_stamps_v2.txt): DMA issue -- back-pressured by the CU's fetch rate -- was 50 % of a wave's time in
//           the deep layers and 15-25 % at full resolution, all of it in front of the MFMAs of the same wave.  The fifth wave costs
//           registers: with two workgroups per CU one SIMD hosts three waves (<= 168 VGPRs), so the widest instances keep LOADER = 0.
template <int KH, int KW, int TERMS, int MT, int EPI, int NC, int LOADER>
__global__ void __launch_bounds__(LOADER ? 320 : 256, LOADER ? (((TERMS != 1 && MT * NC >= 4) || (TERMS == 3 && MT >= 3)) ? 2 : 3) : 2)
    lm_k_g2(const LmG2Args a, const long long* __restrict__ t_psrc, const int4* __restrict__ t_groups, const int4* __restrict__ t_t4,
            const float* __restrict__ t_bias)
{
    // The tables come as __restrict__ kernel arguments of their own: read-only and never aliased by the kernel's stores, they are read
    // with SCALAR loads.  As members of `a` they were vector loads -- and a wave waits for a vector load's result with vmcnt, which also
    // waits for every LDS-DMA issued before it: one such load per slice serialised the whole fetch pipeline.
    LM_DYN_SMEM(smem);
    [[maybe_unused]] const unsigned long long st_t0 = LM_STAMP_NOW(), st_r0 = LM_STAMP_REAL();
    [[maybe_unused]] unsigned long long st_compute = 0, st_wait = 0, st_switch = 0, st_pro = 0, st_mark = 0, st_issue = 0, st_bar = 0;
    constexpr int TW = 16 * NC;                         // tile width in pixels
    constexpr int PW = TW + KW - 1, PH = 16 + KH - 1, NSLOT = PH * PW;
    constexpr int PLS = (NSLOT * 16 + 255) & ~255;      // LDS bytes per plane: planes a multiple of the 256-B bank row apart, so the 16
                                                        // lanes of a ds_read_b128 service group (16 different pixels of a row, two
                                                        // k-groups) fall on 16 different 16-B slots
    // TERMS names the operand format: 1 = f16 x f16; 2 = weights hi x activations hi + lo; 3 = hi.hi + lo.hi + hi.lo; 4 = weights
    // hi + lo x activations hi (the input tensor then needs no lo planes at all: half the patch bytes in HBM, DMA and LDS)
    constexpr bool SPLIT_B = TERMS == 2 || TERMS == 3, SPLIT_A = TERMS >= 3;
    constexpr int NHL = SPLIT_B ? 2 : 1;                // patch planes per octet: hi (, lo)
    constexpr int NWL = SPLIT_A ? 2 : 1;                // weight fragments per tile: hi (, lo)
    constexpr int NT = 4 * NC;                          // pixel fragments per wave: index c * 4 + r = column tile c, row r
    constexpr int NIW = LOADER ? 1 : 4;                 // waves that issue DMA
    constexpr int NTHR = LOADER ? 320 : 256;
    constexpr int NJ = (NSLOT + 64 * NIW - 1) / (64 * NIW);    // DMA instructions per issuing wave and plane
    const int lane = lm_lane(), wave = LM_UNIFORM((int)(threadIdx.x >> 6)), kg = lane >> 4, col = lane & 15;
    const bool issuer = LOADER ? wave == 4 : true, computes = wave < 4;
    const int iw = LOADER ? 0 : wave;
    // Workgroup -> (pixel tile, channel block).  The launch is one-dimensional; consecutive ids go round the 8 XCDs, so id % 8 names
    // the workgroups that share an L2.  Each of them takes a contiguous share of the (channel block, tile) list: an XCD then streams
    // the weights of one or two channel blocks (they stay in its 4 MB L2) instead of every block's (5-10 MB in the deep layers).
    int wg = (int)blockIdx.x;
    {
        const int nwg = (int)gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7, j = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tiles = a.tiles_x * a.tiles_y;
    const int cblk = wg / tiles, tile = wg - cblk * tiles;          // channel block (x parity for EPI_TC), tile
    const int cby = (EPI == LM_G2_EPI_TC) ? cblk % a.cblocks : cblk, par = (EPI == LM_G2_EPI_TC) ? cblk / a.cblocks : 0;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int pbuf_bytes = a.npc * NHL * PLS;
    char* const s_t4 = smem;                                            // [nslices + 2] int4: the slices' four pair offsets
    char* const s_p0 = smem + (((a.nslices + 2) * 16 + 255) & ~255);
    char* const s_w0 = s_p0 + (a.pdouble ? 2 : 1) * pbuf_bytes;
    for (int i = (int)threadIdx.x; i < a.nslices + 2; i += NTHR) ((int4*)s_t4)[i] = t_t4[i];       // visible after the prologue's barrier
    const int wrow = wave * (4 * PW * 16);                              // the wave's first pixel row inside a plane

    // DMA source offsets of the lane's slots inside a plane (the same for every plane and chunk)
    int goff[NJ];
    bool gval[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int slot = (j * NIW + iw) * 64 + lane;
        const int row = slot / PW, x = slot - row * PW;
        goff[j] = (row * a.Wp_in + x) * 16;
        gval[j] = slot < NSLOT;
    }
    // byte offsets below 4 GB: the arena's size is checked by lm_fcn2_create, a layer's packed weights by lm_fcn2_set_layer
    const unsigned tile_org = (unsigned)(((long long)(ty * a.tile_rows) * a.Wp_in + tx * TW + a.org_in) * 16);
    const unsigned wsrc = (unsigned)(((long long)par * a.cblocks + cby) * a.wblock_bytes);
    const LmRsrc r_arena = LM_MAKE_RSRC(a.arena), r_w = LM_MAKE_RSRC(a.wpk);

    auto issue_patch = [&](int chunk, int buf) {
#if LM_G2_CUT == 6      // timing-only build: only the first chunk's planes are fetched
        if (chunk > 0) return;
#endif
        char* const dst = s_p0 + buf * pbuf_bytes;
        for (int p = 0; p < a.npc; p++)
#pragma unroll
            for (int hl = 0; hl < NHL; hl++) {
                const unsigned src = (unsigned)t_psrc[(chunk * a.npc + p) * 2 + hl] + tile_org;
#pragma unroll
                for (int j = 0; j < NJ; j++)
                    if (gval[j]) LM_DMA16(r_arena, goff[j], src, dst + (p * NHL + hl) * PLS + (j * NIW + iw) * 1024);
            }
    };
    // returns the number of DMA instructions THIS wave issued (a group is a whole number of 1-KB fragments)
    auto issue_weights = [&](const int4 g4, int buf) -> int {
#if LM_G2_CUT == 1      // timing-only build: weights fetched for the first groups only (later groups read what is there)
        if (g4.x > 8) return 0;
#endif
        const unsigned src = wsrc + (unsigned)g4.w;
        char* const dst = s_w0 + buf * a.wbuf_bytes;
        const int n = g4.y * (MT * NWL);
        int cnt = 0;
        for (int i = iw; i < n; i += NIW, cnt++) LM_DMA16(r_w, lane * 16, src + (unsigned)i * 1024, dst + i * 1024);
        return cnt;
    };

    lm_f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[m][n][r] = 0.0f;

    // Weight groups go through a ring of a.wring buffers: while group g is read, g + 1 has landed or is landing and (ring of 3) g + 2
    // is being fetched.  Per group: the issuers issue the fetches, the compute waves run the group's MFMAs, the issuers wait until only
    // their youngest fetches (the group after next) are outstanding, barrier.
    const int ring = a.wring;
    int4 grp = t_groups[0];
    int n_ahead = 0;                            // DMA instructions of this wave that may stay in flight across the next barrier
    int4 nxt = grp;
    if (a.ngroups > 1) nxt = t_groups[1];
    if (issuer) {
        issue_patch(0, 0);
        issue_weights(grp, 0);
        if (a.ngroups > 1) n_ahead = issue_weights(nxt, 1);
        if (ring < 3) n_ahead = 0;
        lm_vmwait(n_ahead);
    }
    lm_lds_barrier();
    st_pro = st_mark = LM_STAMP_NOW();
    int pb = 0, wb = 0, prev_chunk = -1;
    for (int g = 0; g < a.ngroups; g++) {
        const bool more = g + 1 < a.ngroups;
        int4 nx2 = nxt;
        if (g + 2 < a.ngroups) nx2 = t_groups[g + 2];       // scalar load, back long before the group ends
        if (issuer) {
            int wb2 = wb + 2; if (wb2 >= ring) wb2 -= ring;
            // first group of a chunk: the next chunk's planes go into the other patch buffer, read last before the barrier that ended the
            // previous chunk.  Issued BEFORE the weights: it must have landed by the end of this group, the weights need not.
            if (a.pdouble && grp.z + 1 < a.nchunks && grp.z != prev_chunk) issue_patch(grp.z + 1, pb ^ 1);
            n_ahead = 0;
            if (ring >= 3) {
                if (g + 2 < a.ngroups) n_ahead = issue_weights(nx2, wb2);
            } else if (more && g > 0) {
                // ring of 2: group g + 1 goes into the buffer group g - 1 was read from (group 1 was fetched in the prologue)
                issue_weights(nxt, wb ^ 1);
            }
        }
        prev_chunk = grp.z;
#if LM_G2_STAMPS && !LM_HIP_EMULATED
        { const unsigned long long t = LM_STAMP_NOW(); st_issue += t - st_mark; st_mark = t; }
#endif
        if (computes) {
            const char* const pbase = s_p0 + pb * pbuf_bytes + wrow;
            const char* const wbase = s_w0 + wb * a.wbuf_bytes + lane * 16;
            struct Frag { lm_h8 bh[NT], bl[NT], ah[MT], al[MT]; };
            const int* const tq = (const int*)s_t4 + grp.x * 4 + kg;        // the lane's k-group column of the group's table rows
            const int lanecol = col * 16;
            auto load = [&](int po, int sl, Frag& f) {
                const char* pa = pbase + po + lanecol;
                const char* wa = wbase + sl * (MT * NWL * 1024);
#pragma unroll
                for (int c = 0; c < NC; c++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
#if LM_G2_CUT == 7      // timing-only build: a quarter of the pixel-fragment LDS reads (rows 1..3 reuse row 0's registers)
                        if (r > 0) { f.bh[c * 4 + r] = f.bh[c * 4]; if (SPLIT_B) f.bl[c * 4 + r] = f.bl[c * 4]; continue; }
#endif
                        f.bh[c * 4 + r] = *(const lm_h8*)(pa + r * (PW * 16) + c * 256);
                        if (SPLIT_B) f.bl[c * 4 + r] = *(const lm_h8*)(pa + r * (PW * 16) + c * 256 + PLS);
                    }
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    f.ah[m] = *(const lm_h8*)(wa + m * (NWL * 1024));
                    if (SPLIT_A) f.al[m] = *(const lm_h8*)(wa + m * (NWL * 1024) + 1024);
                }
            };
            auto mma = [&](const Frag& f) {
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int n = 0; n < NT; n++) acc[m][n] = LM_MFMA16(f.ah[m], f.bh[n], acc[m][n]);
                if (SPLIT_A) {
#pragma unroll
                    for (int m = 0; m < MT; m++)
#pragma unroll
                        for (int n = 0; n < NT; n++) acc[m][n] = LM_MFMA16(f.al[m], f.bh[n], acc[m][n]);
                }
                if (SPLIT_B) {
#pragma unroll
                    for (int m = 0; m < MT; m++)
#pragma unroll
                        for (int n = 0; n < NT; n++) acc[m][n] = LM_MFMA16(f.ah[m], f.bl[n], acc[m][n]);
                }
            };
            // The group's slices, software-pipelined by hand with an UNCONDITIONAL body: the fragments of slice s + 1 are requested
            // before the MFMAs of slice s, the table word of slice s + 2 before that (the table is padded, the look-ahead needs no bounds).
            const int ns = grp.y;
            Frag f0, f1;
            int o1 = tq[4];
            load(tq[0], 0, f0);
            int sl = 0;
            for (; sl + 2 < ns; sl += 2) {
                const int o2 = tq[(sl + 2) * 4];
                load(o1, sl + 1, f1);
                LM_SCHED_BARRIER();         // the requests stay IN FRONT of the MFMAs they hide behind (left alone, the scheduler sank them to
                mma(f0);                    // the end of the MFMA block, right before the wait for them)
                o1 = tq[(sl + 3) * 4];
                load(o2, sl + 2, f0);
                LM_SCHED_BARRIER();
                mma(f1);
            }
            // one or two slices are left.  (An if / else with the MFMAs in both arms made the compiler accumulate into fresh registers in
            // either arm and copy all accumulators back at the merge: 32 v_mov_b64 per group at two channel tiles on 16 x 32 tiles.)
            const bool two = sl + 1 < ns;
            if (two) load(o1, sl + 1, f1);
            LM_SCHED_BARRIER();
            mma(f0);
            if (two) mma(f1);
        }
#if LM_G2_STAMPS && !LM_HIP_EMULATED
        { const unsigned long long t = LM_STAMP_NOW(); st_compute += t - st_mark; st_mark = t; }
#endif
        if (issuer) lm_vmwait(n_ahead);
#if LM_G2_STAMPS && !LM_HIP_EMULATED
        { const unsigned long long t = LM_STAMP_NOW(); st_wait += t - st_mark; st_mark = t; }
#endif
        lm_lds_barrier();       // this group's buffer is free; the next group's weights (and the next chunk's planes) have landed
#if LM_G2_STAMPS && !LM_HIP_EMULATED
        { const unsigned long long t = LM_STAMP_NOW(); st_bar += t - st_mark; st_mark = t; }
#endif
        if (++wb >= ring) wb = 0;
        if (more) {
            const int4 cur = grp;
            grp = nxt;
            nxt = nx2;
            if (grp.z != cur.z) {
                if (a.pdouble) pb ^= 1;
                else {              // one patch buffer: the next chunk's planes are fetched now, in the open
                    if (issuer) {
                        issue_patch(grp.z, 0);
                        LM_VMWAIT0();
                    }
                    lm_lds_barrier();
#if LM_G2_STAMPS && !LM_HIP_EMULATED
                    { const unsigned long long t = LM_STAMP_NOW(); st_switch += t - st_mark; st_mark = t; }
#endif
                }
            }
        }
    }
    if (!computes) return;
#if LM_G2_CUT == 4      // timing-only build: no epilogue (one value per lane keeps the MFMAs alive)
    {
        float sum = 0.0f;
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int n = 0; n < NT; n++) sum += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
        if (sum == 12345.678f) a.tout[threadIdx.x] = sum;
        return;
    }
#endif

    if constexpr (EPI == LM_G2_EPI_V) {
        // The wave's rows of the row convolution go to LDS (the loop's last barrier has freed the patch and weight buffers), then the
        // workgroup's threads finish LM_G2_V_ROWS x TW pixels: row oy of the result sums computed rows oy .. oy + 6 (text, output logit:
        // kernel row kh from computed row oy + kh) or oy + 2 .. oy + 4 (reconstruction, 3 x 3), exactly as lm_k_vsum2_text_rec / lm_k_vsum
        // do from the global row buffer (FCN_lecturenet.py:366-379, 395-403).  Computed rows outside the image come from zero planes.
        static_assert(MT == 1 && !LOADER, "the fused head runs one channel tile on four waves");
        float* const s_T = (float*)smem;                            // [16 rows][TW][16]
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int n = 0; n < 4; n++) {
                const lm_f32x4 v = acc[0][c * 4 + n];
                *(float4*)(s_T + (((wave * 4 + n) * TW + c * 16 + col) * 16 + 4 * kg)) = make_float4(v[0], v[1], v[2], v[3]);
            }
        lm_lds_barrier();
        const int yb = ty * LM_G2_V_ROWS, xb = tx * TW;
        for (int p = (int)threadIdx.x; p < LM_G2_V_ROWS * TW; p += NTHR) {
            const int oy = p / TW, px = p - oy * TW, y = yb + oy, x = xb + px;
            if (y >= a.H || x >= a.W) continue;
            const long long pi = (long long)y * a.W + x;
            if (a.vmode == 2) {
                float v = 0.f;
#pragma unroll
                for (int kh = 0; kh < 7; kh++) v += s_T[((oy + kh) * TW + px) * 16 + kh];
                a.v_out[pi] = v + a.vbias[0];
                continue;
            }
            float t = a.vbias[0], r0 = a.vbias[1], r1 = a.vbias[2], r2 = a.vbias[3];
#pragma unroll
            for (int kh = 0; kh < 7; kh++) t += s_T[((oy + kh) * TW + px) * 16 + kh];
#pragma unroll
            for (int kh = 0; kh < 3; kh++) {
                const float* q = s_T + ((oy + 2 + kh) * TW + px) * 16 + 7 + kh * 3;
                r0 += q[0]; r1 += q[1]; r2 += q[2];
            }
            r0 = tanhf(r0); r1 = tanhf(r1); r2 = tanhf(r2);
            const float m = 1.0f / (1.0f + expf(-t));
            const long long so = ((long long)(y + a.halo_v) * a.Wp_v + x + a.halo_v) * 16;
            const lm_h4 xh = *(const lm_h4*)(a.x0_hi + so), xl4 = *(const lm_h4*)(a.x0_lo + so);
            const float x0 = (float)xh[0] + (float)xl4[0], x1 = (float)xh[1] + (float)xl4[1], x2 = (float)xh[2] + (float)xl4[2];
            a.v_text[pi] = t;
            if (a.v_rec4) {         // the caller's [3][H][W] reconstruction image
                const long long npx = (long long)a.H * a.W;
                a.v_rec4[pi] = r0; a.v_rec4[npx + pi] = r1; a.v_rec4[2 * npx + pi] = r2;
            }
            lm_pair_store(a.dp_hi, a.dp_lo, a.Wp_v, a.halo_v, y, x, (x0 - r0) * m, (x1 - r1) * m, (x2 - r2) * m);
        }
        return;
    }
#if LM_G2_CUT == 3      // timing-only build: everything of the epilogue but the stores (their values and addresses feed a checksum)
    unsigned lm_cut_chk = 0;
#define LM_G2_STORE(T, addr, val) do { T v_ = (val); unsigned w_[sizeof(T) / 4]; __builtin_memcpy(w_, &v_, sizeof(T)); for (unsigned q_ = 0; q_ < sizeof(T) / 4; q_++) lm_cut_chk ^= w_[q_]; lm_cut_chk += (unsigned)(size_t)(addr); } while (0)
#else
#define LM_G2_STORE(T, addr, val) (*(T*)(addr) = (val))
#endif
    // ---------------------------------------------------------------- epilogues
    // D[row = 4 * kg + r][col]: row = channel of the tile, col = pixel `col` of the wave's row r in column tile c
    const int y0 = ty * 16 + wave * 4;
#pragma unroll
    for (int c = 0; c < NC; c++) {
    const int x = tx * TW + c * 16 + col;
    if constexpr (EPI == LM_G2_EPI_T) {
        static_assert(MT == 1, "the head rows have at most 16 outputs");
        if (4 * kg < a.tn && x < a.W) {
#pragma unroll
            for (int n = 0; n < 4; n++) {
                const int y = y0 + n;
                const lm_f32x4 v = acc[0][c * 4 + n];
                if (y < a.H) LM_G2_STORE(float4, a.tout + ((long long)y * a.W + x) * a.ts + 4 * kg, make_float4(v[0], v[1], v[2], v[3]));
            }
        }
    } else {
        // PO / TC epilogues: bias, GELU (every planar-output layer of the network has it; lm_f2_run refuses anything else), f16 hi (+ lo),
        // whole-octet stores.  Kept lean on the VALU -- timing-only builds (profiles/r04_cuts_*.txt) put the epilogues at 27 % of the frame,
        // 0.26 ms of it the stores themselves and the rest vector instructions: one base address per (column tile, channel pair) and a
        // scalar row stride, no per-value select on the activation, no fmaxf (two v_max_f32 per value for its NaN canonicalisation).
        const bool merged = (EPI == LM_G2_EPI_TC) && a.tc_merged;
        const int cb = merged ? cby * 32 : cby * (MT * 16);                  // first channel of the workgroup
        const int dy = merged ? par : (par >> 1), dx = par & 1;
        constexpr int sc = (EPI == LM_G2_EPI_TC) ? 2 : 1;
        const bool has_lo = a.out_lo != nullptr;
        const long long lo_delta = has_lo ? (long long)(a.out_lo - a.out_hi) : 0;
        const long long rstride = (long long)sc * a.Wp_out * 16;             // bytes between the wave's output rows
        const bool xin = x < a.W;
        // pairs of tiles: the host packs tile 2q with the channels 32q + 8kg + (0..3) in rows 4kg + (0..3) and tile 2q + 1 with
        // 32q + 8kg + 4 + (0..3), so a lane holds one whole octet of its pixel: one 16-byte store per part
#pragma unroll
        for (int q = 0; q < MT / 2; q++) {
            const int ch = cb + (merged ? 0 : 32 * q) + 8 * kg;
            const int dxq = merged ? q : dx;
            const float4 b0 = *(const float4*)(t_bias + ch), b1 = *(const float4*)(t_bias + ch + 4);
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            char* const obase = a.out_hi + (long long)(ch >> 3) * a.out_plane + ((long long)(sc * y0 + dy + a.halo_out) * a.Wp_out + sc * x + dxq + a.halo_out) * 16;
            float v[4][8];
#pragma unroll
            for (int n = 0; n < 4; n++)
#pragma unroll
                for (int j = 0; j < 8; j++) v[n][j] = lm_gelu_fast(acc[2 * q + (j >> 2)][c * 4 + n][j & 3] + bb[j]);
#pragma unroll
            for (int n = 0; n < 4; n++) {
                if (!xin || y0 + n >= a.H) continue;
                lm_h8 hi;
#pragma unroll
                for (int j = 0; j < 8; j++) hi[j] = (_Float16)v[n][j];
                char* const o = obase + n * rstride;
                LM_G2_STORE(lm_h8, o, hi);
                if (has_lo) {
                    lm_h8 lo;
#pragma unroll
                    for (int j = 0; j < 8; j++) lo[j] = (_Float16)(v[n][j] - (float)hi[j]);
                    LM_G2_STORE(lm_h8, o + lo_delta, lo);
                }
            }
            if constexpr (EPI == LM_G2_EPI_PO) {
                if (a.pool_hi) {        // 2x2 / stride 2 max pooling (floor): rows (n, n + 1), columns (col, col ^ 1)
#pragma unroll
                    for (int n = 0; n < 4; n += 2) {
                        lm_h8 ph, pl;
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const float m2 = v[n][j] > v[n + 1][j] ? v[n][j] : v[n + 1][j];
                            const float m3 = __shfl_xor(m2, 1);
                            const float m4 = m2 > m3 ? m2 : m3;
                            ph[j] = (_Float16)m4; pl[j] = (_Float16)(m4 - (float)ph[j]);
                        }
                        const int py = (y0 + n) >> 1, px = x >> 1;
                        if (!(col & 1) && py < (a.H >> 1) && px < (a.W >> 1)) {
                            const long long so = (long long)(ch >> 3) * a.pool_plane + ((long long)(py + a.halo_pool) * a.Wp_pool + px + a.halo_pool) * 16;
                            LM_G2_STORE(lm_h8, a.pool_hi + so, ph);
                            if (a.pool_lo) LM_G2_STORE(lm_h8, a.pool_lo + so, pl);
                        }
                    }
                }
            }
        }
        if constexpr (MT & 1) {         // the unpaired last tile: channels in row order, a lane holds half an octet (8-byte stores).  Measured: swapping
                                        // halves with the partner lane (^ 16) for whole-slot stores cost more in shuffles than it saved (+5..15 % on the MT = 3 layers)
            constexpr int m = MT - 1;
            const int ch = cb + 16 * m + 4 * kg;
            const float4 b0 = *(const float4*)(t_bias + ch);
            const float bb[4] = {b0.x, b0.y, b0.z, b0.w};
            char* const obase = a.out_hi + (long long)(ch >> 3) * a.out_plane + (kg & 1) * 8 + ((long long)(sc * y0 + dy + a.halo_out) * a.Wp_out + sc * x + dx + a.halo_out) * 16;
            float v[4][4];
#pragma unroll
            for (int n = 0; n < 4; n++)
#pragma unroll
                for (int j = 0; j < 4; j++) v[n][j] = lm_gelu_fast(acc[m][c * 4 + n][j] + bb[j]);
#pragma unroll
            for (int n = 0; n < 4; n++) {
                if (!xin || y0 + n >= a.H) continue;
                lm_h4 hi;
#pragma unroll
                for (int j = 0; j < 4; j++) hi[j] = (_Float16)v[n][j];
                char* const o = obase + n * rstride;
                LM_G2_STORE(lm_h4, o, hi);
                if (has_lo) {
                    lm_h4 lo;
#pragma unroll
                    for (int j = 0; j < 4; j++) lo[j] = (_Float16)(v[n][j] - (float)hi[j]);
                    LM_G2_STORE(lm_h4, o + lo_delta, lo);
                }
            }
            if constexpr (EPI == LM_G2_EPI_PO) {
                if (a.pool_hi) {
#pragma unroll
                    for (int n = 0; n < 4; n += 2) {
                        lm_h4 ph, pl;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const float m2 = v[n][j] > v[n + 1][j] ? v[n][j] : v[n + 1][j];
                            const float m3 = __shfl_xor(m2, 1);
                            const float m4 = m2 > m3 ? m2 : m3;
                            ph[j] = (_Float16)m4; pl[j] = (_Float16)(m4 - (float)ph[j]);
                        }
                        const int py = (y0 + n) >> 1, px = x >> 1;
                        if (!(col & 1) && py < (a.H >> 1) && px < (a.W >> 1)) {
                            const long long so = (long long)(ch >> 3) * a.pool_plane + (kg & 1) * 8 + ((long long)(py + a.halo_pool) * a.Wp_pool + px + a.halo_pool) * 16;
                            LM_G2_STORE(lm_h4, a.pool_hi + so, ph);
                            if (a.pool_lo) LM_G2_STORE(lm_h4, a.pool_lo + so, pl);
                        }
                    }
                }
            }
        }
    }
    }
#if LM_G2_CUT == 3
    if (lm_cut_chk == 0x12345678u) a.tout[threadIdx.x] = 1.0f;
#endif
#undef LM_G2_STORE
#if LM_G2_STAMPS && !LM_HIP_EMULATED
    if (a.stamp && lane == 0) {
        const unsigned long long t_loop = st_mark;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = LM_STAMP_NOW();
        const size_t w = (size_t)blockIdx.x * 4 + wave;
        if (w < LM_G2_STAMP_WAVES) {
            unsigned long long* o = lm_g2_stamp_buf + w * LM_G2_NSTAMP;
            o[0] = st_t0; o[1] = st_pro; o[2] = st_compute; o[3] = st_wait; o[4] = st_switch; o[5] = t_loop; o[6] = t_end; o[7] = __builtin_amdgcn_s_memrealtime(); o[8] = st_r0; o[9] = st_issue; o[10] = st_bar; o[11] = 0;
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// pair planes and heads
// ------------------------------------------------------------------------------------------------
// Writes one pixel's three values into a pair plane: slot x gets them as its first half {c0 c1 c2 0}, slot x - 1 as its second half
// (x - 1 = -1 lies in the halo: that slot's first half stays zero = the convolution's padding).  hi and lo parts.
LM_DEV void lm_pair_store(char* hi_plane, char* lo_plane, int Wp, int halo, int y, int x, float v0, float v1, float v2)
{
    lm_h4 hi, lo;
    hi[0] = (_Float16)v0; hi[1] = (_Float16)v1; hi[2] = (_Float16)v2; hi[3] = (_Float16)0.0f;
    lo[0] = (_Float16)(v0 - (float)hi[0]); lo[1] = (_Float16)(v1 - (float)hi[1]); lo[2] = (_Float16)(v2 - (float)hi[2]); lo[3] = (_Float16)0.0f;
    const long long so = ((long long)(y + halo) * Wp + x + halo) * 16;
    *(lm_h4*)(hi_plane + so) = hi;
    *(lm_h4*)(hi_plane + so - 8) = hi;
    *(lm_h4*)(lo_plane + so) = lo;
    *(lm_h4*)(lo_plane + so - 8) = lo;
}

// uint8 HWC RGB -> the network's input pair plane, (x / 255 - 0.5) / 0.5 (to_tensor + normalize, FCN_lecturenet.py:607-618)
__global__ void __launch_bounds__(256) lm_k_prepare2(const uint8_t* __restrict__ rgb, char* hi_plane, char* lo_plane, int H, int W, int Wp, int halo)
{
    const long long npx = (long long)H * W;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(p / W), x = (int)(p - (long long)y * W);
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; c++) v[c] = ((float)rgb[p * 3 + c] / 255.0f - 0.5f) / 0.5f;
        lm_pair_store(hi_plane, lo_plane, Wp, halo, y, x, v[0], v[1], v[2]);
    }
}

// Text mask (7x7, one output) and reconstruction (3x3, three outputs) from their common 1x7 row convolution T[pixel][16] (rows 0..6:
// text kernel rows, 7 + kh * 3 + co: reconstruction rows, see fcn.pack_text_rec_rows): text logit, rec = tanh(.),
// diff = (x0 - rec) * sigmoid(text) (:370-379) written as the pixel branch's pair plane.  bias: [0] text, [1..3] rec.
__global__ void __launch_bounds__(256) lm_k_vsum2_text_rec(const float* __restrict__ T, int H, int W, const float* __restrict__ bias,
                                                           const char* x0_hi, const char* x0_lo, float* __restrict__ text, float* __restrict__ rec4,
                                                           char* dp_hi, char* dp_lo, int Wp, int halo)
{
    constexpr int TW = 32, TH = 16, PR = TH + 6, TS = 16, RQ = TW * TS / 4;
    __shared__ float s_t[PR * TW * TS];         // 44 KB
    const int tiles_x = (W + TW - 1) / TW;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    for (int i = threadIdx.x; i < PR * RQ; i += blockDim.x) {
        const int r = i / RQ, q = i - r * RQ;
        const int y = ty0 + r - 3, x = tx0 + (q >> 2);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < H && x < W) v = *(const float4*)(T + ((long long)y * W + tx0) * TS + q * 4);
        *(float4*)(s_t + (r * TW) * TS + q * 4) = v;
    }
    __syncthreads();
    const int ly = (int)(threadIdx.x >> 4), lx = (int)(threadIdx.x & 15);
    const int y = ty0 + ly;
    if (y >= H) return;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int xl = lx + 16 * h, x = tx0 + xl;
        if (x >= W) continue;
        float t = bias[0], r0 = bias[1], r1 = bias[2], r2 = bias[3];
#pragma unroll
        for (int kh = 0; kh < 7; kh++) t += s_t[((ly + kh) * TW + xl) * TS + kh];
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
            const float* q = s_t + ((ly + 2 + kh) * TW + xl) * TS + 7 + kh * 3;
            r0 += q[0]; r1 += q[1]; r2 += q[2];
        }
        r0 = tanhf(r0); r1 = tanhf(r1); r2 = tanhf(r2);
        const long long p = (long long)y * W + x;
        const float m = 1.0f / (1.0f + expf(-t));
        const long long so = ((long long)(y + halo) * Wp + x + halo) * 16;
        const lm_h4 xh = *(const lm_h4*)(x0_hi + so), xl4 = *(const lm_h4*)(x0_lo + so);
        const float x0 = (float)xh[0] + (float)xl4[0], x1 = (float)xh[1] + (float)xl4[1], x2 = (float)xh[2] + (float)xl4[2];
        text[p] = t;
        *(float4*)(rec4 + p * 4) = make_float4(r0, r1, r2, 0.f);
        lm_pair_store(dp_hi, dp_lo, Wp, halo, y, x, (x0 - r0) * m, (x1 - r1) * m, (x2 - r2) * m);
    }
}

// rows / columns of a transposed-conv output that no input pixel reaches (output_size = 2 * in + 1): act(bias), planar octets
__global__ void __launch_bounds__(256) lm_k_convT_border2(char* out_hi, char* out_lo, long long plane, int Wp, int halo, int OH, int OW, int H2, int W2,
                                                          int C8, const float* __restrict__ bias, int act)
{
    const long long nb_px = (long long)(OH - H2) * OW + (long long)H2 * (OW - W2);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nb_px * C8; i += (long long)gridDim.x * blockDim.x) {
        const int o = (int)(i / nb_px);
        const long long p = i - (long long)o * nb_px;
        int y, x;
        if (p < (long long)(OH - H2) * OW) { y = H2 + (int)(p / OW); x = (int)(p % OW); }
        else { const long long q = p - (long long)(OH - H2) * OW; y = (int)(q / (OW - W2)); x = W2 + (int)(q % (OW - W2)); }
        lm_h8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float v = lm_act(bias[o * 8 + j], act);
            hi[j] = (_Float16)v; lo[j] = (_Float16)(v - (float)hi[j]);
        }
        const long long so = (long long)o * plane + ((long long)(y + halo) * Wp + x + halo) * 16;
        *(lm_h8*)(out_hi + so) = hi;
        if (out_lo) *(lm_h8*)(out_lo + so) = lo;
    }
}

// ================================================================================================
// host side
// ================================================================================================
#define LM_F2_TENSORS 25
enum { LM_F2_X0P = 0, LM_F2_PRE0 = 1, LM_F2_POOL0 = 6, LM_F2_MID = 11, LM_F2_UPT0 = 12, LM_F2_CU0 = 17, LM_F2_XUP = 21, LM_F2_DP = 22, LM_F2_P1 = 23, LM_F2_P2 = 24 };

struct LmF2Tensor {
    int c8 = 0, lo = 0, level = 0;      // octet planes, lo parts kept, pyramid level (0 = full resolution)
    long long off = 0;                  // byte offset in the arena
    // geometry for the current frame size
    int H = 0, W = 0, halo = 0, Hp = 0, Wp = 0;
    long long plane = 0;
};

struct LmF2Layer {
    int kh = 0, kw = 0, terms = 0, mt = 0, epi = 0, nchunks = 0, npc = 0, ngroups = 0, nslices = 0, pdouble = 0, wbuf_bytes = 0, cout = 0, nc = 1, loader = 0, lds_kb = 0;
    std::vector<int> planes;            // [nchunks * npc][2] tensor id, octet
    long long wblock_bytes = 0;
    char* d_w = nullptr; float* d_bias = nullptr;
    int4* d_groups = nullptr; int4* d_t4 = nullptr; long long* d_psrc = nullptr;
    bool set = false;
};

struct LmFcn2 {
    int widths[18];
    int max_h, max_w;
    char* arena = nullptr;
    long long arena_bytes = 0;
    LmF2Tensor t[LM_F2_TENSORS];
    LmF2Layer layer[LM_FCN_LAYERS];
    float *tbuf = nullptr, *text = nullptr, *rec4 = nullptr, *outl = nullptr;
    int cur_h = 0, cur_w = 0;
};

static inline int lm_f2_halo(int level) { return level == 0 ? 3 : 1; }

static void lm_f2_geometry(LmFcn2* f, int h, int w)
{
    for (auto& t : f->t) {
        t.H = h >> t.level; t.W = w >> t.level; t.halo = lm_f2_halo(t.level);
        t.Hp = ((t.H + 15) & ~15) + 2 * t.halo + (t.level == 0 ? 16 : 0);      // whole tiles; full resolution: + the overhang of the fused heads' 10-row tile pitch
        t.Wp = ((t.W + 31) & ~31) + 2 * t.halo;
        t.plane = (long long)t.Hp * t.Wp * 16;
    }
}

extern "C" void lm_fcn2_destroy(LmFcn2* f)
{
    if (!f) return;
    if (f->arena) (void)hipFree(f->arena);
    for (float* p : {f->tbuf, f->text, f->rec4, f->outl}) if (p) (void)hipFree(p);
    for (auto& l : f->layer)
        for (void* p : {(void*)l.d_w, (void*)l.d_bias, (void*)l.d_groups, (void*)l.d_t4, (void*)l.d_psrc}) if (p) (void)hipFree(p);
    delete f;
}

// widths18 as lm_fcn_create (every width a multiple of 16); lo25[t] = 1 keeps the lo parts of tensor t (a consumer runs a split format)
extern "C" LmFcn2* lm_fcn2_create(const int32_t* widths18, const int32_t* lo25, int max_h, int max_w)
{
    if (!widths18 || !lo25 || max_h < 32 || max_w < 32) { lm_set_error("lm_fcn2_create: bad arguments (frames must be at least 32 px per side)"); return nullptr; }
    for (int i = 0; i < 18; i++)
        if (widths18[i] <= 0 || (widths18[i] & 15)) { lm_set_error("lm_fcn2_create: layer widths must be positive multiples of 16"); return nullptr; }
    LmFcn2* f = new LmFcn2();
    memcpy(f->widths, widths18, sizeof(f->widths));
    f->max_h = max_h; f->max_w = max_w;
    const int* w = f->widths;
    auto def = [&](int id, int channels, int level) { f->t[id].c8 = channels / 8; f->t[id].level = level; f->t[id].lo = lo25[id] ? 1 : 0; };
    def(LM_F2_X0P, 8, 0); f->t[LM_F2_X0P].lo = 1;
    for (int n = 0; n < 5; n++) { def(LM_F2_PRE0 + n, w[n], n); def(LM_F2_POOL0 + n, w[n], n + 1); }
    def(LM_F2_MID, w[5], 5);
    for (int n = 0; n < 5; n++) def(LM_F2_UPT0 + n, w[6 + 2 * n], 4 - n);
    for (int n = 0; n < 4; n++) def(LM_F2_CU0 + n, w[7 + 2 * n], 4 - n);
    def(LM_F2_XUP, w[15], 0);
    def(LM_F2_DP, 8, 0); f->t[LM_F2_DP].lo = 1;
    def(LM_F2_P1, w[16], 0);
    def(LM_F2_P2, w[17], 0);
    lm_f2_geometry(f, max_h, max_w);
    long long off = 0;
    for (auto& t : f->t) { t.off = off; off += (long long)t.c8 * (1 + t.lo) * t.plane + 4096; }
    f->arena_bytes = off;
    if (off >= (1ll << 32) - (1 << 20)) {        // lm_k_g2 addresses the arena through ONE raw buffer with 32-bit offsets
        lm_set_error("lm_fcn2_create: %lld MB of activations for %dx%d frames exceed the 4 GB one buffer addresses (the network runs at <= 2.5 MP)", off >> 20, max_w, max_h);
        delete f;
        return nullptr;
    }
    const size_t px = (size_t)max_h * max_w;
    if (hipMalloc((void**)&f->arena, (size_t)off) != hipSuccess || hipMalloc((void**)&f->tbuf, px * 16 * 4) != hipSuccess ||
        hipMalloc((void**)&f->text, px * 4) != hipSuccess || hipMalloc((void**)&f->rec4, px * 16) != hipSuccess ||
        hipMalloc((void**)&f->outl, px * 4) != hipSuccess) {
        lm_set_error("lm_fcn2_create: out of device memory (%lld MB of activations)", off >> 20);
        lm_fcn2_destroy(f);
        return nullptr;
    }
    return f;
}

// One layer's recipe (lecturemath_amd/fcn2.py builds it and documents the layout).  desc: kh, kw, terms, mt, epi, nchunks, npc, ngroups,
// nslices, flags (bits 0-3: column tiles per wave, 1 or 2; bit 8: loader wave; bits 16-23: LDS target in KB), pdouble, wbuf_bytes, cout, then planes [nchunks * npc][2], groups [ngroups][3], slice table [nslices][4].
// HOST pointers.  wblocks = channel blocks (x 4 parities for a transposed convolution) of wbytes / wblocks bytes each.
extern "C" int lm_fcn2_set_layer(LmFcn2* f, int layer, const int32_t* desc, int ndesc, const void* h_w, int64_t wbytes, int wblocks, const float* h_bias,
                                 int nbias)
{
    if (!f || layer < 0 || layer >= LM_FCN_LAYERS || !desc || ndesc < 13 || !h_w || wbytes <= 0 || wblocks <= 0 || !h_bias || nbias <= 0) {
        lm_set_error("lm_fcn2_set_layer: bad arguments");
        return LM_ERR_ARG;
    }
    LmF2Layer& l = f->layer[layer];
    for (void* p : {(void*)l.d_w, (void*)l.d_bias, (void*)l.d_groups, (void*)l.d_t4, (void*)l.d_psrc}) if (p) (void)hipFree(p);
    l = LmF2Layer();
    l.kh = desc[0]; l.kw = desc[1]; l.terms = desc[2]; l.mt = desc[3]; l.epi = desc[4]; l.nchunks = desc[5]; l.npc = desc[6]; l.ngroups = desc[7];
    l.nslices = desc[8]; l.nc = (desc[9] & 15) ? (desc[9] & 15) : 1; l.loader = (desc[9] >> 8) & 1; l.lds_kb = (desc[9] >> 16) & 0xff; l.pdouble = desc[10]; l.wbuf_bytes = desc[11]; l.cout = desc[12];
    const long long need = 13 + (long long)l.nchunks * l.npc * 2 + (long long)l.ngroups * 3 + (long long)l.nslices * 4;
    if (l.nchunks <= 0 || l.npc <= 0 || l.ngroups <= 0 || l.nslices <= 0 || need != ndesc || wbytes % wblocks || wbytes >= (1ll << 31) || l.nc < 1 || l.nc > 2) {
        lm_set_error("lm_fcn2_set_layer: inconsistent recipe for layer %d", layer);
        return LM_ERR_ARG;
    }
    const int32_t* p = desc + 13;
    l.planes.assign(p, p + (size_t)l.nchunks * l.npc * 2); p += (size_t)l.nchunks * l.npc * 2;
    for (size_t i = 0; i < l.planes.size(); i += 2)
        if (l.planes[i] < 0 || l.planes[i] >= LM_F2_TENSORS || l.planes[i + 1] < 0 || l.planes[i + 1] >= f->t[l.planes[i]].c8 ||
            ((l.terms == 2 || l.terms == 3) && !f->t[l.planes[i]].lo)) {
            lm_set_error("lm_fcn2_set_layer: layer %d reads plane %d of tensor %d (octets %d, lo %d)", layer, l.planes[i + 1], l.planes[i],
                         f->t[l.planes[i] < 0 || l.planes[i] >= LM_F2_TENSORS ? 0 : l.planes[i]].c8, f->t[l.planes[i] < 0 || l.planes[i] >= LM_F2_TENSORS ? 0 : l.planes[i]].lo);
            return LM_ERR_ARG;
        }
    l.wblock_bytes = wbytes / wblocks;
    std::vector<int4> groups((size_t)l.ngroups);
    const int nwl = l.terms >= 3 ? 2 : 1;
    for (int g = 0; g < l.ngroups; g++, p += 3) {
        groups[g] = make_int4(p[0], p[1], p[2], p[0] * l.mt * nwl * 1024);
        if (p[0] < 0 || p[1] <= 0 || p[0] + p[1] > l.nslices || p[2] < 0 || p[2] >= l.nchunks || p[1] * l.mt * nwl * 1024 > l.wbuf_bytes) {
            lm_set_error("lm_fcn2_set_layer: bad weight group %d of layer %d", g, layer);
            return LM_ERR_ARG;
        }
    }
    if ((long long)l.nslices * l.mt * nwl * 1024 != l.wblock_bytes) { lm_set_error("lm_fcn2_set_layer: layer %d: weight bytes do not match the slices", layer); return LM_ERR_ARG; }
    // slice table: every offset must leave the lane's reads (16 x 16 pixels, four rows per wave) inside the chunk's patch planes
    std::vector<int4> t4((size_t)l.nslices + 2);
    {
        const int PW = 16 * l.nc + l.kw - 1, PH = 16 + l.kh - 1, PLS = (PH * PW * 16 + 255) & ~255, nhl = (l.terms == 2 || l.terms == 3) ? 2 : 1;
        for (int i = 0; i < l.nslices; i++, p += 4) {
            t4[i] = make_int4(p[0], p[1], p[2], p[3]);
            for (int k = 0; k < 4; k++) {
                const int plane = p[k] / (nhl * PLS), rest = p[k] - plane * nhl * PLS;
                if (p[k] < 0 || plane >= l.npc || rest % 16 || rest / 16 / PW > l.kh - 1 || rest / 16 % PW > l.kw - 1) {
                    lm_set_error("lm_fcn2_set_layer: slice %d of layer %d reads outside its patch", i, layer);
                    return LM_ERR_ARG;
                }
            }
        }
        t4[l.nslices] = t4[l.nslices + 1] = t4[l.nslices - 1];
    }
    LM_HIP(hipMalloc((void**)&l.d_w, (size_t)wbytes));
    LM_HIP(hipMalloc((void**)&l.d_bias, (size_t)nbias * 4));
    LM_HIP(hipMalloc((void**)&l.d_groups, groups.size() * sizeof(int4)));
    LM_HIP(hipMalloc((void**)&l.d_t4, t4.size() * sizeof(int4)));             // + 2: the kernel's look-ahead
    LM_HIP(hipMalloc((void**)&l.d_psrc, (size_t)l.nchunks * l.npc * 16));
    LM_HIP(hipMemcpy(l.d_w, h_w, (size_t)wbytes, hipMemcpyHostToDevice));
    LM_HIP(hipMemcpy(l.d_bias, h_bias, (size_t)nbias * 4, hipMemcpyHostToDevice));
    LM_HIP(hipMemcpy(l.d_groups, groups.data(), groups.size() * sizeof(int4), hipMemcpyHostToDevice));
    LM_HIP(hipMemcpy(l.d_t4, t4.data(), t4.size() * sizeof(int4), hipMemcpyHostToDevice));
    l.set = true;
    f->cur_h = f->cur_w = 0;        // plane tables are rebuilt by the next forward
    return LM_OK;
}

template <int KH, int KW, int TERMS, int MT, int EPI, int NC, int LOADER> static int lm_g2_launch_t(const LmG2Args& a, dim3 grid, size_t smem, hipStream_t st)
{
#if !LM_HIP_EMULATED
    LM_HIP(hipFuncSetAttribute((const void*)lm_k_g2<KH, KW, TERMS, MT, EPI, NC, LOADER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
#endif
    hipLaunchKernelGGL((lm_k_g2<KH, KW, TERMS, MT, EPI, NC, LOADER>), grid, dim3(LOADER ? 320 : 256), smem, st, a, a.psrc, a.groups, a.t4, a.bias);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// The instances that exist (each is a kernel of its own in the code object; lecturemath_amd/fcn2.py only asks for these):
//   variant 0 = 16 x 16 tile, four waves;  1 = 16 x 16 tile + loader wave;  2 = 16 x 32 tile, four waves
//   3 x 3 convolutions: formats 1 / 3 / 4, 1..4 channel tiles (loader: format 1, format 4 with <= 2 tiles; wide: <= 3 tiles on f16, <= 2 on a split format)
//   7 x 7 convolutions: formats 1..4, 1..2 channel tiles, all variants;  1 x 7 head rows: formats 1..4, variants 0 and 2; fused with their
//   vertical sums (EPI_V): formats 3 / 4, variants 0 and 2
//   transposed convolutions: formats 1 / 3, 1..4 channel tiles, variant 0
#define LM_G2_TRY(KH, KW, T, M, E, N, L) \
    if (l.terms == T && l.mt == M && l.nc == N && l.loader == L) return lm_g2_launch_t<KH, KW, T, M, E, N, L>(a, grid, smem, st);
#define LM_G2_TRY_MT4(KH, KW, T, E, N, L) LM_G2_TRY(KH, KW, T, 1, E, N, L) LM_G2_TRY(KH, KW, T, 2, E, N, L) LM_G2_TRY(KH, KW, T, 3, E, N, L) LM_G2_TRY(KH, KW, T, 4, E, N, L)
static int lm_g2_launch(const LmF2Layer& l, const LmG2Args& a, dim3 grid, size_t smem, hipStream_t st)
{
    const int shape = l.kh * 10 + l.kw;
    if (l.terms < 1 || l.terms > 4) { lm_set_error("lm_fcn2: operand formats are 1 (f16), 2 (activations split), 3 (both split) or 4 (weights split)"); return LM_ERR_ARG; }
    if (shape == 33 && l.epi == LM_G2_EPI_PO) {
        LM_G2_TRY_MT4(3, 3, 1, LM_G2_EPI_PO, 1, 0) LM_G2_TRY_MT4(3, 3, 3, LM_G2_EPI_PO, 1, 0) LM_G2_TRY_MT4(3, 3, 4, LM_G2_EPI_PO, 1, 0)
        LM_G2_TRY_MT4(3, 3, 1, LM_G2_EPI_PO, 1, 1) LM_G2_TRY(3, 3, 4, 1, LM_G2_EPI_PO, 1, 1) LM_G2_TRY(3, 3, 4, 2, LM_G2_EPI_PO, 1, 1)
        LM_G2_TRY(3, 3, 1, 1, LM_G2_EPI_PO, 2, 0) LM_G2_TRY(3, 3, 1, 2, LM_G2_EPI_PO, 2, 0) LM_G2_TRY(3, 3, 1, 3, LM_G2_EPI_PO, 2, 0)
        LM_G2_TRY(3, 3, 3, 1, LM_G2_EPI_PO, 2, 0) LM_G2_TRY(3, 3, 3, 2, LM_G2_EPI_PO, 2, 0)       // (three tiles of a split format spill)
        LM_G2_TRY(3, 3, 4, 1, LM_G2_EPI_PO, 2, 0) LM_G2_TRY(3, 3, 4, 2, LM_G2_EPI_PO, 2, 0)
    } else if (shape == 11 && (l.epi == LM_G2_EPI_TC || l.epi == LM_G2_EPI_TC2)) {
        if (l.epi == LM_G2_EPI_TC2 && l.mt != 4) { lm_set_error("lm_fcn2: the merged transposed convolution runs four channel tiles per workgroup"); return LM_ERR_ARG; }
        LM_G2_TRY_MT4(1, 1, 1, LM_G2_EPI_TC, 1, 0) LM_G2_TRY_MT4(1, 1, 3, LM_G2_EPI_TC, 1, 0)
    } else if (shape == 17 && l.epi == LM_G2_EPI_T) {
        LM_G2_TRY(1, 7, 1, 1, LM_G2_EPI_T, 1, 0) LM_G2_TRY(1, 7, 2, 1, LM_G2_EPI_T, 1, 0) LM_G2_TRY(1, 7, 3, 1, LM_G2_EPI_T, 1, 0) LM_G2_TRY(1, 7, 4, 1, LM_G2_EPI_T, 1, 0)
        LM_G2_TRY(1, 7, 1, 1, LM_G2_EPI_T, 2, 0) LM_G2_TRY(1, 7, 2, 1, LM_G2_EPI_T, 2, 0) LM_G2_TRY(1, 7, 3, 1, LM_G2_EPI_T, 2, 0) LM_G2_TRY(1, 7, 4, 1, LM_G2_EPI_T, 2, 0)
    } else if (shape == 17 && l.epi == LM_G2_EPI_V) {
        LM_G2_TRY(1, 7, 3, 1, LM_G2_EPI_V, 1, 0) LM_G2_TRY(1, 7, 4, 1, LM_G2_EPI_V, 1, 0) LM_G2_TRY(1, 7, 3, 1, LM_G2_EPI_V, 2, 0) LM_G2_TRY(1, 7, 4, 1, LM_G2_EPI_V, 2, 0)
    } else if (shape == 77 && l.epi == LM_G2_EPI_PO) {
#define LM_G2_TRY77(N, L) \
        LM_G2_TRY(7, 7, 1, 1, LM_G2_EPI_PO, N, L) LM_G2_TRY(7, 7, 2, 1, LM_G2_EPI_PO, N, L) LM_G2_TRY(7, 7, 3, 1, LM_G2_EPI_PO, N, L) LM_G2_TRY(7, 7, 4, 1, LM_G2_EPI_PO, N, L) \
        LM_G2_TRY(7, 7, 1, 2, LM_G2_EPI_PO, N, L) LM_G2_TRY(7, 7, 2, 2, LM_G2_EPI_PO, N, L) LM_G2_TRY(7, 7, 3, 2, LM_G2_EPI_PO, N, L) LM_G2_TRY(7, 7, 4, 2, LM_G2_EPI_PO, N, L)
        LM_G2_TRY77(1, 0) LM_G2_TRY77(1, 1) LM_G2_TRY77(2, 0)
#undef LM_G2_TRY77
    }
    lm_set_error("lm_fcn2: no kernel for a %dx%d layer: format %d, %d channel tiles, epilogue %d, %d column tiles, loader %d", l.kh, l.kw, l.terms, l.mt, l.epi, l.nc, l.loader);
    return LM_ERR_ARG;
}
#undef LM_G2_TRY
#undef LM_G2_TRY_MT4

// launches layer `li`: input planes per the recipe; `out` (EPI_PO / EPI_TC) with an optional pooled copy, or the T rows (EPI_T)
// EPI_V layers: vmode 1 (text mask + reconstruction + diff; rec4 may be null) or 2 (output logit)
static int lm_f2_run(LmFcn2* f, int li, const LmF2Tensor* out, const LmF2Tensor* pool, int act, float* tout, int ts, int tn, hipStream_t st, int vmode = 0,
                     float* rec4 = nullptr, float* text = nullptr, float* outl = nullptr)
{
    const LmF2Layer& l = f->layer[li];
    if (!l.set) { lm_set_error("lm_fcn2_forward: layer %d has no weights (call lm_fcn2_set_layer)", li); return LM_ERR_STATE; }
    const LmF2Tensor& in = f->t[l.planes[0]];
    LmG2Args a;
    memset(&a, 0, sizeof(a));
    a.arena = f->arena; a.psrc = l.d_psrc; a.wpk = l.d_w; a.groups = l.d_groups; a.t4 = l.d_t4; a.bias = l.d_bias;
    a.wblock_bytes = l.wblock_bytes; a.nchunks = l.nchunks; a.npc = l.npc; a.ngroups = l.ngroups; a.nslices = l.nslices; a.wbuf_bytes = l.wbuf_bytes;
    a.pdouble = l.pdouble; a.Wp_in = in.Wp;
    a.org_in = (in.halo - (l.kh - 1) / 2) * in.Wp + in.halo - (l.kw - 1) / 2;
    a.H = in.H; a.W = in.W;         // a convolution's output grid is its input grid; EPI_TC bounds its stores by the input grid
    a.tiles_x = (in.W + 16 * l.nc - 1) / (16 * l.nc);
    a.act = act;
    if ((l.epi == LM_G2_EPI_PO || l.epi == LM_G2_EPI_TC || l.epi == LM_G2_EPI_TC2) && act != LM_ACT_GELU) { lm_set_error("lm_fcn2: the planar-output epilogues apply GELU"); return LM_ERR_ARG; }
    if (out) {
        a.out_hi = f->arena + out->off; a.out_lo = out->lo ? a.out_hi + (long long)out->c8 * out->plane : nullptr;
        a.out_plane = out->plane; a.Wp_out = out->Wp; a.halo_out = out->halo;
        if (l.cout != out->c8 * 8 * (l.epi == LM_G2_EPI_TC2 ? 2 : 1)) { lm_set_error("lm_fcn2_forward: layer %d has %d outputs for a tensor of %d channels", li, l.cout, out->c8 * 8); return LM_ERR_STATE; }
    }
    if (pool) { a.pool_hi = f->arena + pool->off; a.pool_lo = pool->lo ? a.pool_hi + (long long)pool->c8 * pool->plane : nullptr; a.pool_plane = pool->plane; a.Wp_pool = pool->Wp; a.halo_pool = pool->halo; }
    a.tout = tout; a.ts = ts; a.tn = tn;
    a.tile_rows = 16;
    if (l.epi == LM_G2_EPI_V) {
        // 16 computed rows per tile starting 3 above the tile's first finished row: plane row (halo - 3) + ty * 10
        if (vmode != 1 && vmode != 2) { lm_set_error("lm_fcn2_forward: layer %d is a fused head", li); return LM_ERR_STATE; }
        const LmF2Tensor &x0 = f->t[LM_F2_X0P], &dp = f->t[LM_F2_DP];
        a.tile_rows = LM_G2_V_ROWS;
        a.org_in = (in.halo - 3) * in.Wp + in.halo - (l.kw - 1) / 2;
        a.vmode = vmode; a.vbias = l.d_bias + 16;
        a.x0_hi = f->arena + x0.off; a.x0_lo = a.x0_hi + x0.plane; a.dp_hi = f->arena + dp.off; a.dp_lo = a.dp_hi + dp.plane;
        a.Wp_v = dp.Wp; a.halo_v = dp.halo;
        a.v_text = text ? text : f->text; a.v_rec4 = rec4; a.v_out = outl ? outl : f->outl;
        if (in.halo < 3 || x0.Wp != dp.Wp || x0.halo != dp.halo) { lm_set_error("lm_fcn2_forward: pair planes of different geometry"); return LM_ERR_STATE; }
    }
    const int nhl = (l.terms == 2 || l.terms == 3) ? 2 : 1;
    const int PW = 16 * l.nc + l.kw - 1, PH = 16 + l.kh - 1, PLS = (PH * PW * 16 + 255) & ~255;
    // weight ring: three buffers when they fit beside two resident workgroups (or the layer cannot have two anyway), else two
    const size_t fixed = (size_t)(((l.nslices + 2) * 16 + 255) & ~255) + (size_t)(l.pdouble ? 2 : 1) * l.npc * nhl * PLS;
    static const int ring_env = [] { const char* e = getenv("LM_FCN2_RING"); return e ? atoi(e) : 0; }();
    int ring = l.ngroups > 2 ? 3 : (l.ngroups > 1 ? 2 : 1);
    const size_t lds_limit = (size_t)(l.lds_kb ? l.lds_kb : 80) * 1024;     // what the recipe was sized for (80 KB: two workgroups per CU)
    if (ring == 3 && fixed + 3 * (size_t)l.wbuf_bytes > lds_limit && fixed + 2 * (size_t)l.wbuf_bytes <= lds_limit) ring = 2;
    if (ring == 3 && fixed + 3 * (size_t)l.wbuf_bytes > 160 * 1024) ring = 2;
    if (ring_env == 2 && ring == 3) ring = 2;
    a.wring = ring < 2 ? 2 : ring;
    size_t smem = fixed + (size_t)ring * l.wbuf_bytes;
    if (l.epi == LM_G2_EPI_V && smem < (size_t)16 * 16 * l.nc * 16 * 4) smem = (size_t)16 * 16 * l.nc * 16 * 4;        // the computed rows, [16][TW][16] floats
    if (smem > 160 * 1024) { lm_set_error("lm_fcn2_forward: layer %d needs %zu bytes of LDS", li, smem); return LM_ERR_STATE; }
    a.tiles_y = (in.H + a.tile_rows - 1) / a.tile_rows;
    const int tiles = a.tiles_x * a.tiles_y;
    const int blocks = l.cout / (16 * l.mt);
    if (blocks * 16 * l.mt != l.cout) { lm_set_error("lm_fcn2_forward: layer %d: %d outputs are not whole blocks of %d tiles", li, l.cout, l.mt); return LM_ERR_STATE; }
    a.cblocks = blocks;
    a.tc_merged = l.epi == LM_G2_EPI_TC2 ? 1 : 0;
    static const int stamp_layer = [] { const char* e = getenv("LM_G2_STAMP_LAYER"); return e ? atoi(e) : -1; }();
    a.stamp = li == stamp_layer;
    return lm_g2_launch(l, a, dim3((unsigned)tiles * blocks * (l.epi == LM_G2_EPI_TC ? 4 : (l.epi == LM_G2_EPI_TC2 ? 2 : 1))), smem, st);
}

#if LM_G2_STAMPS && !LM_HIP_EMULATED
// diagnostic builds: copies the stamp buffer (LM_G2_STAMP_WAVES x LM_G2_NSTAMP u64) to the host
extern "C" int lm_debug_g2_read_stamps(unsigned long long* dst, int64_t n)
{
    LM_HIP(hipDeviceSynchronize());
    LM_HIP(hipMemcpyFromSymbol(dst, HIP_SYMBOL(lm_g2_stamp_buf), (size_t)n * 8, 0, hipMemcpyDeviceToHost));
    return LM_OK;
}
#endif

extern "C" int lm_fcn2_forward(LmFcn2* f, const uint8_t* d_rgb, int h, int w, float* d_out, float* d_text, float* d_rec, void* stream)
{
    if (!f || !d_rgb || h < 32 || w < 32 || h > f->max_h || w > f->max_w) {
        lm_set_error("lm_fcn2_forward: bad arguments (frame %dx%d, network sized for %dx%d)", w, h, f ? f->max_w : 0, f ? f->max_h : 0);
        return LM_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    if (h != f->cur_h || w != f->cur_w) {
        // new frame size: plane pitches change and everything outside the new images must read as zero (halos, tile overhang)
        lm_f2_geometry(f, h, w);
        LM_HIP(hipMemsetAsync(f->arena, 0, (size_t)f->arena_bytes, st));
        for (auto& l : f->layer) {
            if (!l.set) continue;
            std::vector<long long> psrc((size_t)l.nchunks * l.npc * 2);
            for (size_t i = 0; i < (size_t)l.nchunks * l.npc; i++) {
                const LmF2Tensor& t = f->t[l.planes[2 * i]];
                psrc[2 * i] = t.off + (long long)l.planes[2 * i + 1] * t.plane;
                psrc[2 * i + 1] = psrc[2 * i] + (t.lo ? (long long)t.c8 * t.plane : 0);
            }
            LM_HIP(hipMemcpyAsync(l.d_psrc, psrc.data(), psrc.size() * 8, hipMemcpyHostToDevice, st));
            LM_HIP(hipStreamSynchronize(st));       // psrc is a stack vector
        }
        f->cur_h = h; f->cur_w = w;
    }
    const long long npx = (long long)h * w;
    LmF2Tensor* T = f->t;
    int rc;
    {
        const LmF2Tensor& x0 = T[LM_F2_X0P];
        char* hi = f->arena + x0.off;
        hipLaunchKernelGGL(lm_k_prepare2, dim3((unsigned)std::min<long long>((npx + 255) / 256, 8192)), dim3(256), 0, st, d_rgb, hi, hi + x0.plane, h, w, x0.Wp, x0.halo);
    }
    // ---- encoder: conv_down_block_n -> PRE n (before pooling, the skip connection) + POOL n
    for (int n = 0; n < 5; n++)
        if ((rc = lm_f2_run(f, n, &T[LM_F2_PRE0 + n], &T[LM_F2_POOL0 + n], LM_ACT_GELU, nullptr, 0, 0, st))) return rc;
    if ((rc = lm_f2_run(f, 5, &T[LM_F2_MID], nullptr, LM_ACT_GELU, nullptr, 0, 0, st))) return rc;
    // ---- decoder
    for (int n = 0; n < 5; n++) {
        const LmF2Tensor& up = T[LM_F2_UPT0 + n];
        const LmF2Tensor& in = T[f->layer[6 + n].planes[0]];
        if ((rc = lm_f2_run(f, 6 + n, &up, nullptr, LM_ACT_GELU, nullptr, 0, 0, st))) return rc;
        if (up.H > 2 * in.H || up.W > 2 * in.W) {
            const long long nb = ((long long)(up.H - 2 * in.H) * up.W + (long long)2 * in.H * (up.W - 2 * in.W)) * up.c8;
            char* hi = f->arena + up.off;
            hipLaunchKernelGGL(lm_k_convT_border2, dim3((unsigned)std::min<long long>((nb + 255) / 256, 4096)), dim3(256), 0, st, hi,
                               up.lo ? hi + (long long)up.c8 * up.plane : nullptr, up.plane, up.Wp, up.halo, up.H, up.W, 2 * in.H, 2 * in.W, up.c8,
                               f->layer[6 + n].d_bias, LM_ACT_GELU);
        }
        if ((rc = lm_f2_run(f, 11 + n, n < 4 ? &T[LM_F2_CU0 + n] : &T[LM_F2_XUP], nullptr, LM_ACT_GELU, nullptr, 0, 0, st))) return rc;
    }
    // ---- heads
    const bool fused16 = f->layer[16].epi == LM_G2_EPI_V, fused20 = f->layer[20].epi == LM_G2_EPI_V;
    // the two logit images go straight into the caller's buffers when it gave any
    float* const text = d_text ? d_text : f->text;
    float* const outl = d_out ? d_out : f->outl;
    if ((rc = lm_f2_run(f, 16, nullptr, nullptr, LM_ACT_NONE, f->tbuf, 16, 16, st, fused16 ? 1 : 0, fused16 ? d_rec : nullptr, text))) return rc;
    if (!fused16) {
        const LmF2Tensor &x0 = T[LM_F2_X0P], &dp = T[LM_F2_DP];
        const int tiles = ((w + 31) / 32) * ((h + 15) / 16);
        hipLaunchKernelGGL(lm_k_vsum2_text_rec, dim3(tiles), dim3(256), 0, st, f->tbuf, h, w, f->layer[16].d_bias + 16, f->arena + x0.off,
                           f->arena + x0.off + x0.plane, text, f->rec4, f->arena + dp.off, f->arena + dp.off + dp.plane, dp.Wp, dp.halo);
    }
    if ((rc = lm_f2_run(f, 18, &T[LM_F2_P1], nullptr, LM_ACT_GELU, nullptr, 0, 0, st))) return rc;
    if ((rc = lm_f2_run(f, 19, &T[LM_F2_P2], nullptr, LM_ACT_GELU, nullptr, 0, 0, st))) return rc;
    if ((rc = lm_f2_run(f, 20, nullptr, nullptr, LM_ACT_NONE, f->tbuf, 8, 8, st, fused20 ? 2 : 0, nullptr, nullptr, outl))) return rc;
    if (!fused20) {
        const int tiles = ((w + 31) / 32) * ((h + 15) / 16);
        hipLaunchKernelGGL((lm_k_vsum<7, 1, 8, false>), dim3(tiles), dim3(256), 0, st, f->tbuf, h, w, f->layer[20].d_bias + 16, LM_ACT_NONE, outl, 1,
                           nullptr, nullptr, nullptr);
    }
    if (d_rec && !fused16) hipLaunchKernelGGL(lm_k_nhwc4_to_chw3, dim3((unsigned)std::min<long long>((npx + 255) / 256, 8192)), dim3(256), 0, st, f->rec4, d_rec, npx);
    LM_HIP(hipGetLastError());
    return LM_OK;
}
