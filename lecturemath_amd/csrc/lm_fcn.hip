// lm_fcn.hip -- FCN-LectureNet inference on gfx950: hand-written MFMA convolution stack.
//
// Replaces the torch modules behind FCN_LectureNet.binarize (AccessMath/lecturenet_v1/FCN_lecturenet.py,
// relative to /root/reference/ACCESS2021_release):
//   prepare_image :607-618            lm_k_prepare (uint8 HWC RGB -> fp32 NHWC in [-1,1], channels padded to 8)
//   encode_decode :260-323            lm_k_conv_mfma (3x3 conv + folded BN + GELU; 2x2/s2 transposed conv as four
//                                     scattered 1x1 convs; skip concats read as a second input, never materialised),
//                                     lm_k_maxpool2, lm_k_convT_border (output_size rows/cols that only see the bias)
//   forward heads :364-403            lm_k_conv_small (Cout <= 4: 7x7 text mask, 3x3 reconstruction + tanh, 7x7 output),
//                                     lm_k_diff ((x0 - rec) * sigmoid(text)), lm_k_conv_mfma (7x7 pixel convs)
// Arithmetic: fp32 in, fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains), BatchNorm folded into
// weights/bias on the host in fp32 (eval mode, eps 1e-5), GELU = exact erf form (nn.GELU default).
//
// Layout: activations NHWC fp32 with a per-buffer pixel stride (so a conv can write into a channel slice of a
// wider buffer: the (diff, features) concatenations of the pixel branch are produced in place).
// Implicit GEMM: M = output pixels (block tile 16 x 16 px, wave = 4 rows x 16 cols = two 32-px MFMA blocks),
// N = output channels (NT blocks of 32 per wave), K = taps x input channels walked in chunks of CK channels.
// Per chunk a block stages in LDS the (16+KH-1) x (16+KW-1) x CK input patch (pixel stride CK+4 floats: the
// 16 B A-fragment reads of 16 neighbouring pixels then fall on distinct banks) and the chunk's weights for ALL
// taps, pre-packed on the host in fragment order [tap][kstep][nblock][lane][4].
#include "lm_common.h"

#include <vector>

// Kernels marked LM_NO_PACKED_F32 are compiled without the packed fp32 VALU instructions (v_pk_fma_f32 ...).  The one-channel
// head kernels below, vectorised by the compiler into v_pk_fma_f32 chains fed straight from ds_read_b128 results, returned
// sporadic wrong values in lanes 48-63 (the low register of the packed pair) whenever workgroups of ANOTHER kernel shared
// their CUs -- two forward passes in flight -- and never alone; the same source built with scalar v_fmac_f32 is bit-stable
// under the same overlap (tools/fcn_overlap_repro.py).
#if LM_HIP_EMULATED || defined(LM_FCN_HEAD_VARIANT)
#define LM_NO_PACKED_F32
#else
#define LM_NO_PACKED_F32 __attribute__((target("no-packed-fp32-ops")))
#endif

#define LM_ACT_NONE 0
#define LM_ACT_GELU 1
#define LM_ACT_TANH 2

// GELU (nn.GELU default, the erf form) = max(x, 0) - |x|/2 * erfc(|x|/sqrt 2), with erfc by Abramowitz & Stegun 7.1.26
// (t * P5(t) * exp(-z^2), t = 1 / (1 + p z); |error| <= 1.5e-7): one v_rcp, one v_exp and nine FMAs, branch-free.  Maximum
// absolute error against the float64 GELU over [-12, 12]: 3.3e-7 -- 0.5 x (1 + erff(x / sqrt 2)) evaluated in fp32 is off by up
// to 4.5e-7 from its own roundings.  The library erff costs ~90 VALU instructions and 14 branches per value: in the epilogues
// below (32-128 values per lane) it was 80 % of the convolution kernels' code and, for the thin layers, most of their time.
LM_DEV float lm_gelu(float x)
{
    const float ax = fabsf(x), z = ax * 0.70710678118654752440f;
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
    return fmaxf(x, 0.0f) - 0.5f * ax * (p * t * e);
}

LM_DEV float lm_act(float v, int act)
{
    if (act == LM_ACT_GELU) return lm_gelu(v);
    if (act == LM_ACT_TANH) return tanhf(v);
    return v;
}

// Epilogue helper: calls body(activation functor) with the activation resolved ONCE per kernel (the per-value run-time switch
// kept tanhf and the GELU side by side in every unrolled store of the epilogues).
// GELU or none: what the MFMA kernels are launched with (tanh belongs to the reconstruction head: lm_k_vsum / lm_k_conv_small).
template <class F>
LM_DEV void lm_with_act(int act, F body)
{
    if (act == LM_ACT_GELU) body([](float v) { return lm_gelu(v); });
    else body([](float v) { return v; });
}

struct LmConvArgs {
    const float* in0; int c0, ps0;      // first input: channels used, pixel stride (floats)
    const float* in1; int c1, ps1;      // optional second input (channel concat), c1 = 0 when absent
    int H, W;                           // input grid
    const float* wpk;                   // packed weights [chunk][tap][kstep][nblock][64][4]
    const float* bias;                  // [nblocks * 32] folded bias
    float* out; int ops, ooff;          // output pixel stride (floats) and channel offset
    int Cout, nblocks;                  // real output channels, ceil(Cout / 32)
    int K;                              // kernel side (1, 3, 7), padding (K-1)/2
    int act;
    int tmode, dy, dx, OH, OW;          // transposed mode: input (y, x) -> output (2y+dy, 2x+dx) of an OH x OW grid
    int tg;                             // f16x3 kernel: taps whose weights are staged together (divides K*K)
    int terms;                          // fp16-split kernels: products per operand pair (3, 2 or 1; see lm_k_conv_mfma_h)
    int krows;                          // fp16-split kernels: kernel rows (0 = K: square kernel; 1 = a 1 x K row convolution)
    float* pool; int pool_ps;           // fp16-split kernels: when set, the 2x2/s2 max-pooled output is written as well ([H/2][W/2], this pixel stride)
};

template <int CK, int NT>
__global__ void __launch_bounds__(256) lm_k_conv_mfma(const LmConvArgs a)
{
    LM_DYN_SMEM(smem);
    constexpr int PSTR = CK + 4;            // padded pixel stride in LDS (floats)
    constexpr int KS = CK / 8;              // k-steps of 8 channels per chunk
    const int K = a.K, pad = (K - 1) >> 1, taps = K * K;
    const int PW = 16 + K - 1, PH = 16 + K - 1;
    float* s_patch = (float*)smem;
    float* s_w = s_patch + PH * PW * PSTR;               // [tap][ks][NT][64][4]
    const int tiles_x = (a.W + 15) >> 4;
    const int ty0 = (blockIdx.x / tiles_x) << 4, tx0 = (blockIdx.x % tiles_x) << 4;
    const int nb0 = blockIdx.y * NT;
    const int lane = lm_lane(), wave = (int)(threadIdx.x >> 6), half = lane >> 5;
    const int nchunks = (a.c0 + a.c1) / CK;

    lm_f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.0f;

    // this lane's pixel inside the two 32-px MFMA blocks of the wave: rows wave*4 + m*2 + (i>>4), col i & 15
    const int pi = lane & 31;
    const int prow = wave * 4 + (pi >> 4), pcol = pi & 15;

    for (int ch = 0; ch < nchunks; ch++) {
        const bool first = ch * CK < a.c0;
        const float* src = first ? a.in0 : a.in1;
        const int ps = first ? a.ps0 : a.ps1;
        const int coff = first ? ch * CK : ch * CK - a.c0;
        __syncthreads();
        // ---- stage the input patch (zero outside the image)
        for (int i = threadIdx.x; i < PH * PW * (CK / 4); i += blockDim.x) {
            const int px = i / (CK / 4), q = i - px * (CK / 4);
            const int py = px / PW, pxx = px - py * PW;
            const int y = ty0 + py - pad, x = tx0 + pxx - pad;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y >= 0 && y < a.H && x >= 0 && x < a.W) v = *(const float4*)(src + ((long long)y * a.W + x) * ps + coff + q * 4);
            *(float4*)(s_patch + px * PSTR + q * 4) = v;
        }
        // ---- stage the chunk's weights for every tap (the block's NT n-blocks)
        {
            const float* wsrc = a.wpk + (long long)ch * taps * KS * a.nblocks * 256;
            const int n4 = taps * KS * NT * 64;          // float4 count
            for (int i = threadIdx.x; i < n4; i += blockDim.x) {
                const int tk = i / (NT * 64), rem = i - tk * (NT * 64);      // tk = tap * KS + ks
                const int nb = rem >> 6, l = rem & 63;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (nb0 + nb < a.nblocks) v = *(const float4*)(wsrc + ((long long)(tk * a.nblocks + nb0 + nb) * 64 + l) * 4);
                *(float4*)(s_w + (long long)i * 4) = v;
            }
        }
        __syncthreads();
        // ---- MFMA over taps x k-steps
        for (int t = 0; t < taps; t++) {
            const int kh = t / K, kw = t - kh * K;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                float4 af[2], bf[NT];
#pragma unroll
                for (int m = 0; m < 2; m++)
                    af[m] = *(const float4*)(s_patch + ((prow + m * 2 + kh) * PW + pcol + kw) * PSTR + ks * 8 + half * 4);
#pragma unroll
                for (int n = 0; n < NT; n++) bf[n] = *(const float4*)(s_w + ((long long)((t * KS + ks) * NT + n) * 64 + lane) * 4);
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int n = 0; n < NT; n++) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].x, bf[n].x, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].y, bf[n].y, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].z, bf[n].z, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].w, bf[n].w, acc[m][n], 0, 0, 0);
                    }
            }
        }
    }
    // ---- epilogue: D[row = (r&3) + 8*(r>>2) + 4*half][col = lane & 31]; row = pixel in the 32-px block, col = channel
    const int cj = lane & 31;
#pragma unroll
    for (int n = 0; n < NT; n++) {
        const int co = (nb0 + n) * 32 + cj;
        if (nb0 + n >= a.nblocks || co >= a.Cout) continue;
        const float b = a.bias[co];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int y = ty0 + wave * 4 + m * 2 + (i >> 4), x = tx0 + (i & 15);
                if (y >= a.H || x >= a.W) continue;
                const float v = lm_act(acc[m][n][r] + b, a.act);
                long long opix = a.tmode ? ((long long)(2 * y + a.dy) * a.OW + (2 * x + a.dx)) : ((long long)y * a.W + x);
                a.out[opix * a.ops + a.ooff + co] = v;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// fp16-split variant ("f16x3"): every fp32 operand is split as x = hi + lo with hi = (f16)x, lo = (f16)(x - hi) (about 22
// significant bits together; products of two f16 are exact in fp32), and a.b is accumulated in fp32 as
// hi_a.hi_b + hi_a.lo_b + lo_a.hi_b on v_mfma_f32_32x32x16_f16: 3 MFMAs of 32 cycles per 16 channels instead of 8 MFMAs of
// 64 cycles -- 5.3x the fp32 MFMA rate at ~1e-6 relative error (the dropped lo.lo term is ~2^-22).
// Same tiling as lm_k_conv_mfma; chunks are 16 logical channels; activations stay fp32 in HBM and are split while the
// patch is staged (pixel stride in LDS: 16 hi + 16 lo halfs + 16 B pad = 80 B, conflict-free 16-B fragment reads);
// weights are split and packed on the host: [chunk][tap][nblock][hi|lo][lane][8 halfs],
// element j of lane l = W[co = nblock*32 + (l & 31)][ci = chunk*16 + 8*(l >> 5) + j].
// ------------------------------------------------------------------------------------------------
typedef _Float16 lm_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 lm_h4 __attribute__((ext_vector_type(4)));

#if LM_HIP_EMULATED
lm_f32x16 hipemu_mfma_32x32x16f16(lm_h8 a, lm_h8 b, lm_f32x16 c);
#define LM_MFMA_F16(a, b, c) hipemu_mfma_32x32x16f16(a, b, c)
#else
#define LM_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif

// Staging is software-pipelined: the next group of weights and the next chunk's input patch are fetched into registers
// while the MFMAs of the current group run (weights double-buffered in LDS, one LDS-only barrier per group -- a full
// __syncthreads() would drain the prefetches).
#define LM_CV_MAXP 8        // float4 patch items per thread, worst case (22 x 22 px x 4 / 256 for 7x7)
#define LM_CV_MAXW 8        // 16-B weight items per thread and group, worst case: taps-per-group x NT <= 16
// The prefetch arrays are sized per instantiation (kernel side and NT are template parameters): with worst-case arrays every
// variant needed ~200 VGPRs and the NT = 2 / transposed variants fell to one wave per SIMD.

template <int MAXP>
LM_DEV void lm_cv_load_patch(const LmConvArgs& a, int ch, int ty0, int tx0, int pady, int pad, int PW, int items, int ctot, float4 (&pr)[MAXP])
{
#pragma unroll
    for (int k = 0; k < MAXP; k++) {
        const int i = (int)threadIdx.x + k * 256;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < items) {
            const int px = i >> 2, q = i & 3;
            const int py = px / PW, pxx = px - py * PW;
            const int y = ty0 + py - pady, x = tx0 + pxx - pad;
            const int cl = ch * 16 + q * 4;                     // logical (concatenated) channel
            if (y >= 0 && y < a.H && x >= 0 && x < a.W && cl < ctot) {
                const long long p = (long long)y * a.W + x;
                v = (cl < a.c0) ? *(const float4*)(a.in0 + p * a.ps0 + cl) : *(const float4*)(a.in1 + p * a.ps1 + (cl - a.c0));
            }
        }
        pr[k] = v;
    }
}

// LDS patch layout: pixel stride 80 B (16 hi + 16 lo halfs + 16 B pad), row stride rounded up to 256 B.  ds_read_b128 is
// serviced in 16-lane groups that mix lanes of two patch rows ({0-3,12-15,20-27}, ...): with the row stride a multiple of
// the 256-B bank row the 16 lanes of a group fall on pixels 0..15 of one residue class, i.e. on 16 distinct 16-B slots
// (5 slots per pixel, 5 coprime to 16).  Measured before the padding: 37-39 % of the LDS cycles were bank conflicts.
LM_DEV constexpr int lm_cv_row_bytes(int PW) { return (PW * 80 + 255) & ~255; }

template <int MAXP>
LM_DEV void lm_cv_store_patch(char* s_patch, int PW, int items, const float4 (&pr)[MAXP])
{
    const int RB = lm_cv_row_bytes(PW);
#pragma unroll
    for (int k = 0; k < MAXP; k++) {
        const int i = (int)threadIdx.x + k * 256;
        if (i < items) {
            const int px = i >> 2, q = i & 3;
            const int py = px / PW, pxx = px - py * PW;
            char* dst = s_patch + py * RB + pxx * 80;
            const float4 v = pr[k];
            lm_h4 hi, lo;
            hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
            lo[0] = (_Float16)(v.x - (float)hi[0]); lo[1] = (_Float16)(v.y - (float)hi[1]);
            lo[2] = (_Float16)(v.z - (float)hi[2]); lo[3] = (_Float16)(v.w - (float)hi[3]);
            *(lm_h4*)(dst + q * 8) = hi;
            *(lm_h4*)(dst + 32 + q * 8) = lo;
        }
    }
}

template <int NT, int MAXW>
LM_DEV void lm_cv_load_w(const LmConvArgs& a, int ch, int t0, int ntg, int taps, int nb0, uint4 (&wr)[MAXW])
{
    const uint4* wsrc = (const uint4*)a.wpk + (long long)ch * taps * a.nblocks * 128;
    const int n16 = ntg * NT * 128;
#pragma unroll
    for (int k = 0; k < MAXW; k++) {
        const int i = (int)threadIdx.x + k * 256;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i < n16) {
            const int t = i / (NT * 128), rem = i - t * (NT * 128);
            const int nb = rem >> 7, u = rem & 127;
            if (nb0 + nb < a.nblocks) v = wsrc[(long long)((t0 + t) * a.nblocks + nb0 + nb) * 128 + u];
        }
        wr[k] = v;
    }
}

template <int MAXW>
LM_DEV void lm_cv_store_w(char* s_w, int n16, const uint4 (&wr)[MAXW])
{
#pragma unroll
    for (int k = 0; k < MAXW; k++) {
        const int i = (int)threadIdx.x + k * 256;
        if (i < n16) *(uint4*)(s_w + (long long)i * 16) = wr[k];
    }
}

// The MFMAs of NTAPS taps with every LDS address an immediate offset from two per-lane bases (pa: the lane's pixel in the patch row
// of the group's first tap; wl: the lane's slot in the weight buffer).  Tap tt of the group sits KW taps per kernel row:
// (kh, kw) = (tt / KW, tt % KW) relative to the group's first tap, which must be the first tap of a kernel row.
// Fully unrolled: the compiler is free to issue the fragment loads of the next taps under the MFMAs of the current one (the
// run-time tap loop recomputed ~28 VALU instructions of addresses per tap and could not look across iterations).
template <int NT, int TERMS, int NTAPS, int KW, int RB, bool SWAP>
LM_DEV void lm_cv_taps(const char* pa, const char* wl, lm_f32x16 (&acc)[2][NT])
{
#pragma unroll
    for (int tt = 0; tt < NTAPS; tt++) {
        const int kh = tt / KW, kw = tt - kh * KW;
        lm_h8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const char* pp = pa + (m * 2 + kh) * RB + kw * 80;
            ah[m] = *(const lm_h8*)pp;
            if (TERMS >= 2) al[m] = *(const lm_h8*)(pp + 32);
        }
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const char* wp = wl + (tt * NT + n) * 2048;
            bh[n] = *(const lm_h8*)wp;
            if (TERMS >= 3) bl[n] = *(const lm_h8*)(wp + 1024);
        }
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < NT; n++) acc[m][n] = SWAP ? LM_MFMA_F16(bh[n], ah[m], acc[m][n]) : LM_MFMA_F16(ah[m], bh[n], acc[m][n]);
        if (TERMS >= 3) {
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < NT; n++) acc[m][n] = SWAP ? LM_MFMA_F16(bl[n], ah[m], acc[m][n]) : LM_MFMA_F16(ah[m], bl[n], acc[m][n]);
        }
        if (TERMS >= 2) {
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int n = 0; n < NT; n++) acc[m][n] = SWAP ? LM_MFMA_F16(bh[n], al[m], acc[m][n]) : LM_MFMA_F16(al[m], bh[n], acc[m][n]);
        }
    }
}

// taps per weight group.  Chosen so that the block's LDS (patch + two weight buffers of tg * NT * 2 KB) lets at least two
// workgroups share a CU (three for 3x3 / NT = 1): 3x3 -> a kernel row, 7x7 -> a kernel row (NT = 1) or 4 taps (NT = 2);
// never more than LM_CV_MAXW 16-B items per thread.
constexpr int lm_cv_tg(int K, int NT)
{
    const int cap = (LM_CV_MAXW * 2) / NT;
    int want = K * K;
    if (K == 3) want = 3;       // a kernel row per group (lm_cv_taps wants groups that start a kernel row)
    else if (K >= 5) want = (NT == 1) ? K : (NT == 2 ? 4 : 2);
    return want < cap ? want : cap;
}

// KS: kernel side known at compile time (1, 3, 7), or 0 = taken from the arguments (worst-case prefetch arrays)
// TERMS: products per pair of operands.  3 = hi.hi + hi.lo + lo.hi (the "f16x3" format above, ~22 bits per operand);
// 2 = hi.hi + lo.hi (activations split, weights rounded to f16); 1 = hi.hi (both operands rounded to f16).
// KH: kernel rows known at compile time (1: a 1 x KS row convolution, see lm_rowconv_layer), 0 = KS (square kernel)
// SWAP: the weights are the MFMA's A operand and the pixels its B operand (see the second epilogue below): layers with <= 16 outputs
template <int NT, int KS, int TG = 0, int TERMS = 3, int KH = 0, bool SWAP = false>     // TG: taps per weight group, 0 = lm_cv_tg(KS, NT)
__global__ void __launch_bounds__(256, (NT <= 2) ? 2 : 1) lm_k_conv_mfma_h(const LmConvArgs a)      // two waves per SIMD whenever the accumulators allow
{
    LM_DYN_SMEM(smem);
    static_assert(KH == 0 || KS != 0, "a fixed row count needs a fixed kernel side");
    constexpr int PB = 80;                  // bytes per pixel in LDS
    constexpr int KHC = KH ? KH : KS;
    constexpr int MAXP = KS ? ((16 + KHC - 1) * (16 + KS - 1) * 4 + 255) / 256 : LM_CV_MAXP;
    constexpr int MAXW = KS ? ((TG ? TG : lm_cv_tg(KS, NT)) * NT + 1) / 2 : LM_CV_MAXW;
    const int K = KS ? KS : a.K, KR = KH ? KH : K, pad = (K - 1) >> 1, pady = (KR - 1) >> 1, taps = KR * K;
    const int PW = 16 + K - 1, PH = 16 + KR - 1;
    const int patch_items = PH * PW * 4;
    char* s_patch = smem;
    const int RB = lm_cv_row_bytes(PW);
    char* s_wbuf = s_patch + PH * RB;                           // 2 x [tap in group][NT][2][64] x 16 B
    const int wbuf_bytes = a.tg * NT * 2048;
    const int tiles_x = (a.W + 15) >> 4;
    const int ty0 = (blockIdx.x / tiles_x) << 4, tx0 = (blockIdx.x % tiles_x) << 4;
    const int nb0 = blockIdx.y * NT;
    const int lane = lm_lane(), wave = (int)(threadIdx.x >> 6), half = lane >> 5;
    const int ctot = a.c0 + a.c1;
    const int nchunks = (ctot + 15) >> 4;
    const int ngroups = (taps + a.tg - 1) / a.tg;

    lm_f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.0f;
    const int pi = lane & 31;
    const int prow = wave * 4 + (pi >> 4), pcol = pi & 15;

    float4 pr[MAXP];
    uint4 wr[MAXW];
    lm_cv_load_patch<MAXP>(a, 0, ty0, tx0, pady, pad, PW, patch_items, ctot, pr);
    lm_cv_load_w<NT, MAXW>(a, 0, 0, a.tg < taps ? a.tg : taps, taps, nb0, wr);
    int buf = 0;
    for (int ch = 0; ch < nchunks; ch++) {
        lm_lds_barrier();                                       // everybody is done with the previous chunk's patch
        lm_cv_store_patch<MAXP>(s_patch, PW, patch_items, pr);
        if (ch + 1 < nchunks) lm_cv_load_patch<MAXP>(a, ch + 1, ty0, tx0, pady, pad, PW, patch_items, ctot, pr);   // in flight for a whole chunk
        for (int g = 0; g < ngroups; g++) {
            const int t0 = g * a.tg;
            const int ntg = (taps - t0 < a.tg) ? taps - t0 : a.tg;
            char* s_w = s_wbuf + buf * wbuf_bytes;
            lm_cv_store_w<MAXW>(s_w, ntg * NT * 128, wr);
            lm_lds_barrier();       // group g (and, for g == 0, the patch) visible; buffer `buf` was last read two groups ago
            if (g + 1 < ngroups) {
                const int t1 = t0 + a.tg;
                lm_cv_load_w<NT, MAXW>(a, ch, t1, (taps - t1 < a.tg) ? taps - t1 : a.tg, taps, nb0, wr);
            } else if (ch + 1 < nchunks) {
                lm_cv_load_w<NT, MAXW>(a, ch + 1, 0, a.tg < taps ? a.tg : taps, taps, nb0, wr);
            }
            // groups of whole kernel rows (all instantiations with a compile-time side except 7x7 / NT = 2): unrolled taps
            constexpr int TGC = KS ? (TG ? TG : lm_cv_tg(KS, NT)) : 0;
            constexpr bool ROWS = KS && (TGC % (KS ? KS : 1) == 0);
            static_assert(!SWAP || ROWS, "the swapped operand order is built for the unrolled row groups only");
            if (ROWS) {
                constexpr int RBC = lm_cv_row_bytes(16 + (KS ? KS : 1) - 1);
                lm_cv_taps<NT, TERMS, (ROWS ? TGC : 1), (KS ? KS : 1), RBC, SWAP>(s_patch + (prow + t0 / (KS ? KS : 1)) * RBC + pcol * PB + half * 16,
                                                                              s_w + lane * 16, acc);
            } else
            for (int tt = 0; tt < ntg; tt++) {
                const int t = t0 + tt;
                const int kh = t / K, kw = t - kh * K;
                lm_h8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    const char* pp = s_patch + (prow + m * 2 + kh) * RB + (pcol + kw) * PB + half * 16;
                    ah[m] = *(const lm_h8*)pp;
                    if (TERMS >= 2) al[m] = *(const lm_h8*)(pp + 32);
                }
#pragma unroll
                for (int n = 0; n < NT; n++) {
                    const char* wp = s_w + ((long long)(tt * NT + n) * 128 + lane) * 16;
                    bh[n] = *(const lm_h8*)wp;
                    if (TERMS >= 3) bl[n] = *(const lm_h8*)(wp + 64 * 16);
                }
                // the three products of one accumulator are a dependent chain: issue them term by term across the 2 * NT
                // accumulators so that neighbouring MFMAs are independent
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int n = 0; n < NT; n++) acc[m][n] = SWAP ? LM_MFMA_F16(bh[n], ah[m], acc[m][n]) : LM_MFMA_F16(ah[m], bh[n], acc[m][n]);
                if (TERMS >= 3) {
#pragma unroll
                    for (int m = 0; m < 2; m++)
#pragma unroll
                        for (int n = 0; n < NT; n++) acc[m][n] = SWAP ? LM_MFMA_F16(bl[n], ah[m], acc[m][n]) : LM_MFMA_F16(ah[m], bl[n], acc[m][n]);
                }
                if (TERMS >= 2) {
#pragma unroll
                    for (int m = 0; m < 2; m++)
#pragma unroll
                        for (int n = 0; n < NT; n++) acc[m][n] = SWAP ? LM_MFMA_F16(bh[n], al[m], acc[m][n]) : LM_MFMA_F16(al[m], bh[n], acc[m][n]);
                }
            }
            buf ^= 1;
        }
    }
    if constexpr (!SWAP) {
    // Epilogue.  D[row i = (r & 3) + 8 * (r >> 2) + 4 * half][col = lane & 31]: row = pixel of the 32-px block (image row i >> 4,
    // column i & 15), col = channel.  Everything of a store address except the lane's own part (its half and its channel) is the same
    // for the whole wave, so it is kept in scalar registers: per value the VALU does the bias, the activation and nothing else (with
    // per-value 64-bit index arithmetic and bounds checks the epilogue was ~45 vector instructions per value -- half of the first
    // layer's wave time was instruction issue).
    const int cj = lane & 31;
    const int y_w = ty0 + LM_UNIFORM(wave) * 4;                     // first of the wave's four image rows
    const bool full = (ty0 + 16 <= a.H) && (tx0 + 16 <= a.W);       // interior tile: no bounds checks
    const int rs = a.W * a.ops;                                     // floats per image row
    float* const wbase = a.out + ((long long)y_w * a.W + tx0) * a.ops + a.ooff;
    const int OH = a.H >> 1, OW = a.W >> 1;
    float* const pbase = a.pool ? a.pool + ((long long)(y_w >> 1) * OW + (tx0 >> 1)) * a.pool_ps : nullptr;
    // Stores.  A lane holds ONE channel of 16 pixels, so written straight from the registers a pixel's 128 bytes come from 32 lanes
    // as dword stores -- measured, those run at ~2.4 TB/s over the chip (64 cycles of a CU's memory pipeline per wave instruction),
    // less than half of what 16-byte stores reach.  So each wave turns its 32 px x 32 ch block around in LDS (its own 4.5 KB of the
    // patch area, rows padded to 36 floats) and writes it as four global_store_dwordx4, every instruction 8 whole pixels (1 KB
    // contiguous when the buffer is exactly 32 channels wide).  Same wave, in-order LDS: no barrier between its writes and reads.
    const bool tvec = (((a.ops | a.ooff) & 3) == 0) && ((((uintptr_t)a.out) & 15) == 0);
    float* const s_tr = (float*)smem + LM_UNIFORM(wave) * (32 * 36);
    if (tvec) lm_lds_barrier();         // everybody is done with the patch and the weights
    lm_with_act(a.act, [&](auto actf) {
#pragma unroll
    for (int n = 0; n < NT; n++) {
        if (nb0 + n >= a.nblocks) continue;
        const int co = (nb0 + n) * 32 + cj;
        const bool cvalid = co < a.Cout;
        const float b = cvalid ? a.bias[co] : 0.0f;
        const int loff = 4 * half * a.ops + co;                     // the lane's part of the address (floats)
#pragma unroll
        for (int m = 0; m < 2; m++) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = actf(acc[m][n][r] + b);
            if (tvec) {
#pragma unroll
                for (int r = 0; r < 16; r++) s_tr[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + cj] = v[r];
                LM_WAVE_SYNC();
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int p = q * 8 + (lane >> 3), ch4 = (lane & 7) * 4;
                    const float4 t = *(const float4*)(s_tr + p * 36 + ch4);
                    const int row = m * 2 + (p >> 4), col = p & 15;
                    if ((nb0 + n) * 32 + ch4 < a.Cout && (full || (y_w + row < a.H && tx0 + col < a.W)))
                        *(float4*)(wbase + (row * rs + col * a.ops) + (nb0 + n) * 32 + ch4) = t;
                }
                LM_WAVE_SYNC();         // the block has been read before the next one is written
            } else if (cvalid) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m * 2 + (r >> 3), c0 = (r & 3) + 8 * ((r >> 2) & 1);       // compile-time
                if (full || (y_w + row < a.H && tx0 + c0 + 4 * half < a.W)) (wbase + (row * rs + c0 * a.ops))[loff] = v[r];
            }
            }
            // 2x2 max pooling from registers: accumulator rows r, r + 1 are neighbouring columns of the block's first image row,
            // r + 8, r + 9 the same columns of its second (the block's two rows start at an even y, tiles at an even x)
            if (pbase && cvalid) {
                const int ploff = 2 * half * a.pool_ps + co;
#pragma unroll
                for (int r = 0; r < 8; r += 2) {
                    const int pc0 = ((r & 3) + 8 * ((r >> 2) & 1)) >> 1;
                    if (full || ((y_w >> 1) + m < OH && (tx0 >> 1) + pc0 + 2 * half < OW))
                        (pbase + ((long long)m * OW + pc0) * a.pool_ps)[ploff] = fmaxf(fmaxf(v[r], v[r + 1]), fmaxf(v[r + 8], v[r + 9]));
                }
            }
        }
    }
    });
    } else {
    // Epilogue (SWAP).  The weights are the MFMA's A operand and the pixels its B operand, so D[row = channel][col = pixel]: a lane owns ONE
    // pixel of each 32-px block (lane & 31: image row >> 4, column & 15) and, in registers 4k .. 4k + 3, the four CONSECUTIVE channels
    // 8k + 4 * half + (0..3) -- 16 contiguous bytes of the NHWC output, one global_store_dwordx4.  Measured per layer against the
    // other operand order (a lane = one channel of 16 pixels, dword stores that cover whole 128-B lines per instruction): faster where
    // a pixel has <= 16 outputs (half of those dword stores' lanes are idle: pixel layer 2 755 -> 671 us, head rows 152 -> 125), slower
    // for 32+ outputs (32-B pieces of 32 different lines per instruction) -- so only those layers are launched this way.
    const int y_w = ty0 + LM_UNIFORM(wave) * 4;                     // first of the wave's four image rows
    const bool full = (ty0 + 16 <= a.H) && (tx0 + 16 <= a.W);       // interior tile: no bounds checks
    const int prow_l = pi >> 4, pcol_l = pi & 15;                    // the lane's pixel inside a block
    const bool vec = (((a.ops | a.ooff) & 3) == 0) && ((((uintptr_t)a.out) & 15) == 0);
    const int cout4 = (a.Cout + 3) & ~3;
    float* const wbase = a.out + ((long long)y_w * a.W + tx0) * a.ops + a.ooff;       // wave-uniform
    const int loff = (prow_l * a.W + pcol_l) * a.ops + 4 * half;                       // the lane's part (floats)
    lm_with_act(a.act, [&](auto actf) {
#pragma unroll
    for (int n = 0; n < NT; n++) {
        if (nb0 + n >= a.nblocks) continue;
        const int cb = (nb0 + n) * 32 + 4 * half;                   // the lane's first channel of group k = 0
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const bool inb = full || (y_w + m * 2 + prow_l < a.H && tx0 + pcol_l < a.W);
            float* const prow_p = wbase + (m * 2) * a.W * a.ops + loff;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int ch = cb + 8 * k;
                if (ch >= cout4) continue;
                const float4 b4 = *(const float4*)(a.bias + ch);
                float4 v;
                v.x = actf(acc[m][n][4 * k + 0] + b4.x); v.y = actf(acc[m][n][4 * k + 1] + b4.y);
                v.z = actf(acc[m][n][4 * k + 2] + b4.z); v.w = actf(acc[m][n][4 * k + 3] + b4.w);
                if (inb) {
                    float* dst = prow_p + (nb0 + n) * 32 + 8 * k;
                    if (vec) *(float4*)dst = v;
                    else { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
                }
            }
        }
    }
    });
    }
}

// 2x2 stride-2 transposed convolution (fp16-split operands): the four (dy, dx) sub-convolutions are 1x1 GEMMs over the SAME
// input pixels, so one workgroup stages a 16x16-px patch chunk once, reads its A fragments once and feeds four accumulator
// sets (weights packed like a conv with 4 taps, tap = dy * 2 + dx).  One n-block (32 output channels) per workgroup.
template <int TERMS>
__global__ void __launch_bounds__(256, 2) lm_k_convT_mfma_h(const LmConvArgs a)
{
    LM_DYN_SMEM(smem);
    constexpr int PW = 16, TAPS = 4;
    const int RB = lm_cv_row_bytes(PW);
    char* s_patch = smem;
    char* s_wbuf = s_patch + PW * RB;                           // 2 x [tap][2][64] x 16 B
    const int tiles_x = (a.W + 15) >> 4;
    const int ty0 = (blockIdx.x / tiles_x) << 4, tx0 = (blockIdx.x % tiles_x) << 4;
    const int nb0 = blockIdx.y;
    const int lane = lm_lane(), wave = (int)(threadIdx.x >> 6), half = lane >> 5;
    const int ctot = a.c0 + a.c1;
    const int nchunks = (ctot + 15) >> 4;
    lm_f32x16 acc[TAPS][2];
#pragma unroll
    for (int t = 0; t < TAPS; t++)
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][m][r] = 0.0f;
    const int pi = lane & 31;
    const int prow = wave * 4 + (pi >> 4), pcol = pi & 15;
    float4 pr[4];           // 16 x 16 px x 4 items / 256 threads
    uint4 wr[2];            // 4 taps x 128 items / 256 threads
    lm_cv_load_patch<4>(a, 0, ty0, tx0, 0, 0, PW, PW * PW * 4, ctot, pr);
    lm_cv_load_w<1, 2>(a, 0, 0, TAPS, TAPS, nb0, wr);
    int buf = 0;
    for (int ch = 0; ch < nchunks; ch++) {
        char* s_w = s_wbuf + buf * (TAPS * 2048);
        lm_lds_barrier();                                       // the previous chunk's patch has been read by everybody
        lm_cv_store_patch<4>(s_patch, PW, PW * PW * 4, pr);
        lm_cv_store_w<2>(s_w, TAPS * 128, wr);
        lm_lds_barrier();
        if (ch + 1 < nchunks) {
            lm_cv_load_patch<4>(a, ch + 1, ty0, tx0, 0, 0, PW, PW * PW * 4, ctot, pr);
            lm_cv_load_w<1, 2>(a, ch + 1, 0, TAPS, TAPS, nb0, wr);
        }
        lm_h8 ah[2], al[2];
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const char* pp = s_patch + (prow + m * 2) * RB + pcol * 80 + half * 16;
            ah[m] = *(const lm_h8*)pp;
            al[m] = *(const lm_h8*)(pp + 32);
        }
#pragma unroll
        for (int t = 0; t < TAPS; t++) {
            const char* wp = s_w + ((long long)t * 128 + lane) * 16;
            const lm_h8 bh = *(const lm_h8*)wp, bl = *(const lm_h8*)(wp + 64 * 16);
#pragma unroll
            for (int m = 0; m < 2; m++) acc[t][m] = LM_MFMA_F16(ah[m], bh, acc[t][m]);
            if (TERMS >= 3) {
#pragma unroll
                for (int m = 0; m < 2; m++) acc[t][m] = LM_MFMA_F16(ah[m], bl, acc[t][m]);
            }
            if (TERMS >= 2) {
#pragma unroll
                for (int m = 0; m < 2; m++) acc[t][m] = LM_MFMA_F16(al[m], bh, acc[t][m]);
            }
        }
        buf ^= 1;
    }
    const int co = nb0 * 32 + (lane & 31);
    if (co >= a.Cout) return;
    const float b = a.bias[co];
    // input pixel (y, x) -> output pixels (2y + dy, 2x + dx); address split as in lm_k_conv_mfma_h: uniform part in scalar registers
    const int y_w = ty0 + LM_UNIFORM(wave) * 4;
    const bool full = (ty0 + 16 <= a.H) && (tx0 + 16 <= a.W);
    const int rs = a.OW * a.ops;
    float* const wbase = a.out + ((long long)(2 * y_w) * a.OW + 2 * tx0) * a.ops + a.ooff;
    const int loff = 8 * half * a.ops + co;
    lm_with_act(a.act, [&](auto actf) {
#pragma unroll
    for (int t = 0; t < TAPS; t++)
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m * 2 + (r >> 3), c0 = (r & 3) + 8 * ((r >> 2) & 1);
                if (full || (y_w + row < a.H && tx0 + c0 + 4 * half < a.W))
                    (wbase + ((2 * row + (t >> 1)) * rs + (2 * c0 + (t & 1)) * a.ops))[loff] = actf(acc[t][m][r] + b);
            }
    });
}

// rows / columns of a transposed-conv output that no input pixel reaches (output_size = 2*in + 1): act(bias)
__global__ void __launch_bounds__(256) lm_k_convT_border(float* out, int ops, int ooff, int OH, int OW, int H2, int W2, int Cout,
                                                         const float* __restrict__ bias, int act)
{
    // pixels with y >= H2 or x >= W2
    const long long nb_px = (long long)(OH - H2) * OW + (long long)H2 * (OW - W2);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nb_px * Cout; i += (long long)gridDim.x * blockDim.x) {
        const long long p = i / Cout;
        const int c = (int)(i - p * Cout);
        int y, x;
        if (p < (long long)(OH - H2) * OW) { y = H2 + (int)(p / OW); x = (int)(p % OW); }
        else { const long long q = p - (long long)(OH - H2) * OW; y = (int)(q / (OW - W2)); x = W2 + (int)(q % (OW - W2)); }
        out[((long long)y * OW + x) * ops + ooff + c] = lm_act(bias[c], act);
    }
}

__global__ void __launch_bounds__(256) lm_k_maxpool2(const float* __restrict__ in, int ips, float* __restrict__ out, int ops,
                                                     int H, int W, int C)
{
    const int OH = H >> 1, OW = W >> 1, C4 = C >> 2;
    const long long total = (long long)OH * OW * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long p = i / C4;
        const int q = (int)(i - p * C4);
        const int oy = (int)(p / OW), ox = (int)(p - (long long)oy * OW);
        const float* s = in + ((long long)(2 * oy) * W + 2 * ox) * ips + q * 4;
        const float4 a = *(const float4*)s, b = *(const float4*)(s + ips), c = *(const float4*)(s + (long long)W * ips),
                     d = *(const float4*)(s + (long long)W * ips + ips);
        float4 r;
        r.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x));
        r.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
        r.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z));
        r.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
        *(float4*)(out + p * ops + q * 4) = r;
    }
}

// uint8 HWC RGB -> fp32 NHWC (x/255 - 0.5)/0.5, channels 3..7 zero (to_tensor + normalize, :607-618)
__global__ void __launch_bounds__(256) lm_k_prepare(const uint8_t* __restrict__ rgb, float* __restrict__ out, long long npx)
{
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += (long long)gridDim.x * blockDim.x) {
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; c++) v[c] = ((float)rgb[p * 3 + c] / 255.0f - 0.5f) / 0.5f;
        *(float4*)(out + p * 8) = make_float4(v[0], v[1], v[2], 0.f);
        *(float4*)(out + p * 8 + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// Direct convolution for Cout <= 4 (heads): 16 x TW px tile; the input patch is staged in LDS 8 channels at a time (pixel
// stride 12 floats -> conflict-free 16-B reads) so that several blocks fit a CU and hide each other's LDS latency.
// COUT == 1 (text mask, output logit): weights [chunk][tap][8], every thread computes TWO pixels 16 columns apart so one
// weight read serves 8 FMAs.  COUT == 4 (3-channel reconstruction): weights [chunk][tap][8][4].
template <int COUT>
__global__ void LM_NO_PACKED_F32 __launch_bounds__(256) lm_k_conv_small(const float* __restrict__ in, int ips, int C, int H, int W,
                                                       const float* __restrict__ wts, const float* __restrict__ bias, int K, int Cout,
                                                       int act, float* __restrict__ out, int ops)
{
    LM_DYN_SMEM(smem);
    constexpr int TW = (COUT == 1) ? 32 : 16;
    constexpr int CS = 12;                                  // 8 channels + 4 pad (floats)
    constexpr int WL = (COUT == 1) ? 1 : 4;                 // weight floats per (tap, channel)
    const int pad = (K - 1) >> 1, taps = K * K, PW = TW + K - 1, PH = 16 + K - 1;
    float* s_patch = (float*)smem;
    float* s_w = s_patch + PH * PW * CS;                    // [tap][8][WL]
    const int tiles_x = (W + TW - 1) / TW;
    const int ty0 = (blockIdx.x / tiles_x) << 4, tx0 = (blockIdx.x % tiles_x) * TW;
    const int ly = (int)(threadIdx.x >> 4), lx = (int)(threadIdx.x & 15);
    float acc[(COUT == 1) ? 2 : 4];
#pragma unroll
    for (int i = 0; i < ((COUT == 1) ? 2 : 4); i++) acc[i] = 0.f;
    for (int c0 = 0; c0 < C; c0 += 8) {
        __syncthreads();
        for (int i = threadIdx.x; i < PH * PW * 2; i += blockDim.x) {
            const int px = i >> 1, q = i & 1;
            const int py = px / PW, pxx = px - py * PW;
            const int y = ty0 + py - pad, x = tx0 + pxx - pad;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y >= 0 && y < H && x >= 0 && x < W) v = *(const float4*)(in + ((long long)y * W + x) * ips + c0 + q * 4);
            *(float4*)(s_patch + px * CS + q * 4) = v;
        }
        const float* wsrc = wts + (long long)(c0 >> 3) * taps * 8 * WL;
        for (int i = threadIdx.x; i < taps * 8 * WL / 4; i += blockDim.x) *(float4*)(s_w + i * 4) = *(const float4*)(wsrc + (long long)i * 4);
        __syncthreads();
        for (int t = 0; t < taps; t++) {
            const int kh = t / K, kw = t - kh * K;
            const float* p0 = s_patch + ((ly + kh) * PW + lx + kw) * CS;
            if (COUT == 1) {
                const float* p1 = p0 + 16 * CS;
                const float4 wa = *(const float4*)(s_w + t * 8), wb = *(const float4*)(s_w + t * 8 + 4);
                const float4 u0 = *(const float4*)p0, u1 = *(const float4*)(p0 + 4), v0 = *(const float4*)p1, v1 = *(const float4*)(p1 + 4);
                acc[0] = fmaf(u0.x, wa.x, acc[0]); acc[0] = fmaf(u0.y, wa.y, acc[0]); acc[0] = fmaf(u0.z, wa.z, acc[0]); acc[0] = fmaf(u0.w, wa.w, acc[0]);
                acc[0] = fmaf(u1.x, wb.x, acc[0]); acc[0] = fmaf(u1.y, wb.y, acc[0]); acc[0] = fmaf(u1.z, wb.z, acc[0]); acc[0] = fmaf(u1.w, wb.w, acc[0]);
                acc[1] = fmaf(v0.x, wa.x, acc[1]); acc[1] = fmaf(v0.y, wa.y, acc[1]); acc[1] = fmaf(v0.z, wa.z, acc[1]); acc[1] = fmaf(v0.w, wa.w, acc[1]);
                acc[1] = fmaf(v1.x, wb.x, acc[1]); acc[1] = fmaf(v1.y, wb.y, acc[1]); acc[1] = fmaf(v1.z, wb.z, acc[1]); acc[1] = fmaf(v1.w, wb.w, acc[1]);
            } else {
                const float* ww = s_w + t * 32;
#pragma unroll
                for (int c = 0; c < 8; c += 4) {
                    const float4 v = *(const float4*)(p0 + c);
                    const float4 w0 = *(const float4*)(ww + (c + 0) * 4), w1 = *(const float4*)(ww + (c + 1) * 4),
                                 w2 = *(const float4*)(ww + (c + 2) * 4), w3 = *(const float4*)(ww + (c + 3) * 4);
                    acc[0] = fmaf(v.x, w0.x, acc[0]); acc[1] = fmaf(v.x, w0.y, acc[1]); acc[2] = fmaf(v.x, w0.z, acc[2]); acc[3] = fmaf(v.x, w0.w, acc[3]);
                    acc[0] = fmaf(v.y, w1.x, acc[0]); acc[1] = fmaf(v.y, w1.y, acc[1]); acc[2] = fmaf(v.y, w1.z, acc[2]); acc[3] = fmaf(v.y, w1.w, acc[3]);
                    acc[0] = fmaf(v.z, w2.x, acc[0]); acc[1] = fmaf(v.z, w2.y, acc[1]); acc[2] = fmaf(v.z, w2.z, acc[2]); acc[3] = fmaf(v.z, w2.w, acc[3]);
                    acc[0] = fmaf(v.w, w3.x, acc[0]); acc[1] = fmaf(v.w, w3.y, acc[1]); acc[2] = fmaf(v.w, w3.z, acc[2]); acc[3] = fmaf(v.w, w3.w, acc[3]);
                }
            }
        }
    }
    const int y = ty0 + ly;
    if (y >= H) return;
    if (COUT == 1) {
        const float bb = bias[0];
        if (tx0 + lx < W) out[((long long)y * W + tx0 + lx) * ops] = lm_act(acc[0] + bb, act);
        if (tx0 + lx + 16 < W) out[((long long)y * W + tx0 + lx + 16) * ops] = lm_act(acc[1] + bb, act);
    } else {
        const int x = tx0 + lx;
        if (x < W)
            for (int c = 0; c < Cout; c++) out[((long long)y * W + x) * ops + c] = lm_act(acc[c] + bias[c], act);
    }
}

// Cout == 1 heads (text mask 7x7, output logit 7x7), direct convolution on the VALU with a sliding window in registers:
// 16 x 64 px tile, every thread computes FOUR adjacent pixels of a row, so one kernel row costs 10 patch reads + 7 weight
// reads (16 B each) per 4 channels for 112 FMAs (the two-pixel version needed 2.4x the LDS traffic and was LDS-bound).
// LDS layout per patch row: pixel x sits at position (x % 4) * Q + x / 4 (Q = ceil(PW / 4)), 12 floats per position, so for a
// fixed window offset the 16 lanes of a row read consecutive positions (3 slots apart: conflict-free); row stride is a
// multiple of 256 B because ds_read_b128 lane groups straddle two rows.  Weights [chunk of 8 channels][tap][8].
#define LM_C1_TW 64

LM_DEV int lm_c1_q(int K) { return (LM_C1_TW + K - 1 + 3) >> 2; }
LM_DEV int lm_c1_row_floats(int K) { return (4 * lm_c1_q(K) * 12 + 63) & ~63; }

// KS: kernel side known at compile time (7: the shipped configuration), or 0 = taken from the argument.  With the side known
// the ten window reads of a kernel row are unconditional and issued back to back; with a run-time side every read beyond the
// third is guarded and waits for its own result.
template <int KS>
__global__ void LM_NO_PACKED_F32 __launch_bounds__(256) lm_k_conv_c1(const float* __restrict__ in, int ips, int C, int H, int W,
                                                                     const float* __restrict__ wts, const float* __restrict__ bias, int Karg, int act,
                                                                     float* __restrict__ out, int ops)
{
    LM_DYN_SMEM(smem);
    const int K = KS ? KS : Karg;
    const int pad = (K - 1) >> 1, taps = K * K, PW = LM_C1_TW + K - 1, PH = 16 + K - 1;
    const int Q = lm_c1_q(K), RF = lm_c1_row_floats(K);
    float* s_patch = (float*)smem;                          // [PH][RF]
    const int tiles_x = (W + LM_C1_TW - 1) / LM_C1_TW;
    const int ty0 = (blockIdx.x / tiles_x) << 4, tx0 = (blockIdx.x % tiles_x) * LM_C1_TW;
    const int ly = (int)(threadIdx.x >> 4), lx = (int)(threadIdx.x & 15);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // the next chunk's patch is fetched into registers while the current chunk is being convolved
    constexpr int MAXI = KS ? ((16 + KS - 1) * (LM_C1_TW + KS - 1) * 2 + 255) / 256 : 13;      // 13 = ceil(22 * 70 * 2 / 256) patch items per thread (7x7)
    constexpr int NWIN = KS ? KS + 3 : 10;                  // window pixels 4*lx .. 4*lx + K + 2
    const int items = PH * PW * 2;
    float4 pr[MAXI];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int k = 0; k < MAXI; k++) {
            const int i = (int)threadIdx.x + k * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < items) {
                const int px = i >> 1, q = i & 1;
                const int py = px / PW, pxx = px - py * PW;
                const int y = ty0 + py - pad, x = tx0 + pxx - pad;
                if (y >= 0 && y < H && x >= 0 && x < W) v = *(const float4*)(in + ((long long)y * W + x) * ips + c0 + q * 4);
            }
            pr[k] = v;
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < C; c0 += 8) {
        lm_lds_barrier();                                   // the previous chunk has been read by everybody
#pragma unroll
        for (int k = 0; k < MAXI; k++) {
            const int i = (int)threadIdx.x + k * 256;
            if (i < items) {
                const int px = i >> 1, q = i & 1;
                const int py = px / PW, pxx = px - py * PW;
                *(float4*)(s_patch + py * RF + ((pxx & 3) * Q + (pxx >> 2)) * 12 + q * 4) = pr[k];
            }
        }
        const float* wsrc = wts + (long long)(c0 >> 3) * taps * 8;
        lm_lds_barrier();
        if (c0 + 8 < C) fetch(c0 + 8);
        for (int kh = 0; kh < K; kh++) {
            const float* prow = s_patch + (ly + kh) * RF;
#pragma unroll
            for (int hq = 0; hq < 2; hq++) {                // channels 0-3, 4-7 of the chunk
                float4 win[NWIN];
#pragma unroll
                for (int s2 = 0; s2 < NWIN; s2++)
                    win[s2] = (KS || s2 < K + 3) ? *(const float4*)(prow + ((s2 & 3) * Q + lx + (s2 >> 2)) * 12 + hq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int kw = 0; kw < (KS ? KS : 7); kw++) {
                    if (kw >= K) break;
                    // the weights are the same for every lane: read through the scalar cache into SGPRs (no LDS traffic, no VGPRs)
                    const float4 wv = *(const float4*)(wsrc + (kh * K + kw) * 8 + hq * 4);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float4 u = win[j + kw];
                        acc[j] = fmaf(u.x, wv.x, acc[j]); acc[j] = fmaf(u.y, wv.y, acc[j]);
                        acc[j] = fmaf(u.z, wv.z, acc[j]); acc[j] = fmaf(u.w, wv.w, acc[j]);
                    }
                }
            }
        }
    }
    const int y = ty0 + ly;
    if (y >= H) return;
    const float bb = bias[0];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = tx0 + 4 * lx + j;
        if (x < W) out[((long long)y * W + x) * ops] = lm_act(acc[j] + bb, act);
    }
}

// ------------------------------------------------------------------------------------------------
// Heads with 1 or 3 output channels (text mask 7x7, reconstruction 3x3, output logit 7x7) on the MFMA path: a K x K convolution
// with NV outputs is a 1 x K ROW convolution with K * NV outputs per pixel (output kh * NV + co = the contribution of kernel row kh
// to channel co: lm_k_conv_mfma_h<1, K, 0, TERMS, 1>, weights packed that way by the host) followed by the vertical sum
//     out[y][x][co] = act(bias[co] + sum_kh T[y + kh - pad][x][kh * NV + co])
// below.  On the VALU (lm_k_conv_c1, lm_k_conv_small) the two 7x7 heads took 1.26 of a frame's 6.9 ms.
// T: [pixel][TS] floats.  DIFF (reconstruction head): besides rec the kernel writes diff = (x0 - rec) * sigmoid(text) (:379)
// as one [pixel][4] buffer, which the pixel branch reads as the first of two concatenated inputs (written into channels 0..2 of
// three wider buffers, 12 bytes at a stride of 96-160, the same values took 190-310 us per frame instead of ~15).
// ------------------------------------------------------------------------------------------------
template <int KR, int NV, int TS, bool DIFF>
__global__ void __launch_bounds__(256) lm_k_vsum(const float* __restrict__ T, int H, int W, const float* __restrict__ bias, int act,
                                                 float* __restrict__ out, int ops, const float* __restrict__ x0, const float* __restrict__ text,
                                                 float* __restrict__ diff4)
{
    constexpr int TW = 32, TH = 16, PR = TH + KR - 1, PADR = (KR - 1) / 2, RQ = TW * TS / 4;
    __shared__ float s_t[PR * TW * TS];
    const int tiles_x = (W + TW - 1) / TW;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    for (int i = threadIdx.x; i < PR * RQ; i += blockDim.x) {
        const int r = i / RQ, q = i - r * RQ;
        const int y = ty0 + r - PADR, x = tx0 + (q * 4) / TS;
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
        float v[NV];
#pragma unroll
        for (int co = 0; co < NV; co++) v[co] = 0.f;
#pragma unroll
        for (int kh = 0; kh < KR; kh++)
#pragma unroll
            for (int co = 0; co < NV; co++) v[co] += s_t[((ly + kh) * TW + xl) * TS + kh * NV + co];
        const long long p = (long long)y * W + x;
        if (DIFF) {
            const float m = 1.0f / (1.0f + expf(-text[p]));
            float rec[4] = {0.f, 0.f, 0.f, 0.f}, d[4] = {0.f, 0.f, 0.f, 0.f};
            const float4 xin = *(const float4*)(x0 + p * 8);
            const float xc[3] = {xin.x, xin.y, xin.z};
#pragma unroll
            for (int co = 0; co < NV && co < 3; co++) {
                rec[co] = lm_act(v[co] + bias[co], act);
                d[co] = (xc[co] - rec[co]) * m;
            }
            *(float4*)(out + p * 4) = make_float4(rec[0], rec[1], rec[2], 0.f);
            *(float4*)(diff4 + p * 4) = make_float4(d[0], d[1], d[2], 0.f);
        } else {
#pragma unroll
            for (int co = 0; co < NV; co++) out[p * ops + co] = lm_act(v[co] + bias[co], act);
        }
    }
}

// Text mask (7x7, one output) and reconstruction (3x3, three outputs) read the same input: ONE 1x7 row convolution with 16
// outputs per pixel (0..6: text kernel rows; 7 + kh * 3 + co: reconstruction rows, their three taps centred in the seven),
// then this kernel: text logit, rec = tanh(.), diff = (x0 - rec) * sigmoid(text) (:370-379).  bias: [0] text, [1..3] rec.
__global__ void __launch_bounds__(256) lm_k_vsum_text_rec(const float* __restrict__ T, int H, int W, const float* __restrict__ bias,
                                                          const float* __restrict__ x0, float* __restrict__ text, float* __restrict__ rec4,
                                                          float* __restrict__ diff4)
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
        const float4 xin = *(const float4*)(x0 + p * 8);
        text[p] = t;
        *(float4*)(rec4 + p * 4) = make_float4(r0, r1, r2, 0.f);
        *(float4*)(diff4 + p * 4) = make_float4((xin.x - r0) * m, (xin.y - r1) * m, (xin.z - r2) * m, 0.f);
    }
}

// diff = (x0 - rec) * sigmoid(text)  (:379), written to channels 0..2 of three NHWC buffers
__global__ void __launch_bounds__(256) lm_k_diff(const float* __restrict__ x0, const float* __restrict__ rec4,
                                                 const float* __restrict__ text, long long npx, float* __restrict__ o0, int s0,
                                                 float* __restrict__ o1, int s1, float* __restrict__ o2, int s2)
{
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += (long long)gridDim.x * blockDim.x) {
        const float m = 1.0f / (1.0f + expf(-text[p]));
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float d = (x0[p * 8 + c] - rec4[p * 4 + c]) * m;
            o0[p * s0 + c] = d;
            o1[p * s1 + c] = d;
            o2[p * s2 + c] = d;
        }
    }
}

// NHWC (stride 4, 3 used) -> CHW planes
__global__ void __launch_bounds__(256) lm_k_nhwc4_to_chw3(const float* __restrict__ in, float* __restrict__ out, long long npx)
{
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += (long long)gridDim.x * blockDim.x) {
        const float4 v = *(const float4*)(in + p * 4);
        out[p] = v.x; out[npx + p] = v.y; out[2 * npx + p] = v.z;
    }
}

// ================================================================================================
// host side
// ================================================================================================
#define LM_FCN_LAYERS 26
// layer ids: 0..4 conv_down_block_1..5, 5 mid_block, 6..10 transposed_conv_5..1 (+ upsample BN), 11..15 conv_up_block_5..1,
// 16 conv_text_mask_out, 17 conv_reconstruct, 18 conv_pixels_1, 19 conv_pixels_2, 20 conv_out
struct LmFcnLayer {
    float* w = nullptr;       // packed weights (MFMA layout, or [tap][C][4] for the small-Cout kernel; convT: 4 sets back to back)
    float* bias = nullptr;
    long long w_count = 0;
    int cin = 0, cout = 0, k = 0, ck = 0;
};

struct LmFcn {
    int widths[18];
    int pk, kk;
    int max_h, max_w;
    LmFcnLayer layer[LM_FCN_LAYERS];
    std::vector<float*> bufs;
    // activations (allocated for max_h x max_w)
    float *x0 = nullptr, *pre[5] = {}, *pool[5] = {}, *mid = nullptr, *up[5] = {}, *cu[5] = {};
    float *text = nullptr, *rec4 = nullptr, *px0 = nullptr, *px1 = nullptr, *px2 = nullptr, *outl = nullptr;
    float* tbuf = nullptr;      // [pixel][16] row-convolution outputs of the heads (lm_rowconv_layer, lm_text_rec_heads)
    // MFMA-head layout (fp16-split formats, 7x7 pixel kernels): x_up1, diff and the pixel features each in a buffer of their own;
    // the (diff, features) concatenations are two-input convolutions
    float *xup = nullptr, *d4 = nullptr, *p1 = nullptr, *p2 = nullptr;
};

static int lm_fcn_alloc(LmFcn* f, float** p, size_t count)
{
    LM_HIP(hipMalloc((void**)p, count * sizeof(float) + 256));
    LM_HIP(hipMemset(*p, 0, count * sizeof(float) + 256));     // padding channels must stay zero
    f->bufs.push_back(*p);
    return LM_OK;
}

extern "C" void lm_fcn_destroy(LmFcn* f)
{
    if (!f) return;
    for (float* p : f->bufs) (void)hipFree(p);
    for (auto& l : f->layer) { if (l.w) (void)hipFree(l.w); if (l.bias) (void)hipFree(l.bias); }
    delete f;
}

static inline int lm_pad8(int c) { return (c + 7) & ~7; }

extern "C" LmFcn* lm_fcn_create(const int32_t* widths18, int pixel_kernel, int kernel, int max_h, int max_w)
{
    if (!widths18 || max_h < 32 || max_w < 32 || (pixel_kernel != 1 && pixel_kernel != 3 && pixel_kernel != 5 && pixel_kernel != 7) ||
        (kernel != 3 && kernel != 1 && kernel != 5 && kernel != 7)) {
        lm_set_error("lm_fcn_create: bad arguments (frames must be at least 32 px per side, kernel sizes in {1,3,5,7})");
        return nullptr;
    }
    for (int i = 0; i < 18; i++)
        if (widths18[i] <= 0 || (widths18[i] & 7)) { lm_set_error("lm_fcn_create: layer widths must be positive multiples of 8"); return nullptr; }
    LmFcn* f = new LmFcn();
    memcpy(f->widths, widths18, sizeof(f->widths));
    f->pk = pixel_kernel; f->kk = kernel; f->max_h = max_h; f->max_w = max_w;
    const int* w = f->widths;   // d1..d5 0..4, mid 5, (u5,c5) 6,7 (u4,c4) 8,9 (u3,c3) 10,11 (u2,c2) 12,13 (u1,c1) 14,15, pm1 16, pm2 17
    size_t px[6];
    int h = max_h, ww = max_w;
    for (int l = 0; l < 6; l++) { px[l] = (size_t)h * ww; h >>= 1; ww >>= 1; }
    int rc = LM_OK;
    rc |= lm_fcn_alloc(f, &f->x0, px[0] * 8);
    for (int n = 0; n < 5; n++) {
        rc |= lm_fcn_alloc(f, &f->pre[n], px[n] * w[n]);
        rc |= lm_fcn_alloc(f, &f->pool[n], px[n + 1] * w[n]);
    }
    rc |= lm_fcn_alloc(f, &f->mid, px[5] * w[5]);
    for (int n = 0; n < 5; n++) {       // n = 0 is level 5 (deepest)
        rc |= lm_fcn_alloc(f, &f->up[n], px[4 - n] * w[6 + 2 * n]);
        if (n < 4) rc |= lm_fcn_alloc(f, &f->cu[n], px[4 - n] * w[7 + 2 * n]);
    }
    const int c1 = w[15], pm1 = w[16], pm2 = w[17];
    rc |= lm_fcn_alloc(f, &f->px0, px[0] * lm_pad8(3 + c1));       // (diff, x_up1): conv_up_block_1 writes channels 3..
    rc |= lm_fcn_alloc(f, &f->px1, px[0] * lm_pad8(3 + pm1));
    rc |= lm_fcn_alloc(f, &f->px2, px[0] * lm_pad8(3 + pm2));
    rc |= lm_fcn_alloc(f, &f->text, px[0]);
    rc |= lm_fcn_alloc(f, &f->rec4, px[0] * 4);
    rc |= lm_fcn_alloc(f, &f->outl, px[0]);
    rc |= lm_fcn_alloc(f, &f->tbuf, px[0] * 16);
    rc |= lm_fcn_alloc(f, &f->xup, px[0] * c1);
    rc |= lm_fcn_alloc(f, &f->d4, px[0] * 4);
    rc |= lm_fcn_alloc(f, &f->p1, px[0] * pm1);
    rc |= lm_fcn_alloc(f, &f->p2, px[0] * pm2);
    if (rc != LM_OK) { lm_fcn_destroy(f); return nullptr; }
    return f;
}

// Uploads one layer's host-packed weights and folded bias (packing: lecturemath_amd/fcn.py).
extern "C" int lm_fcn_set_layer(LmFcn* f, int layer, const float* h_w, int64_t w_count, const float* h_bias, int bias_count, int cin,
                                int cout, int k, int ck)
{
    if (!f || layer < 0 || layer >= LM_FCN_LAYERS || !h_w || !h_bias || w_count <= 0 || bias_count <= 0) {
        lm_set_error("lm_fcn_set_layer: bad arguments");
        return LM_ERR_ARG;
    }
    LmFcnLayer& l = f->layer[layer];
    if (l.w) (void)hipFree(l.w);
    if (l.bias) (void)hipFree(l.bias);
    l.w = nullptr; l.bias = nullptr;
    LM_HIP(hipMalloc((void**)&l.w, (size_t)w_count * sizeof(float)));
    LM_HIP(hipMalloc((void**)&l.bias, (size_t)bias_count * sizeof(float)));
    LM_HIP(hipMemcpy(l.w, h_w, (size_t)w_count * sizeof(float), hipMemcpyHostToDevice));
    LM_HIP(hipMemcpy(l.bias, h_bias, (size_t)bias_count * sizeof(float), hipMemcpyHostToDevice));
    l.w_count = w_count; l.cin = cin; l.cout = cout; l.k = k; l.ck = ck;
    return LM_OK;
}

// Operand format of one layer packed for the fp16-split kernels (the packing holds hi and lo of every weight, so any of the three
// formats can run from it): terms = 3 (hi.hi + hi.lo + lo.hi), 2 (activations split, weights rounded to f16) or 1 (both rounded).
extern "C" int lm_fcn_set_layer_terms(LmFcn* f, int layer, int terms)
{
    if (!f || layer < 0 || layer >= LM_FCN_LAYERS || terms < 1 || terms > 3) { lm_set_error("lm_fcn_set_layer_terms: bad arguments"); return LM_ERR_ARG; }
    LmFcnLayer& l = f->layer[layer];
    if (!l.w || l.ck > 0) { lm_set_error("lm_fcn_set_layer_terms: layer %d is not packed for the fp16-split kernels", layer); return LM_ERR_STATE; }
    l.ck = terms == 3 ? 0 : (terms == 2 ? -2 : -1);
    return LM_OK;
}

static size_t lm_conv_smem(int K, int CK, int NT)
{
    const int P = 16 + K - 1;
    return (size_t)P * P * (CK + 4) * 4 + (size_t)K * K * (CK / 8) * NT * 1024;
}

template <int CK, int NT> static int lm_launch_conv_t(const LmConvArgs& a, hipStream_t st)
{
    const size_t smem = lm_conv_smem(a.K, CK, NT);
#if !LM_HIP_EMULATED
    static size_t configured = 0;
    if (smem > configured) {
        LM_HIP(hipFuncSetAttribute((const void*)lm_k_conv_mfma<CK, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        configured = smem;
    }
#endif
    const int tiles = ((a.W + 15) / 16) * ((a.H + 15) / 16);
    hipLaunchKernelGGL((lm_k_conv_mfma<CK, NT>), dim3(tiles, (a.nblocks + NT - 1) / NT), dim3(256), smem, st, a);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

static int lm_conv_tg(int K) { return (K >= 5) ? K : K * K; }     // weights staged per kernel row for 5x5 / 7x7

// taps per weight group of the fp16-split kernel: a kernel row at most, and at most LM_CV_MAXW 16-B items per thread
static int lm_conv_tg_h(int K, int NT) { return lm_cv_tg(K, NT); }

static size_t lm_conv_smem_h(int K, int NT)
{
    const int P = 16 + K - 1;
    return (size_t)P * (((size_t)P * 80 + 255) & ~(size_t)255) + 2 * (size_t)lm_conv_tg_h(K, NT) * NT * 2048;   // rows padded to 256 B, weights double-buffered
}

template <int NT, int KS, int TG, int TERMS, int KH = 0, bool SWAP = false> static int lm_launch_conv_hkt(const LmConvArgs& a, hipStream_t st)
{
    const int tg = TG ? TG : lm_conv_tg_h(a.K, NT);
    const int P = 16 + a.K - 1, PR = 16 + (KH ? KH : a.K) - 1;
    const size_t smem = (size_t)PR * (((size_t)P * 80 + 255) & ~(size_t)255) + 2 * (size_t)tg * NT * 2048;   // rows padded to 256 B, weights double-buffered
#if !LM_HIP_EMULATED
    static size_t configured = 0;
    if (smem > configured) {
        LM_HIP(hipFuncSetAttribute((const void*)lm_k_conv_mfma_h<NT, KS, TG, TERMS, KH, SWAP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        configured = smem;
    }
#endif
    const int tiles = ((a.W + 15) / 16) * ((a.H + 15) / 16);
    LmConvArgs b = a;
    b.tg = tg;
    hipLaunchKernelGGL((lm_k_conv_mfma_h<NT, KS, TG, TERMS, KH, SWAP>), dim3(tiles, (a.nblocks + NT - 1) / NT), dim3(256), smem, st, b);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// 1 x K row convolution (one n-block), see lm_rowconv_layer
template <int KS> static int lm_launch_rowconv_h(const LmConvArgs& a, hipStream_t st)
{
    if (a.terms == 1) return lm_launch_conv_hkt<1, KS, 0, 1, 1, true>(a, st);
    if (a.terms == 2) return lm_launch_conv_hkt<1, KS, 0, 2, 1, true>(a, st);
    return lm_launch_conv_hkt<1, KS, 0, 3, 1, true>(a, st);
}

template <int NT, int KS, int TG = 0> static int lm_launch_conv_hk(const LmConvArgs& a, hipStream_t st)
{
    if (a.terms == 1) return lm_launch_conv_hkt<NT, KS, TG, 1>(a, st);
    if (a.terms == 2) return lm_launch_conv_hkt<NT, KS, TG, 2>(a, st);
    return lm_launch_conv_hkt<NT, KS, TG, 3>(a, st);
}

template <int NT> static int lm_launch_conv_h(const LmConvArgs& a, hipStream_t st)
{
    if (a.K > 7) { lm_set_error("lm_fcn: kernel side %d not supported by the fp16-split convolution (<= 7)", a.K); return LM_ERR_ARG; }
    const int blocks = ((a.W + 15) / 16) * ((a.H + 15) / 16) * ((a.nblocks + NT - 1) / NT);
    // two n-blocks per wave exist for 3x3 kernels only (every layer of the shipped network with >= 64 output channels is 3x3);
    // the other sides run one n-block per wave -- a third fewer instantiations of the kernel template to compile
    if constexpr (NT != 1) {
        (void)blocks;
        if (a.K != 3) { lm_set_error("lm_fcn: two n-blocks per wave are built for 3x3 kernels only"); return LM_ERR_ARG; }
        return lm_launch_conv_hk<NT, 3>(a, st);
    } else
    switch (a.K) {
        case 1: return lm_launch_conv_hk<NT, 1>(a, st);
        case 3:
            // few workgroups (deep layers): LDS is not what limits residency, so stage all nine taps at once (one barrier less
            // per chunk); many workgroups: 5 + 4 taps, three workgroups per CU
            if (NT == 1 && blocks < 1000) return lm_launch_conv_hk<NT, 3, (NT == 1) ? 9 : 0>(a, st);
            return lm_launch_conv_hk<NT, 3>(a, st);
        case 7:
            if (a.Cout <= 16 && !a.pool && ((a.ops | a.ooff) & 3) == 0) {       // 16-byte stores (SWAP)
                if (a.terms == 1) return lm_launch_conv_hkt<1, 7, 0, 1, 0, true>(a, st);
                if (a.terms == 2) return lm_launch_conv_hkt<1, 7, 0, 2, 0, true>(a, st);
                return lm_launch_conv_hkt<1, 7, 0, 3, 0, true>(a, st);
            }
            return lm_launch_conv_hk<NT, 7>(a, st);
        default: return lm_launch_conv_hk<NT, 0>(a, st);
    }
}

static inline int lm_terms_of_ck(int ck) { return ck == -1 ? 1 : (ck == -2 ? 2 : 3); }

static int lm_launch_conv(const LmConvArgs& a0, int ck, hipStream_t st)
{
    LmConvArgs a = a0;
    if (ck <= 0) {          // fp16-split packing (ck 0: three products per operand pair, -2: two, -1: one)
        a.terms = lm_terms_of_ck(ck);
        // two n-blocks per wave when the channel blocks divide evenly and the grid still has >= 1.5 workgroups per CU (four
        // would need 128 accumulator registers: one wave per SIMD, measured slower)
        const int tiles = ((a.W + 15) / 16) * ((a.H + 15) / 16);
        static const int nt2_min = [] { const char* e = getenv("LM_FCN_NT2_MIN_BLOCKS"); return e ? atoi(e) : 384; }();     // tuning experiments
        if (a.K == 3 && a.nblocks % 2 == 0 && tiles * (a.nblocks / 2) >= nt2_min) return lm_launch_conv_h<2>(a, st);
        return lm_launch_conv_h<1>(a, st);
    }
    // n-blocks per wave: as many as fit 160 KiB of LDS and divide the channel blocks evenly -- but the deep layers have few
    // spatial tiles (33x60 -> 12), so fall back to a smaller NT until the grid has at least two blocks per CU
    const int tiles = ((a.W + 15) / 16) * ((a.H + 15) / 16);
    int nt = 1;
    for (int cand : {4, 2}) {
        if (a.nblocks % cand == 0 && lm_conv_smem(a.K, ck, cand) <= 150 * 1024 && tiles * (a.nblocks / cand) >= 512) { nt = cand; break; }
    }
    if (ck == 16) {
        if (nt == 4) return lm_launch_conv_t<16, 4>(a, st);
        if (nt == 2) return lm_launch_conv_t<16, 2>(a, st);
        return lm_launch_conv_t<16, 1>(a, st);
    }
    if (nt == 4) return lm_launch_conv_t<8, 4>(a, st);
    if (nt == 2) return lm_launch_conv_t<8, 2>(a, st);
    return lm_launch_conv_t<8, 1>(a, st);
}

static int lm_conv_layer(LmFcn* f, int layer, const float* in0, int c0, int ps0, const float* in1, int c1, int ps1, int H, int W,
                         float* out, int ops, int ooff, int act, hipStream_t st, float* pool = nullptr)
{
    const LmFcnLayer& l = f->layer[layer];
    if (!l.w) { lm_set_error("lm_fcn_forward: layer %d has no weights (call lm_fcn_set_layer)", layer); return LM_ERR_STATE; }
    LmConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in0 = in0; a.c0 = c0; a.ps0 = ps0; a.in1 = in1; a.c1 = c1; a.ps1 = ps1; a.H = H; a.W = W;
    a.wpk = l.w; a.bias = l.bias; a.out = out; a.ops = ops; a.ooff = ooff; a.Cout = l.cout; a.nblocks = (l.cout + 31) / 32;
    a.K = l.k; a.act = act;
    if (l.ck <= 0 && act != LM_ACT_GELU && act != LM_ACT_NONE) { lm_set_error("lm_fcn: the fp16-split convolution kernels apply GELU or nothing"); return LM_ERR_ARG; }
    if (pool && l.ck <= 0) { a.pool = pool; a.pool_ps = l.cout; }      // the fp16-split kernel pools in its epilogue
    return lm_launch_conv(a, l.ck, st);
}

static int lm_convT_layer(LmFcn* f, int layer, const float* in, int cin, int H, int W, float* out, int OH, int OW, hipStream_t st)
{
    const LmFcnLayer& l = f->layer[layer];
    if (!l.w) { lm_set_error("lm_fcn_forward: layer %d has no weights", layer); return LM_ERR_STATE; }
    const int nblocks = (l.cout + 31) / 32;
    // floats per (dy, dx) weight set: fp32 packing [chunk][ks][nblock][64][4]; f16x3 packing [chunk][nblock][2][64] x 16 B
    const long long per_set = l.ck > 0 ? (long long)(cin / l.ck) * (l.ck / 8) * nblocks * 256 : (long long)((cin + 15) / 16) * nblocks * 512;
    if (l.ck <= 0) {        // fp16-split: one launch, the four (dy, dx) weight sets are the four taps of the packing
        LmConvArgs a;
        memset(&a, 0, sizeof(a));
        a.in0 = in; a.c0 = cin; a.ps0 = cin; a.H = H; a.W = W;
        a.wpk = l.w; a.bias = l.bias; a.out = out; a.ops = l.cout; a.ooff = 0; a.Cout = l.cout; a.nblocks = nblocks;
        a.K = 1; a.act = LM_ACT_GELU; a.tmode = 1; a.OH = OH; a.OW = OW;
        const size_t smem = (size_t)16 * ((16 * 80 + 255) & ~255) + 2 * 4 * 2048;
        const dim3 grid(((W + 15) / 16) * ((H + 15) / 16), nblocks);
        const int terms = lm_terms_of_ck(l.ck);
        if (terms == 1) hipLaunchKernelGGL(lm_k_convT_mfma_h<1>, grid, dim3(256), smem, st, a);
        else if (terms == 2) hipLaunchKernelGGL(lm_k_convT_mfma_h<2>, grid, dim3(256), smem, st, a);
        else hipLaunchKernelGGL(lm_k_convT_mfma_h<3>, grid, dim3(256), smem, st, a);
        LM_HIP(hipGetLastError());
    }
    for (int d = 0; d < 4 && l.ck > 0; d++) {
        LmConvArgs a;
        memset(&a, 0, sizeof(a));
        a.in0 = in; a.c0 = cin; a.ps0 = cin; a.H = H; a.W = W;
        a.wpk = l.w + d * per_set; a.bias = l.bias; a.out = out; a.ops = l.cout; a.ooff = 0; a.Cout = l.cout; a.nblocks = nblocks;
        a.K = 1; a.act = LM_ACT_GELU; a.tmode = 1; a.dy = d >> 1; a.dx = d & 1; a.OH = OH; a.OW = OW;
        int rc = lm_launch_conv(a, l.ck, st);
        if (rc) return rc;
    }
    if (OH > 2 * H || OW > 2 * W) {
        const long long n = ((long long)(OH - 2 * H) * OW + (long long)2 * H * (OW - 2 * W)) * l.cout;
        hipLaunchKernelGGL(lm_k_convT_border, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0, st, out, l.cout, 0,
                           OH, OW, 2 * H, 2 * W, l.cout, l.bias, LM_ACT_GELU);
    }
    return LM_OK;
}

template <int COUT> static int lm_small_launch(const LmFcnLayer& l, const float* in, int ips, int C, int H, int W, float* out, int ops, int act,
                                               hipStream_t st)
{
    const int TW = (COUT == 1) ? 32 : 16;
    const size_t smem = (size_t)(16 + l.k - 1) * (TW + l.k - 1) * 12 * 4 + (size_t)l.k * l.k * 8 * ((COUT == 1) ? 4 : 16);
#if !LM_HIP_EMULATED
    static size_t configured = 0;
    if (smem > configured) {
        LM_HIP(hipFuncSetAttribute((const void*)lm_k_conv_small<COUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        configured = smem;
    }
#endif
    const int tiles = ((W + TW - 1) / TW) * ((H + 15) / 16);
    hipLaunchKernelGGL((lm_k_conv_small<COUT>), dim3(tiles), dim3(256), smem, st, in, ips, C, H, W, l.w, l.bias, l.k, l.cout, act, out, ops);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

static int lm_small_layer(LmFcn* f, int layer, const float* in, int ips, int C, int H, int W, float* out, int ops, int act, hipStream_t st)
{
    const LmFcnLayer& l = f->layer[layer];
    if (!l.w) { lm_set_error("lm_fcn_forward: layer %d has no weights", layer); return LM_ERR_STATE; }
    if (l.cout == 1 && l.k <= 7) {
        const int PH = 16 + l.k - 1, Q = (LM_C1_TW + l.k - 1 + 3) >> 2, RF = (4 * Q * 12 + 63) & ~63;
        const size_t smem = (size_t)PH * RF * sizeof(float);
        const int tiles = ((W + LM_C1_TW - 1) / LM_C1_TW) * ((H + 15) / 16);
#if !LM_HIP_EMULATED
        static size_t configured[2] = {0, 0};
        if (smem > configured[l.k == 7]) {
            if (l.k == 7) LM_HIP(hipFuncSetAttribute((const void*)lm_k_conv_c1<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            else LM_HIP(hipFuncSetAttribute((const void*)lm_k_conv_c1<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            configured[l.k == 7] = smem;
        }
#endif
        if (l.k == 7) hipLaunchKernelGGL(lm_k_conv_c1<7>, dim3(tiles), dim3(256), smem, st, in, ips, C, H, W, l.w, l.bias, l.k, act, out, ops);
        else hipLaunchKernelGGL(lm_k_conv_c1<0>, dim3(tiles), dim3(256), smem, st, in, ips, C, H, W, l.w, l.bias, l.k, act, out, ops);
        LM_HIP(hipGetLastError());
        return LM_OK;
    }
    return (l.cout == 1) ? lm_small_launch<1>(l, in, ips, C, H, W, out, ops, act, st) : lm_small_launch<4>(l, in, ips, C, H, W, out, ops, act, st);
}

// A head with l.cout <= 3 output channels and an l.k x l.k kernel on the MFMA path (weights packed by fcn.pack_rows_h, l.ck <= 0):
// row convolution into f->tbuf, then lm_k_vsum.  bias: [0..31] zeros (the convolution's epilogue), [32..] the folded bias.
static int lm_rowconv_layer(LmFcn* f, int layer, const float* in0, int c0, const float* in1, int c1, int H, int W, float* out, int ops, int act,
                            bool diff, hipStream_t st)
{
    const LmFcnLayer& l = f->layer[layer];
    if (!l.w) { lm_set_error("lm_fcn_forward: layer %d has no weights", layer); return LM_ERR_STATE; }
    const bool seven = (l.k == 7 && l.cout == 1), three = (l.k == 3 && l.cout == 3);
    if ((!seven && !three) || (diff && !three)) {
        lm_set_error("lm_fcn_forward: layer %d: the MFMA head path covers 7x7 kernels with one output and 3x3 kernels with three (got %dx%d, %d)", layer, l.k, l.k, l.cout);
        return LM_ERR_ARG;
    }
    LmConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in0 = in0; a.c0 = c0; a.ps0 = c0; a.in1 = in1; a.c1 = c1; a.ps1 = c1; a.H = H; a.W = W;
    a.wpk = l.w; a.bias = l.bias; a.out = f->tbuf; a.ops = seven ? 8 : 12; a.ooff = 0; a.Cout = l.k * l.cout; a.nblocks = 1;
    a.K = l.k; a.act = LM_ACT_NONE; a.terms = lm_terms_of_ck(l.ck); a.krows = 1;
    int rc = seven ? lm_launch_rowconv_h<7>(a, st) : lm_launch_rowconv_h<3>(a, st);
    if (rc) return rc;
    const int tiles = ((W + 31) / 32) * ((H + 15) / 16);
    if (seven)
        hipLaunchKernelGGL((lm_k_vsum<7, 1, 8, false>), dim3(tiles), dim3(256), 0, st, f->tbuf, H, W, l.bias + 32, act, out, ops, nullptr, nullptr, nullptr);
    else if (diff)
        hipLaunchKernelGGL((lm_k_vsum<3, 3, 12, true>), dim3(tiles), dim3(256), 0, st, f->tbuf, H, W, l.bias + 32, act, out, ops, f->x0, f->text, f->d4);
    else
        hipLaunchKernelGGL((lm_k_vsum<3, 3, 12, false>), dim3(tiles), dim3(256), 0, st, f->tbuf, H, W, l.bias + 32, act, out, ops, nullptr, nullptr, nullptr);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// Text-mask and reconstruction heads as one row convolution (layer 16 packed by fcn.pack_text_rec_rows_h: l.cout == 4) + lm_k_vsum_text_rec
static int lm_text_rec_heads(LmFcn* f, const float* in, int C, int H, int W, hipStream_t st)
{
    const LmFcnLayer& l = f->layer[16];
    if (!l.w) { lm_set_error("lm_fcn_forward: layer 16 has no weights"); return LM_ERR_STATE; }
    LmConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in0 = in; a.c0 = C; a.ps0 = C; a.H = H; a.W = W;
    a.wpk = l.w; a.bias = l.bias; a.out = f->tbuf; a.ops = 16; a.ooff = 0; a.Cout = 16; a.nblocks = 1;
    a.K = 7; a.act = LM_ACT_NONE; a.terms = lm_terms_of_ck(l.ck); a.krows = 1;
    const int rc = lm_launch_rowconv_h<7>(a, st);
    if (rc) return rc;
    const int tiles = ((W + 31) / 32) * ((H + 15) / 16);
    hipLaunchKernelGGL(lm_k_vsum_text_rec, dim3(tiles), dim3(256), 0, st, f->tbuf, H, W, l.bias + 32, f->x0, f->text, f->rec4, f->d4);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// forward() of the non-reconstruction branch (:364-403) on one RGB frame resident on the device.
// Outputs (device, fp32): d_out [h*w] binarization logit, d_text [h*w] text-mask logit, d_rec [3][h*w] reconstruction.
static int lm_fcn_forward_impl(LmFcn* f, const uint8_t* d_rgb, int h, int w, float* d_out, float* d_text, float* d_rec, void* stream)
{
    if (!f || !d_rgb || h < 32 || w < 32 || h > f->max_h || w > f->max_w || (long long)h * w > (long long)f->max_h * f->max_w) {
        lm_set_error("lm_fcn_forward: bad arguments (frame %dx%d, network sized for %dx%d)", w, h, f ? f->max_w : 0, f ? f->max_h : 0);
        return LM_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const int* wd = f->widths;
    int H[6], W[6];
    H[0] = h; W[0] = w;
    for (int l = 1; l < 6; l++) { H[l] = H[l - 1] >> 1; W[l] = W[l - 1] >> 1; }
    const long long npx = (long long)h * w;
    const int c1 = wd[15], pm1 = wd[16], pm2 = wd[17];
    const int s_px0 = lm_pad8(3 + c1), s_px1 = lm_pad8(3 + pm1), s_px2 = lm_pad8(3 + pm2);
    int rc;
    const bool heads_mfma = f->layer[16].ck <= 0;       // fp16-split formats: the heads run as row convolutions on the MFMA path
    hipLaunchKernelGGL(lm_k_prepare, dim3((unsigned)std::min<long long>((npx + 255) / 256, 8192)), dim3(256), 0, st, d_rgb, f->x0, npx);
    // ---- encoder
    const float* cur = f->x0;
    int cc = 8;
    for (int n = 0; n < 5; n++) {
        if ((rc = lm_conv_layer(f, n, cur, cc, cc, nullptr, 0, 0, H[n], W[n], f->pre[n], wd[n], 0, LM_ACT_GELU, st, f->pool[n]))) return rc;
        if (f->layer[n].ck > 0) {       // fp32 kernel: pooling is a pass of its own
            const long long tot = (long long)H[n + 1] * W[n + 1] * (wd[n] / 4);
            hipLaunchKernelGGL(lm_k_maxpool2, dim3((unsigned)std::min<long long>((tot + 255) / 256, 8192)), dim3(256), 0, st, f->pre[n], wd[n],
                               f->pool[n], wd[n], H[n], W[n], wd[n]);
        }
        cur = f->pool[n];
        cc = wd[n];
    }
    if ((rc = lm_conv_layer(f, 5, cur, cc, cc, nullptr, 0, 0, H[5], W[5], f->mid, wd[5], 0, LM_ACT_GELU, st))) return rc;
    // ---- decoder
    cur = f->mid;
    cc = wd[5];
    for (int n = 0; n < 5; n++) {           // level L = 5 - n, works on grid H[4 - n]
        const int g = 4 - n;
        const int cu = wd[6 + 2 * n], co = wd[7 + 2 * n];
        if ((rc = lm_convT_layer(f, 6 + n, cur, cc, H[g + 1], W[g + 1], f->up[n], H[g], W[g], st))) return rc;
        float* dst = (n < 4) ? f->cu[n] : (heads_mfma ? f->xup : f->px0);
        const int ops = (n < 4) ? co : (heads_mfma ? c1 : s_px0), ooff = (n < 4 || heads_mfma) ? 0 : 3;
        if ((rc = lm_conv_layer(f, 11 + n, f->up[n], cu, cu, f->pre[g], wd[g], wd[g], H[g], W[g], dst, ops, ooff, LM_ACT_GELU, st))) return rc;
        cur = dst;
        cc = co;
    }
    // ---- heads.  x_up1 lives in channels 3..3+c1 of px0 (pixel stride s_px0)
    const float* xup = f->px0 + 3;
    // x_up1 is not 16-byte aligned at channel offset 3, so the small kernel reads the whole (diff, x_up1) buffer with zero
    // weights on channels 0..2 (the host packs them that way)
    if (heads_mfma) {
        if (f->layer[16].cout == 4) {       // text mask + reconstruction packed together
            if ((rc = lm_text_rec_heads(f, f->xup, c1, h, w, st))) return rc;
        } else {
            if ((rc = lm_rowconv_layer(f, 16, f->xup, c1, nullptr, 0, h, w, f->text, 1, LM_ACT_NONE, false, st))) return rc;
            if ((rc = lm_rowconv_layer(f, 17, f->xup, c1, nullptr, 0, h, w, f->rec4, 4, LM_ACT_TANH, true, st))) return rc;
        }
        if ((rc = lm_conv_layer(f, 18, f->d4, 4, 4, f->xup, c1, c1, h, w, f->p1, pm1, 0, LM_ACT_GELU, st))) return rc;
        if ((rc = lm_conv_layer(f, 19, f->d4, 4, 4, f->p1, pm1, pm1, h, w, f->p2, pm2, 0, LM_ACT_GELU, st))) return rc;
        if ((rc = lm_rowconv_layer(f, 20, f->d4, 4, f->p2, pm2, h, w, f->outl, 1, LM_ACT_NONE, false, st))) return rc;
    } else {
        if ((rc = lm_small_layer(f, 16, f->px0, s_px0, s_px0, h, w, f->text, 1, LM_ACT_NONE, st))) return rc;
        if ((rc = lm_small_layer(f, 17, f->px0, s_px0, s_px0, h, w, f->rec4, 4, LM_ACT_TANH, st))) return rc;
        hipLaunchKernelGGL(lm_k_diff, dim3((unsigned)std::min<long long>((npx + 255) / 256, 8192)), dim3(256), 0, st, f->x0, f->rec4, f->text, npx,
                           f->px0, s_px0, f->px1, s_px1, f->px2, s_px2);
        if ((rc = lm_conv_layer(f, 18, f->px0, s_px0, s_px0, nullptr, 0, 0, h, w, f->px1, s_px1, 3, LM_ACT_GELU, st))) return rc;
        if ((rc = lm_conv_layer(f, 19, f->px1, s_px1, s_px1, nullptr, 0, 0, h, w, f->px2, s_px2, 3, LM_ACT_GELU, st))) return rc;
        if ((rc = lm_small_layer(f, 20, f->px2, s_px2, s_px2, h, w, f->outl, 1, LM_ACT_NONE, st))) return rc;
    }
    (void)xup;
    if (d_out) LM_HIP(hipMemcpyAsync(d_out, f->outl, (size_t)npx * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (d_text) LM_HIP(hipMemcpyAsync(d_text, f->text, (size_t)npx * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (d_rec) hipLaunchKernelGGL(lm_k_nhwc4_to_chw3, dim3((unsigned)std::min<long long>((npx + 255) / 256, 8192)), dim3(256), 0, st, f->rec4, d_rec, npx);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// Forward passes issued on different HIP streams (or by different engines) may overlap on the device: every engine owns its
// activation buffers, and the kernels hold no state between launches.  (Round 1 chained them behind an event because two
// passes in flight disturbed each other's one-channel heads; the cause was the packed-fp32 code of those kernels, see
// LM_NO_PACKED_F32 above, tools/fcn_overlap_repro.py and tests/test_cc_gpu.py::test_fcn_two_engines_on_two_streams.)
extern "C" int lm_fcn_forward(LmFcn* f, const uint8_t* d_rgb, int h, int w, float* d_out, float* d_text, float* d_rec, void* stream)
{
    return lm_fcn_forward_impl(f, d_rgb, h, w, d_out, d_text, d_rec, stream);
}
