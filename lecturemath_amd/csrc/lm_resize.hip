// lm_resize.hip -- the frame pre / post-processing of FCN_LectureNet.binarize's > 2.5 MP branch on the device
// (AccessMath/lecturenet_v1/FCN_lecturenet.py relative to /root/reference/ACCESS2021_release):
//   :434-437  PIL_image.resize((w // 2, h // 2), PIL.Image.LANCZOS) while the frame has more than 2.5 MP   lm_resample_rgb8
//   :481-486  cv2.resize(binary, (o_w, o_h), interpolation=cv2.INTER_NEAREST) back to the frame's size           lm_upsample_nearest_u8
// Pillow's resize (src/libImaging/Resample.c) is two passes -- horizontal, then vertical over the 8-bit result of the first -- of a
// separable filter whose per-output-pixel taps are 22-bit fixed-point integers: out = clip8((sum_x in[xmin + x] * k[x] + 2^21) >> 22).
// The taps and their bounds depend on the sizes only; lecturemath_amd/resize.py computes them once per size pair (float64, the
// formulas of precompute_coeffs / normalize_coeffs_8bpc) and keeps them on the device; the pixel arithmetic is all integer, so the
// result equals Pillow's byte for byte (tests/golden/g6b_lanczos.npz).  HBM-bound streaming kernels, rows of the 3-channel image.
#include "lm_common.h"

#define LM_RS_BITS 22

LM_DEV uint8_t lm_clip8(int v)
{
    v >>= LM_RS_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: in [H][in_w][C] -> out [H][out_w][C]; a thread per output byte, rows in grid.y
template <int C>
__global__ void __launch_bounds__(256) lm_k_resample_h(const uint8_t* __restrict__ in, int in_w, uint8_t* __restrict__ out, int out_w, int H,
                                                       const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ksize)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= out_w * C) return;
    const int xx = i / C, c = i - xx * C;
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int32_t* k = kk + (long long)xx * ksize;
    for (int y = blockIdx.y; y < H; y += gridDim.y) {
        const uint8_t* row = in + ((long long)y * in_w + xmin) * C + c;
        int acc = 1 << (LM_RS_BITS - 1);
        for (int x = 0; x < n; x++) acc += (int)row[x * C] * k[x];
        out[((long long)y * out_w) * C + i] = lm_clip8(acc);
    }
}

// vertical pass: in [in_h][row_bytes] -> out [out_h][row_bytes]; a thread per byte of an output row, 4 bytes at a time when aligned
__global__ void __launch_bounds__(256) lm_k_resample_v(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int out_h, long long row_bytes,
                                                       const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ksize)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= row_bytes) return;
    for (int yy = blockIdx.y; yy < out_h; yy += gridDim.y) {
        const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
        const int32_t* k = kk + (long long)yy * ksize;
        int acc = 1 << (LM_RS_BITS - 1);
        for (int y = 0; y < n; y++) acc += (int)in[(long long)(ymin + y) * row_bytes + i] * k[y];
        out[(long long)yy * row_bytes + i] = lm_clip8(acc);
    }
}

// cv2.INTER_NEAREST as the reference's callers use it (integer ratio): out[y][x] = in[y * in_h / out_h][x * in_w / out_w]
template <int C>
__global__ void __launch_bounds__(256) lm_k_upsample_nearest(const uint8_t* __restrict__ in, int in_h, int in_w, uint8_t* __restrict__ out, int out_h,
                                                             int out_w)
{
    const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (x >= out_w) return;
    const int sx = (int)(((long long)x * in_w) / out_w);
    for (int y = blockIdx.y; y < out_h; y += gridDim.y) {
        const int sy = (int)(((long long)y * in_h) / out_h);
#pragma unroll
        for (int c = 0; c < C; c++) out[((long long)y * out_w + x) * C + c] = in[((long long)sy * in_w + sx) * C + c];
    }
}

// d_tmp: [in_h][out_w][channels] bytes of scratch (the horizontal pass' 8-bit result); bounds / kk: DEVICE tables of
// lecturemath_amd.resize.coefficients (h: per output column, v: per output row).  channels: 1 or 3.  An axis whose size does not
// change is skipped, as Pillow skips it.
extern "C" int lm_resample_rgb8(const uint8_t* d_in, int in_h, int in_w, int channels, uint8_t* d_tmp, uint8_t* d_out, int out_h, int out_w,
                                const int32_t* d_bounds_h, const int32_t* d_kk_h, int ksize_h, const int32_t* d_bounds_v, const int32_t* d_kk_v, int ksize_v,
                                void* stream)
{
    if (!d_in || !d_out || !d_tmp || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || (channels != 1 && channels != 3) ||
        (in_w != out_w && (!d_bounds_h || !d_kk_h || ksize_h <= 0)) || (in_h != out_h && (!d_bounds_v || !d_kk_v || ksize_v <= 0))) {
        lm_set_error("lm_resample_rgb8: bad arguments");
        return LM_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool horiz = in_w != out_w, vert = in_h != out_h;
    const uint8_t* src = d_in;
    if (horiz) {
        uint8_t* dst = vert ? d_tmp : d_out;
        const dim3 grid((unsigned)((out_w * channels + 255) / 256), (unsigned)std::min(in_h, 4096));
        if (channels == 3) hipLaunchKernelGGL(lm_k_resample_h<3>, grid, dim3(256), 0, st, src, in_w, dst, out_w, in_h, d_bounds_h, d_kk_h, ksize_h);
        else hipLaunchKernelGGL(lm_k_resample_h<1>, grid, dim3(256), 0, st, src, in_w, dst, out_w, in_h, d_bounds_h, d_kk_h, ksize_h);
        src = dst;
    }
    if (vert) {
        const long long row_bytes = (long long)out_w * channels;
        hipLaunchKernelGGL(lm_k_resample_v, dim3((unsigned)((row_bytes + 255) / 256), (unsigned)std::min(out_h, 4096)), dim3(256), 0, st, src, d_out, out_h,
                           row_bytes, d_bounds_v, d_kk_v, ksize_v);
    } else if (!horiz) {
        LM_HIP(hipMemcpyAsync(d_out, d_in, (size_t)in_h * in_w * channels, hipMemcpyDeviceToDevice, st));
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}

extern "C" int lm_upsample_nearest_u8(const uint8_t* d_in, int in_h, int in_w, int channels, uint8_t* d_out, int out_h, int out_w, void* stream)
{
    if (!d_in || !d_out || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || (channels != 1 && channels != 3)) {
        lm_set_error("lm_upsample_nearest_u8: bad arguments");
        return LM_ERR_ARG;
    }
    const dim3 grid((unsigned)((out_w + 255) / 256), (unsigned)std::min(out_h, 4096));
    if (channels == 3) hipLaunchKernelGGL(lm_k_upsample_nearest<3>, grid, dim3(256), 0, (hipStream_t)stream, d_in, in_h, in_w, d_out, out_h, out_w);
    else hipLaunchKernelGGL(lm_k_upsample_nearest<1>, grid, dim3(256), 0, (hipStream_t)stream, d_in, in_h, in_w, d_out, out_h, out_w);
    LM_HIP(hipGetLastError());
    return LM_OK;
}
