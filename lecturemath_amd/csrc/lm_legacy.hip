// lm_legacy.hip -- the remaining exports of the reference's accessmath_lib.c, so that the library is a complete stand-in for
// ./accessmath_lib.so (same symbols, same C signatures, host pointers as ctypes hands them over):
//   speaker_detection_handle_frame   accessmath_lib.c:7-111     frame differencing statistics (speaker / motion detection)
//   regionCumulativeDistribution     accessmath_lib.c:113-173   contrast-limited, centred CDF of one image region
//   adapthisteq                      accessmath_lib.c:175-329   CLAHE: per-tile CDFs + bilinear interpolation per pixel
//   combine_results                  accessmath_lib.c:331-355   legacy binarizer: board mask + equalised image -> binary
// These belong to the classical (pre-FCN) binarizers (binarizer.py:139-246, adaptive_equalizer.py:274-291), "next" row 4 of
// SURVEY 8(f).  Pixel passes run on the device; the short float64 recurrences (256-bin CDF, per-column variance) are
// finished on the host with the reference's own statement order, so results are bit-identical to the C library
// (floating-point contraction is switched off in the interpolation kernel: x86-64 gcc does not fuse multiply-adds).
#include "lm_common.h"

#include <math.h>
#include <vector>

// no fused multiply-adds anywhere in this file (host recurrences and the interpolation kernel must round like x86-64 gcc)
#pragma clang fp contract(off)

struct LmRegion { int x0, x1, y0, y1; };

// 256-bin histogram of every region; one workgroup per (region, slice of rows), LDS bins, one atomic per bin and workgroup
__global__ void __launch_bounds__(256) lm_k_region_hist(const uint8_t* __restrict__ gray, int width, const LmRegion* __restrict__ regions,
                                                        int* __restrict__ hist)
{
    __shared__ int s_hist[256];
    const LmRegion r = regions[blockIdx.x];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int rw = r.x1 - r.x0 + 1, rh = r.y1 - r.y0 + 1;
    const long long total = (long long)rw * rh;
    for (long long i = (long long)blockIdx.y * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.y * blockDim.x) {
        const int y = (int)(i / rw), x = (int)(i - (long long)y * rw);
        atomicAdd(&s_hist[gray[(long long)(r.y0 + y) * width + r.x0 + x]], 1);
    }
    __syncthreads();
    if (s_hist[threadIdx.x]) atomicAdd(&hist[blockIdx.x * 256 + threadIdx.x], s_hist[threadIdx.x]);
}

// CLAHE interpolation (:239-318).  lim = [x_max[gx] | x_mid[gx] | y_max[gy] | y_mid[gy]]
__global__ void __launch_bounds__(256) lm_k_clahe_apply(const uint8_t* __restrict__ gray, int width, int height, const double* __restrict__ dist,
                                                        const int* __restrict__ lim, int grid_x, int grid_y, uint8_t* __restrict__ out)
{
    const int* x_max = lim;
    const int* x_mid = lim + grid_x;
    const int* y_max = lim + 2 * grid_x;
    const int* y_mid = lim + 2 * grid_x + grid_y;
    const long long n = (long long)width * height;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(p / width), x = (int)(p - (long long)y * width);
        int cx = 0, cy = 0;
        while (x > x_max[cx]) cx++;
        while (y > y_max[cy]) cy++;
        const uint8_t tone = gray[p];
        const double* local = dist + (long long)(grid_x * cy + cx) * 256;
        uint8_t eq;
        const bool edge_x = (cx == 0 && x <= x_mid[cx]) || (cx == grid_x - 1 && x >= x_mid[cx]);
        const bool edge_y = (cy == 0 && y <= y_mid[cy]) || (cy == grid_y - 1 && y >= y_mid[cy]);
        if (edge_x) {
            if (edge_y) {
                eq = (uint8_t)round(local[tone] * 255);
            } else {
                const int y0 = cy - (y <= y_mid[cy] ? 1 : 0), y1 = y0 + 1;
                const double wy1 = (y - y_mid[y0]) / (double)(y_mid[y1] - y_mid[y0]);
                const double* d00 = dist + (long long)(grid_x * y0 + cx) * 256;
                const double* d01 = dist + (long long)(grid_x * y1 + cx) * 256;
                eq = (uint8_t)round((d00[tone] * (1.0 - wy1) + d01[tone] * wy1) * 255);
            }
        } else if (edge_y) {
            const int x0 = cx - (x <= x_mid[cx] ? 1 : 0), x1 = x0 + 1;
            const double wx1 = (x - x_mid[x0]) / (double)(x_mid[x1] - x_mid[x0]);
            const double* d00 = dist + (long long)(grid_x * cy + x0) * 256;
            const double* d10 = dist + (long long)(grid_x * cy + x1) * 256;
            eq = (uint8_t)round((d00[tone] * (1.0 - wx1) + d10[tone] * wx1) * 255);
        } else {
            const int x0 = cx - (x <= x_mid[cx] ? 1 : 0), x1 = x0 + 1;
            const double wx1 = (x - x_mid[x0]) / (double)(x_mid[x1] - x_mid[x0]);
            const int y0 = cy - (y <= y_mid[cy] ? 1 : 0), y1 = y0 + 1;
            const double wy1 = (y - y_mid[y0]) / (double)(y_mid[y1] - y_mid[y0]);
            const double* d00 = dist + (long long)(grid_x * y0 + x0) * 256;
            const double* d01 = dist + (long long)(grid_x * y1 + x0) * 256;
            const double* d10 = dist + (long long)(grid_x * y0 + x1) * 256;
            const double* d11 = dist + (long long)(grid_x * y1 + x1) * 256;
            eq = (uint8_t)round((d00[tone] * (1.0 - wx1) * (1.0 - wy1) + d01[tone] * (1.0 - wx1) * wy1 + d10[tone] * wx1 * (1.0 - wy1) +
                                 d11[tone] * wx1 * wy1) * 255);
        }
        out[p] = eq;
    }
}

__global__ void __launch_bounds__(256) lm_k_combine(const uint8_t* __restrict__ only_board, const uint8_t* __restrict__ equalized, long long n,
                                                    int threshold, uint8_t* __restrict__ out)
{
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x)
        out[p] = (only_board[p] > 128) ? 0 : ((int)equalized[p] < threshold ? 255 : 0);
}

// sampled pixels (every jump-th row and column) with one colour channel changed by more than threshold (:33-72)
// acc = [total, min_x, max_x, min_y, max_y]; sums = [sum_x, sum_y]
__global__ void __launch_bounds__(256) lm_k_speaker_diff(const uint8_t* __restrict__ frame, const uint8_t* __restrict__ last, int width, int height,
                                                         int channels, int threshold, int jump, int* __restrict__ count_x, int* __restrict__ count_y,
                                                         int* __restrict__ acc, unsigned long long* __restrict__ sums)
{
    const int nx = (width + jump - 1) / jump, ny = (height + jump - 1) / jump;
    const long long n = (long long)nx * ny;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / nx) * jump, col = (int)(i % nx) * jump;
        const long long off = ((long long)row * width + col) * channels;
        bool changed = false;
        for (int c = 0; c < channels; c++) {
            int d = (int)last[off + c] - (int)frame[off + c];
            if ((d < 0 ? -d : d) > threshold) { changed = true; break; }
        }
        if (changed) {
            atomicAdd(&count_x[col], 1);
            atomicAdd(&count_y[row], 1);
            atomicAdd(&acc[0], 1);
            atomicMin(&acc[1], col); atomicMax(&acc[2], col);
            atomicMin(&acc[3], row); atomicMax(&acc[4], row);
            atomicAdd(&sums[0], (unsigned long long)col);
            atomicAdd(&sums[1], (unsigned long long)row);
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
namespace {
struct LmScratch {      // frees its device buffers on every exit path
    std::vector<void*> ptrs;
    ~LmScratch() { for (void* p : ptrs) if (p) (void)hipFree(p); }
    template <class T> T* alloc(size_t n)
    {
        void* p = nullptr;
        if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return (T*)p;
    }
};

// contrast-limited, centred CDF from a 256-bin histogram (:138-172), the reference's statement order
void lm_cdf_from_hist(const int* hist, double slope_max, double* output)
{
    int count = 0;
    for (int i = 0; i < 256; i++) { count += hist[i]; output[i] = count; }
    for (int i = 0; i < 256; i++) output[i] /= count;
    if (slope_max > 0.0) {
        double dh = 0.0;
        for (int i = 0; i < 255; i++) {
            const double diff = output[i + 1] - output[i] - dh - slope_max;
            dh += (diff < 0.0 ? 0.0 : diff);
            output[i + 1] -= dh;
        }
        const double add = (1.0 - (output[255] - output[0])) / 2.0;
        for (int i = 0; i < 256; i++) output[i] += add;
    }
}

// histograms of `regions` of a host image; d_gray_out (optional) receives the device copy of the image
int lm_region_hists(LmScratch& sc, const unsigned char* grayscale, int width, int height, const std::vector<LmRegion>& regions,
                    std::vector<int>& hist, uint8_t** d_gray_out)
{
    const size_t n = (size_t)width * height;
    uint8_t* d_gray = sc.alloc<uint8_t>(n);
    LmRegion* d_reg = sc.alloc<LmRegion>(regions.size());
    int* d_hist = sc.alloc<int>(regions.size() * 256);
    if (!d_gray || !d_reg || !d_hist) { lm_set_error("accessmath_lib: out of device memory"); return LM_ERR_HIP; }
    LM_HIP(hipMemcpy(d_gray, grayscale, n, hipMemcpyHostToDevice));
    LM_HIP(hipMemcpy(d_reg, regions.data(), regions.size() * sizeof(LmRegion), hipMemcpyHostToDevice));
    LM_HIP(hipMemset(d_hist, 0, regions.size() * 256 * sizeof(int)));
    hipLaunchKernelGGL(lm_k_region_hist, dim3((unsigned)regions.size(), LM_HIP_EMULATED ? 1 : 16), dim3(256), 0, (hipStream_t)0, d_gray, width, d_reg,
                       d_hist);
    hist.resize(regions.size() * 256);
    LM_HIP(hipMemcpy(hist.data(), d_hist, hist.size() * sizeof(int), hipMemcpyDeviceToHost));
    if (d_gray_out) *d_gray_out = d_gray;
    return LM_OK;
}
}  // namespace

extern "C" void regionCumulativeDistribution(unsigned char* grayscale, int width, int height, int min_x, int max_x, int min_y, int max_y,
                                             double slope_max, double* output)
{
    if (!grayscale || !output || width <= 0 || height <= 0 || min_x < 0 || min_y < 0 || max_x >= width || max_y >= height || min_x > max_x ||
        min_y > max_y) {
        lm_set_error("regionCumulativeDistribution: bad arguments");
        return;
    }
    LmScratch sc;
    std::vector<int> hist;
    if (lm_region_hists(sc, grayscale, width, height, {LmRegion{min_x, max_x, min_y, max_y}}, hist, nullptr)) return;
    lm_cdf_from_hist(hist.data(), slope_max, output);
}

extern "C" int adapthisteq(unsigned char* grayscale, int width, int height, double slope, int grid_x, int grid_y, unsigned char* output)
{
    if (!grayscale || !output || width <= 0 || height <= 0 || grid_x <= 0 || grid_y <= 0 || grid_x > width || grid_y > height) {
        lm_set_error("adapthisteq: bad arguments");
        return 0;       // the reference has no error path: callers ignore the value
    }
    // tile limits (:192-219)
    std::vector<int> lim((size_t)2 * grid_x + 2 * grid_y);
    int* x_max = lim.data();
    int* x_mid = x_max + grid_x;
    int* y_max = x_mid + grid_x;
    int* y_mid = y_max + grid_y;
    std::vector<int> x_min((size_t)grid_x), y_min((size_t)grid_y);
    const int min_size_x = width / grid_x, min_size_y = height / grid_y, mod_x = width % grid_x, mod_y = height % grid_y;
    int start = 0;
    for (int rx = 0; rx < grid_x; rx++) {
        const int end = start + min_size_x + (rx < mod_x ? 1 : 0) - 1;
        x_min[rx] = start; x_max[rx] = end; x_mid[rx] = (int)round((start + end) / 2.0);
        start = end + 1;
    }
    start = 0;
    for (int ry = 0; ry < grid_y; ry++) {
        const int end = start + min_size_y + (ry < mod_y ? 1 : 0) - 1;
        y_min[ry] = start; y_max[ry] = end; y_mid[ry] = (int)round((start + end) / 2.0);
        start = end + 1;
    }
    std::vector<LmRegion> regions((size_t)grid_x * grid_y);
    for (int ry = 0; ry < grid_y; ry++)
        for (int rx = 0; rx < grid_x; rx++) regions[(size_t)ry * grid_x + rx] = LmRegion{x_min[rx], x_max[rx], y_min[ry], y_max[ry]};
    LmScratch sc;
    std::vector<int> hist;
    uint8_t* d_gray = nullptr;
    if (lm_region_hists(sc, grayscale, width, height, regions, hist, &d_gray)) return 0;
    std::vector<double> dist(regions.size() * 256);
    for (size_t r = 0; r < regions.size(); r++) lm_cdf_from_hist(hist.data() + r * 256, slope, dist.data() + r * 256);
    const size_t n = (size_t)width * height;
    double* d_dist = sc.alloc<double>(dist.size());
    int* d_lim = sc.alloc<int>(lim.size());
    uint8_t* d_out = sc.alloc<uint8_t>(n);
    if (!d_dist || !d_lim || !d_out) { lm_set_error("adapthisteq: out of device memory"); return 0; }
    if (hipMemcpy(d_dist, dist.data(), dist.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_lim, lim.data(), lim.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        lm_set_error("adapthisteq: upload failed");
        return 0;
    }
    hipLaunchKernelGGL(lm_k_clahe_apply, dim3(LM_HIP_EMULATED ? 2 : 2048), dim3(256), 0, (hipStream_t)0, d_gray, width, height, d_dist, d_lim, grid_x,
                       grid_y, d_out);
    if (hipMemcpy(output, d_out, n, hipMemcpyDeviceToHost) != hipSuccess) lm_set_error("adapthisteq: download failed");
    return 0;
}

extern "C" int combine_results(unsigned char* only_board, unsigned char* equalized, int width, int height, unsigned char threshold,
                               unsigned char* final_content)
{
    if (!only_board || !equalized || !final_content || width <= 0 || height <= 0) { lm_set_error("combine_results: bad arguments"); return 0; }
    const size_t n = (size_t)width * height;
    LmScratch sc;
    uint8_t* d_a = sc.alloc<uint8_t>(n);
    uint8_t* d_b = sc.alloc<uint8_t>(n);
    uint8_t* d_o = sc.alloc<uint8_t>(n);
    if (!d_a || !d_b || !d_o) { lm_set_error("combine_results: out of device memory"); return 0; }
    if (hipMemcpy(d_a, only_board, n, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_b, equalized, n, hipMemcpyHostToDevice) != hipSuccess) {
        lm_set_error("combine_results: upload failed");
        return 0;
    }
    hipLaunchKernelGGL(lm_k_combine, dim3(LM_HIP_EMULATED ? 2 : 2048), dim3(256), 0, (hipStream_t)0, d_a, d_b, (long long)n, (int)threshold, d_o);
    if (hipMemcpy(final_content, d_o, n, hipMemcpyDeviceToHost) != hipSuccess) lm_set_error("combine_results: download failed");
    return 0;
}

extern "C" int speaker_detection_handle_frame(unsigned char* frame, unsigned char* last_frame, int width, int height, int channels, int threshold,
                                              int jump_cells, double* change_boundaries, double* change_avg, double* change_deviation)
{
    if (!frame || !last_frame || !change_boundaries || !change_avg || !change_deviation || width <= 0 || height <= 0 || channels <= 0 ||
        jump_cells <= 0) {
        lm_set_error("speaker_detection_handle_frame: bad arguments");
        return 0;
    }
    const size_t n = (size_t)width * height * channels;
    LmScratch sc;
    uint8_t* d_f = sc.alloc<uint8_t>(n);
    uint8_t* d_l = sc.alloc<uint8_t>(n);
    int* d_cx = sc.alloc<int>((size_t)width);
    int* d_cy = sc.alloc<int>((size_t)height);
    int* d_acc = sc.alloc<int>(5);
    unsigned long long* d_sums = sc.alloc<unsigned long long>(2);
    if (!d_f || !d_l || !d_cx || !d_cy || !d_acc || !d_sums) { lm_set_error("speaker_detection_handle_frame: out of device memory"); return 0; }
    const int acc0[5] = {0, width + 1, -1, height + 1, -1};
    bool ok = hipMemcpy(d_f, frame, n, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(d_l, last_frame, n, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemset(d_cx, 0, (size_t)width * sizeof(int)) == hipSuccess && hipMemset(d_cy, 0, (size_t)height * sizeof(int)) == hipSuccess &&
              hipMemcpy(d_acc, acc0, sizeof(acc0), hipMemcpyHostToDevice) == hipSuccess && hipMemset(d_sums, 0, 2 * sizeof(unsigned long long)) == hipSuccess;
    if (!ok) { lm_set_error("speaker_detection_handle_frame: upload failed"); return 0; }
    hipLaunchKernelGGL(lm_k_speaker_diff, dim3(LM_HIP_EMULATED ? 2 : 1024), dim3(256), 0, (hipStream_t)0, d_f, d_l, width, height, channels, threshold,
                       jump_cells, d_cx, d_cy, d_acc, d_sums);
    std::vector<int> cx((size_t)width), cy((size_t)height);
    int acc[5];
    unsigned long long sums[2];
    ok = hipMemcpy(cx.data(), d_cx, cx.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(cy.data(), d_cy, cy.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(acc, d_acc, sizeof(acc), hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(sums, d_sums, sizeof(sums), hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) { lm_set_error("speaker_detection_handle_frame: download failed"); return 0; }
    const int total_changes = acc[0];
    change_boundaries[0] = acc[1]; change_boundaries[1] = acc[2]; change_boundaries[2] = acc[3]; change_boundaries[3] = acc[4];
    // the sums of integer coordinates are exact in float64 whatever the order (:62-63)
    change_avg[0] = (double)sums[0];
    change_avg[1] = (double)sums[1];
    if (total_changes > 0) {
        change_avg[0] /= (double)total_changes;
        change_avg[1] /= (double)total_changes;
        change_deviation[0] = 0.0;
        change_deviation[1] = 0.0;
        for (int col = 0; col < width; col++) change_deviation[0] += ((double)col - change_avg[0]) * ((double)col - change_avg[0]) * (double)cx[col];
        for (int row = 0; row < height; row++) change_deviation[1] += ((double)row - change_avg[1]) * ((double)row - change_avg[1]) * (double)cy[row];
        change_deviation[0] /= (double)total_changes;
        change_deviation[1] /= (double)total_changes;
        change_deviation[0] = sqrt(change_deviation[0]);
        change_deviation[1] = sqrt(change_deviation[1]);
    } else {
        change_deviation[0] = 0.0;
        change_deviation[1] = 0.0;
    }
    return total_changes;
}
