// Store-pattern study for lm_k_write_labels: which part of the pattern keeps it at half the fill rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// A: plain fill, one 16-B store per thread per trip, fully coalesced grid-stride
__global__ void kA(int4* out, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        out[i] = make_int4(1, 2, 3, 4);
}
// B: value depends on a broadcast load of a bit word (16 quads share one 8-B word)
__global__ void kB(int4* out, const unsigned long long* bits, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        unsigned nib = (unsigned)(bits[i >> 4] >> ((i & 15) * 4)) & 15u;
        out[i] = make_int4(nib & 1, nib & 2, nib & 4, nib & 8);
    }
}
// C: like B with 4 quads per thread at stride 64 (the shipped structure), per-frame 2-D grid
__global__ void kC(int4* out, const unsigned long long* bits, unsigned per_frame) {
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    int4* o = out + (long long)blockIdx.y * per_frame;
    const unsigned long long* b = bits + (long long)blockIdx.y * (per_frame >> 4);
    for (unsigned base = wave * 256; base < per_frame; base += nw * 256) {
        unsigned nib[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { unsigned g = base + k * 64 + lane; nib[k] = g < per_frame ? (unsigned)(b[g >> 4] >> ((g & 15) * 4)) & 15u : 0; }
#pragma unroll
        for (int k = 0; k < 4; k++) { unsigned g = base + k * 64 + lane; if (g < per_frame) o[g] = make_int4(nib[k] & 1, nib[k] & 2, nib[k] & 4, nib[k] & 8); }
    }
}
// D: non-temporal stores
__global__ void kD(int4* out, const unsigned long long* bits, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        unsigned nib = (unsigned)(bits[i >> 4] >> ((i & 15) * 4)) & 15u;
        int4 v = make_int4(nib & 1, nib & 2, nib & 4, nib & 8);
        __builtin_nontemporal_store(v.x, &out[i].x); __builtin_nontemporal_store(v.y, &out[i].y);
        __builtin_nontemporal_store(v.z, &out[i].z); __builtin_nontemporal_store(v.w, &out[i].w);
    }
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    hipEventRecord(a); for (int i = 0; i < 20; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 20 * 1000;
}
int main() {
    const int B = 32; const long long px = 1920LL * 1080, n4 = B * px / 4;
    int4* out; unsigned long long* bits;
    CHECK(hipMalloc(&out, n4 * 16)); CHECK(hipMalloc(&bits, n4 / 16 * 8 + 64));
    std::vector<unsigned long long> h(n4 / 16 + 8);
    for (size_t i = 0; i < h.size(); i++) h[i] = (i % 7 == 0) ? 0x00ff00f0f0ull << (i % 13) : 0;
    CHECK(hipMemcpy(bits, h.data(), n4 / 16 * 8, hipMemcpyHostToDevice));
    const double mb = n4 * 16 / 1e6;
    for (int blocks : {2048, 8192, 32768, 65536}) {
        float t = timeit([&] { hipLaunchKernelGGL(kA, dim3(blocks), dim3(256), 0, 0, out, n4); });
        printf("A fill            blocks %6d: %6.1f us  %.2f TB/s\n", blocks, t, mb / t);
        t = timeit([&] { hipLaunchKernelGGL(kB, dim3(blocks), dim3(256), 0, 0, out, bits, n4); });
        printf("B bits-dependent  blocks %6d: %6.1f us  %.2f TB/s\n", blocks, t, mb / t);
        t = timeit([&] { hipLaunchKernelGGL(kD, dim3(blocks), dim3(256), 0, 0, out, bits, n4); });
        printf("D nontemporal     blocks %6d: %6.1f us  %.2f TB/s\n", blocks, t, mb / t);
    }
    // rotate over 6 buffers (1.6 GB): the 256 MiB Infinity Cache cannot absorb the writes any more
    int4* outs[6];
    for (int i = 0; i < 6; i++) CHECK(hipMalloc(&outs[i], n4 * 16));
    int rot = 0;
    for (int blocks : {8192, 32768}) {
        float t = timeit([&] { hipLaunchKernelGGL(kA, dim3(blocks), dim3(256), 0, 0, outs[rot++ % 6], n4); });
        printf("A fill, rotating buffers      blocks %6d: %6.1f us  %.2f TB/s\n", blocks, t, mb / t);
        t = timeit([&] { hipLaunchKernelGGL(kD, dim3(blocks), dim3(256), 0, 0, outs[rot++ % 6], bits, n4); });
        printf("D nontemporal, rotating       blocks %6d: %6.1f us  %.2f TB/s\n", blocks, t, mb / t);
    }
    for (int gx : {257, 1013}) {
        float t = timeit([&] { hipLaunchKernelGGL(kC, dim3(gx, B), dim3(256), 0, 0, outs[rot++ % 6], bits, (unsigned)(px / 4)); });
        printf("C shipped pattern, rotating grid (%4d,%d): %6.1f us  %.2f TB/s\n", gx, B, t, mb / t);
    }
    for (int gx : {64, 128, 257, 507, 1013}) {
        float t = timeit([&] { hipLaunchKernelGGL(kC, dim3(gx, B), dim3(256), 0, 0, out, bits, (unsigned)(px / 4)); });
        printf("C shipped pattern grid (%4d,%d): %6.1f us  %.2f TB/s\n", gx, B, t, mb / t);
    }
    return 0;
}
