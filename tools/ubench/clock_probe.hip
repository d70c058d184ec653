// One wave spins for ~2 ms and reports shader-clock ticks (s_memtime) per 100 MHz wall tick (s_memrealtime): the clock the chip
// holds while something else (e.g. the FCN forward pass on another stream) is running.   hipcc --offload-arch=gfx950 -shared -fPIC
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ void lm_clock_probe(unsigned long long* out, unsigned long long wall_ticks)
{
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned long long w = w0;
    while (w - w0 < wall_ticks) { __builtin_amdgcn_s_sleep(32); w = wall_clock64(); }
    out[0] = clock64() - c0;
    out[1] = w - w0;
}
extern "C" int lm_clock_probe_launch(unsigned long long* d_out, unsigned long long wall_ticks, void* stream)
{
    hipLaunchKernelGGL(lm_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, d_out, wall_ticks);
    return (int)hipGetLastError();
}
