// Effective shader clock seen by a kernel of G workgroups: a chain of dependent v_fma_f32 (4 cycles each
// on CDNA) timed with the 100 MHz constant clock.   hipcc --offload-arch=gfx950 -O2 clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(float *out, unsigned long long *ticks, int iters) {
    float x = threadIdx.x * 1e-9f, a = 1.000001f, b = 1e-7f;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) x = __builtin_fmaf(x, a, b);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main() {
    float *out; unsigned long long *ticks;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&ticks, 4096 * 8);
    const int iters = 400;                       // 25 600 dependent FMAs
    for (int rep = 0; rep < 2; ++rep)
    for (int grid : {2, 8, 64, 256, 1024, 4096, 2}) {
        hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, out, ticks, iters);
        hipDeviceSynchronize();
        unsigned long long h[4096]; hipMemcpy(h, ticks, grid * 8, hipMemcpyDeviceToHost);
        double mx = 0, mn = 1e30; for (int i = 0; i < grid; ++i) { mx = h[i] > mx ? h[i] : mx; mn = h[i] < mn ? h[i] : mn; }
        const double n_fma = iters * 64.0;
        printf("grid %5d: %.1f - %.1f us per workgroup  -> %.2f - %.2f GHz if 4 cycles per dependent FMA\n", grid, mn / 100.0, mx / 100.0,
               n_fma * 4 / (mx * 10.0) , n_fma * 4 / (mn * 10.0));
    }
    return 0;
}
