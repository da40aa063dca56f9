// Does COLD instruction fetch limit a kernel's start?  Every launch starts with invalidated instruction caches; a
// kernel that executes N straight-line instructions once has to fetch all of them.  `straight<N>`: N dependent
// v_fma with distinct constants (nothing to roll back into a loop, ~12 bytes of code each).  If fetch kept up, the
// time is N x the dependent-issue time of a v_fma (~4 ns); if it did not, the time per instruction would grow with
// the code size.  Measured on MI355X (256 workgroups x 256 threads): 256 / 1024 / 2048 / 4096 instructions =
// 2.4 / 5.1 / 8.7 / 15.7 us, an empty kernel 2.5 us: linear, ~3.3 ns per instruction up to 48 KB of code.
//   hipcc --offload-arch=gfx950 -O3 -ftemplate-depth=8192 tools/probes/icache_probe.hip -o /tmp/icache_probe && /tmp/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Unroll {
    static __device__ __forceinline__ float run(float x, float a) {
        x = __builtin_fmaf(x, a, (float)N * 1.0001f);
        return Unroll<N - 1>::run(x, a);
    }
};
template <> struct Unroll<0> { static __device__ __forceinline__ float run(float x, float) { return x; } };
template <int N>
__global__ __launch_bounds__(256) void straight(float *out, float a) {
    float x = threadIdx.x;
    x = Unroll<N>::run(x, a);
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
template <typename F> float time_us(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) f();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 200 * 1e3f;
}
int main() {
    float *d; (void)hipMalloc(&d, 1024 * 256 * 4);
    for (int wgs : {256, 1024}) {
        printf("grid %d x 256:\n", wgs);
#define CASE(N) printf("  %5d dependent fma, straight-line (~%2d KB of code): %6.2f us\n", N, N * 12 / 1024, \
                       time_us([&] { hipLaunchKernelGGL((straight<N>), dim3(wgs), dim3(256), 0, 0, d, 1.0001f); }));
        CASE(1) CASE(256) CASE(1024) CASE(2048) CASE(4096)
    }
    return 0;
}
