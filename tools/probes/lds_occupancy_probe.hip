// How many 256-thread workgroups with L bytes of (dynamic) LDS does a CU of this chip really host at once?  A grid of 256 x k
// workgroups that each spin for T us takes T if all are resident, 2 T if a round has to queue.  (The occupancy API and the
// 160 KB / L arithmetic both say four for the conv bodies' 40 768 bytes; the traces show three.)
//   hipcc --offload-arch=gfx950 -O3 lds_occupancy_probe.hip -o lds_occupancy_probe && ./lds_occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void spin(float *out, int ticks) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) {}
    if (threadIdx.x == 0) out[blockIdx.x] = lds[(threadIdx.x + 1) & 255];
}
int main() {
    float *out; hipMalloc(&out, 4096 * 4);
    hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ticks = 5000;                                 // 50 us
    for (int k = 2; k <= 6; ++k) {
        printf("%d workgroups per CU:", k);
        for (int L : {24576, 27648, 30720, 31744, 32768, 33792, 36864, 38912, 39936, 40768, 40960, 41984, 45056, 53248, 54272}) {
            int api = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, (const void *)spin, 256, L);
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(spin, dim3(256 * k), dim3(256), L, 0, out, ticks);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            printf("  %dB:%s(api %d)", L, best < 0.08f ? "ok" : "QUEUED", api);
        }
        printf("\n");
    }
    return 0;
}
