// Which XCD does workgroup b of a 1-D grid run on?  (HW_REG_XCC_ID, gfx950)  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/probes/xcc_probe.hip -o /tmp/xcc_probe && /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int *out) {
    if (threadIdx.x == 0) {
        const int xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 15;     // hwreg(HW_REG_XCC_ID, 0, 4)
        const int hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));          // HW_REG_HW_ID
        out[blockIdx.x * 2] = xcc;
        out[blockIdx.x * 2 + 1] = hw;
    }
    __builtin_amdgcn_s_sleep(64);
}
int main() {
    for (int n : {64, 512, 768, 1024}) {
        for (int threads : {256, 512}) {
            int *d; hipMalloc(&d, n * 8);
            hipLaunchKernelGGL(probe, dim3(n), dim3(threads), 0, 0, d);
            std::vector<int> h(n * 2); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
            int match = 0; for (int b = 0; b < n; ++b) match += h[b * 2] == (b % 8);
            int same0 = 0; for (int b = 0; b < n; ++b) match += 0, same0 += ((h[b * 2] - h[0] + 8) % 8) == (b % 8);
            printf("grid %4d x %3d threads: xcc == b %% 8 for %d of %d blocks; (xcc - xcc[0]) %% 8 == b %% 8 for %d; first 24:", n, threads, match, n, same0);
            for (int b = 0; b < 24; ++b) printf(" %d", h[b * 2]);
            printf("\n");
            hipFree(d);
        }
    }
    return 0;
}
