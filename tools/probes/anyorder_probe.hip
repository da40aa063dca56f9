// Does hipExtAnyOrderLaunch let a kernel start while its predecessor IN THE SAME STREAM is still running on gfx950?
// A: 128 workgroups spin 20 us and stamp their end; B: 128 workgroups stamp their start (100 MHz constant clock).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/anyorder_probe.hip -o /tmp/anyorder && /tmp/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <algorithm>
#include <vector>
__global__ void spin_k(unsigned long long *t, long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = t0; t[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}
__global__ void stamp_k(unsigned long long *t) {
    if (threadIdx.x == 0) t[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}
int main() {
    unsigned long long *ta, *tb;
    (void)hipMalloc(&ta, 128 * 2 * 8); (void)hipMalloc(&tb, 128 * 8);
    hipStream_t st; (void)hipStreamCreate(&st);
    for (int flags : {0, 1}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipExtLaunchKernelGGL(spin_k, dim3(128), dim3(256), 0, st, nullptr, nullptr, 0, ta, 2000L);
            hipExtLaunchKernelGGL(stamp_k, dim3(128), dim3(256), 0, st, nullptr, nullptr, flags, tb);
            (void)hipStreamSynchronize(st);
        }
        std::vector<unsigned long long> a(256), b(128);
        (void)hipMemcpy(a.data(), ta, 256 * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), tb, 128 * 8, hipMemcpyDeviceToHost);
        unsigned long long a_start = ~0ull, a_end = 0, b_start = ~0ull;
        for (int i = 0; i < 128; ++i) { a_start = std::min(a_start, a[2 * i]); a_end = std::max(a_end, a[2 * i + 1]); b_start = std::min(b_start, b[i]); }
        printf("flags %d: A ran %.2f us; B's first workgroup started %.2f us after A's START (%.2f us %s A's end)\n", flags,
               (a_end - a_start) * 0.01, (double)(long long)(b_start - a_start) * 0.01, fabs((double)(long long)(b_start - a_end)) * 0.01,
               b_start < a_end ? "BEFORE" : "after");
    }
    return 0;
}
