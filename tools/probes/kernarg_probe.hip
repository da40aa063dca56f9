// What does reading a LARGE by-value kernel argument cost at the start of a kernel?  The backward launches take their
// parameter record (three conv records, ~1.3 KB) by value; the compiler fetches the fields with scalar loads close
// to their first use, i.e. in several DEPENDENT batches, each a scalar-cache miss on a kernel-argument segment that
// was written by the command processor just before the launch.
//   by_value<CHAIN>: a 1280-byte struct by value; thread 0 of every workgroup walks CHAIN fields in different 64-byte
//                    lines, each index depending on the previous value (nothing can be batched), and stores the
//                    100 MHz clock difference;
//   by_pointer<CHAIN>: the same walk through a pointer to a copy in device memory.
// Launched eagerly and as a hipGraph replay (the product's path).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/kernarg_probe.hip -o /tmp/kernarg_probe && /tmp/kernarg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
struct Big { int v[320]; };
template <int CHAIN>
__global__ __launch_bounds__(256) void by_value(const Big p, unsigned long long *out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int i = 20;
#pragma unroll
    for (int k = 0; k < CHAIN; ++k) i = p.v[i] + 16;            // host: v[j] = j + 48 -> next index one KB-quarter further
    asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(i) : "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) + (unsigned long long)(i == 12345);
}
template <int CHAIN>
__global__ __launch_bounds__(256) void by_pointer(const Big *__restrict__ p, unsigned long long *out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int i = 20;
#pragma unroll
    for (int k = 0; k < CHAIN; ++k) i = p->v[i] + 16;
    asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(i) : "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) + (unsigned long long)(i == 12345);
}
static void report(const char *what, unsigned long long *d_out, int n) {
    std::vector<unsigned long long> h(n);
    (void)hipMemcpy(h.data(), d_out, n * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("  %-34s min %5.2f  median %5.2f  max %5.2f us\n", what, h[0] / 100.0, h[n / 2] / 100.0, h[n - 1] / 100.0);
}
int main() {
    const int n = 512;
    Big hb;
    for (int j = 0; j < 320; ++j) hb.v[j] = (j + 48) % 300;
    Big *db; (void)hipMalloc(&db, sizeof(Big)); (void)hipMemcpy(db, &hb, sizeof(Big), hipMemcpyHostToDevice);
    unsigned long long *out; (void)hipMalloc(&out, n * 8);
    hipStream_t st; (void)hipStreamCreate(&st);
#define RUN(K, CH) do { \
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((K<CH>), dim3(n), dim3(256), 0, st, ARG, out); \
        (void)hipStreamSynchronize(st); report(#K " eager, chain " #CH, out, n); \
        hipGraph_t g; hipGraphExec_t ge; \
        (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal); \
        hipLaunchKernelGGL((K<CH>), dim3(n), dim3(256), 0, st, ARG, out); \
        (void)hipStreamEndCapture(st, &g); (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0); \
        for (int r = 0; r < 3; ++r) (void)hipGraphLaunch(ge, st); \
        (void)hipStreamSynchronize(st); report(#K " graph, chain " #CH, out, n); \
    } while (0)
#define ARG hb
    RUN(by_value, 1); RUN(by_value, 3); RUN(by_value, 5);
#undef ARG
#define ARG db
    RUN(by_pointer, 1); RUN(by_pointer, 3); RUN(by_pointer, 5);
    return 0;
}
