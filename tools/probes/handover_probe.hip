// What would fusing the three middle exit-path launches (exit_tail_fwd -> route -> exit_tail_bwd) buy?  MEASURED with a model
// of them instead of argued: three dependent stages A (16 workgroups) -> B (2) -> C (16); every stage is a RAMP (a chain of
// dependent global loads: argument record, tables, first data -- ~3 us cold) followed by WORK (a timed spin: 4.6 / 6.8 /
// 6.1 us, the phase-trace figures of profiles/r02_phase_trace_exit_path.txt) and a small output the next stage reads.
//   form 1: three kernels, one hipGraph (what ships);
//   form 2: ONE kernel of 34 workgroups; a later stage does its ramp at once, then waits for a ticket the earlier stage's
//           workgroups increment behind a device-scope release fence, acquires, reads their outputs (checked), works.
// Prints the time per chain of both forms and the hand-over latency (last producer's end -> consumer's first use).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/handover_probe.hip -o /tmp/handover && /tmp/handover [dependent loads per ramp]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define NA 16
#define NB 2
#define NC 16
struct Args {
    const int *chase;            // pointer-chase table (the ramp)
    int hops;
    float *out_a, *out_b, *out_c;   // [N*][64]
    unsigned *ticket;            // [0] A done, [1] B done, [2] C done, [3] epoch
    unsigned long long *stamp;   // [stage][wg][2] start-of-work / end
    long work[3];                // ticks (100 MHz)
    int *bad;
};

__device__ __forceinline__ int ramp(const Args &a, int seed) {
    int p = seed;
    for (int h = 0; h < a.hops; ++h) p = __builtin_nontemporal_load(a.chase + p);      // dependent loads
    return p;
}
__device__ __forceinline__ void work(long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ void stage(const Args &a, int st, int wg, const float *in, int n_in, float *out, unsigned epoch, int r) {
    float acc = (float)r * 1e-30f;
    if (in) for (int i = threadIdx.x; i < n_in * 64; i += blockDim.x) {
        const float v = in[i];
        if (v != (float)epoch) atomicAdd(a.bad, 1);
        acc += v;
    }
    if (threadIdx.x == 0) a.stamp[(st * 16 + wg) * 2] = __builtin_amdgcn_s_memrealtime();
    work(a.work[st]);
    if (threadIdx.x < 64) out[wg * 64 + threadIdx.x] = (float)epoch + acc * 0.f;
    if (threadIdx.x == 0) a.stamp[(st * 16 + wg) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
}

__global__ __launch_bounds__(256) void stage_a(Args a) { const unsigned e = a.ticket[3]; stage(a, 0, blockIdx.x, nullptr, 0, a.out_a, e, ramp(a, blockIdx.x)); }
__global__ __launch_bounds__(256) void stage_b(Args a) { const unsigned e = a.ticket[3]; stage(a, 1, blockIdx.x, a.out_a, NA, a.out_b, e, ramp(a, 64 + blockIdx.x)); }
__global__ __launch_bounds__(256) void stage_c(Args a) {
    const unsigned e = a.ticket[3];
    stage(a, 2, blockIdx.x, a.out_b, NB, a.out_c, e, ramp(a, 128 + blockIdx.x));
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(a.ticket + 2, 1u) == NC - 1) { a.ticket[2] = 0; atomicAdd(a.ticket + 3, 1u); }
}

__device__ __forceinline__ void wait_for(unsigned *t, unsigned want) {
    if (threadIdx.x == 0) while (__hip_atomic_load(t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ __forceinline__ void done(unsigned *t) {
    __syncthreads();
    if (threadIdx.x == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __hip_atomic_fetch_add(t, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
}
__global__ __launch_bounds__(256) void fused(Args a) {
    const int b = blockIdx.x;
    const unsigned e = __hip_atomic_load(a.ticket + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b < NA) {
        stage(a, 0, b, nullptr, 0, a.out_a, e, ramp(a, b));
        done(a.ticket + 0);
    } else if (b < NA + NB) {
        const int r = ramp(a, 64 + b - NA);
        wait_for(a.ticket + 0, NA);
        stage(a, 1, b - NA, a.out_a, NA, a.out_b, e, r);
        done(a.ticket + 1);
    } else {
        const int r = ramp(a, 128 + b - NA - NB);
        wait_for(a.ticket + 1, NB);
        stage(a, 2, b - NA - NB, a.out_b, NB, a.out_c, e, r);
        __syncthreads();
        if (threadIdx.x == 0 && atomicAdd(a.ticket + 2, 1u) == NC - 1) {       // the last workgroup of the chain re-arms it
            a.ticket[0] = 0; a.ticket[1] = 0; a.ticket[2] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            atomicAdd(a.ticket + 3, 1u);
        }
    }
}

__global__ void flush_k(int *p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = (p[i] * 1 + 0); }

int main(int argc, char **argv) {
    Args a{};
    const int T = 1 << 16;
    std::vector<int> tab(T);
    for (int i = 0; i < T; ++i) tab[i] = (int)(((long)i * 40503 + 12345) % T);
    int *chase; (void)hipMalloc(&chase, T * 4); (void)hipMemcpy(chase, tab.data(), T * 4, hipMemcpyHostToDevice);
    a.chase = chase; a.hops = argc > 1 ? atoi(argv[1]) : 3;      // (dependent loads per ramp: ~0.7 us each, cold)
    (void)hipMalloc(&a.out_a, NA * 64 * 4); (void)hipMalloc(&a.out_b, NB * 64 * 4); (void)hipMalloc(&a.out_c, NC * 64 * 4);
    (void)hipMalloc(&a.ticket, 16); (void)hipMemset(a.ticket, 0, 16);
    (void)hipMalloc(&a.stamp, 3 * 16 * 2 * 8); (void)hipMalloc(&a.bad, 4); (void)hipMemset(a.bad, 0, 4);
    a.work[0] = 460; a.work[1] = 680; a.work[2] = 610;
    hipStream_t st; (void)hipStreamCreate(&st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int CH = 20, REP = 50;
    for (int form = 1; form <= 2; ++form) {
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int c = 0; c < CH; ++c) {
            hipLaunchKernelGGL(flush_k, dim3(256), dim3(256), 0, st, chase, T);      // (a predecessor that leaves the caches cold for the chain)
            if (form == 1) {
                hipLaunchKernelGGL(stage_a, dim3(NA), dim3(256), 0, st, a);
                hipLaunchKernelGGL(stage_b, dim3(NB), dim3(256), 0, st, a);
                hipLaunchKernelGGL(stage_c, dim3(NC), dim3(256), 0, st, a);
            } else {
                hipLaunchKernelGGL(fused, dim3(NA + NB + NC), dim3(256), 0, st, a);
            }
        }
        (void)hipStreamEndCapture(st, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        // the flush kernel alone, to subtract it
        hipGraph_t g0; hipGraphExec_t ge0;
        (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int c = 0; c < CH; ++c) hipLaunchKernelGGL(flush_k, dim3(256), dim3(256), 0, st, chase, T);
        (void)hipStreamEndCapture(st, &g0);
        (void)hipGraphInstantiate(&ge0, g0, nullptr, nullptr, 0);
        float ms = 0, ms0 = 0;
        for (int w = 0; w < 3; ++w) (void)hipGraphLaunch(ge, st);
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < REP; ++r) (void)hipGraphLaunch(ge, st);
        (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        for (int w = 0; w < 3; ++w) (void)hipGraphLaunch(ge0, st);
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < REP; ++r) (void)hipGraphLaunch(ge0, st);
        (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms0, e0, e1);
        std::vector<unsigned long long> s(3 * 16 * 2);
        (void)hipMemcpy(s.data(), a.stamp, s.size() * 8, hipMemcpyDeviceToHost);
        int bad = 0; (void)hipMemcpy(&bad, a.bad, 4, hipMemcpyDeviceToHost);
        unsigned long long a_end = 0, b_start = ~0ull, b_end = 0, c_start = ~0ull, a_start = ~0ull, c_end = 0;
        for (int i = 0; i < NA; ++i) { a_start = std::min(a_start, s[(0 * 16 + i) * 2]); a_end = std::max(a_end, s[(0 * 16 + i) * 2 + 1]); }
        for (int i = 0; i < NB; ++i) { b_start = std::min(b_start, s[(1 * 16 + i) * 2]); b_end = std::max(b_end, s[(1 * 16 + i) * 2 + 1]); }
        for (int i = 0; i < NC; ++i) { c_start = std::min(c_start, s[(2 * 16 + i) * 2]); c_end = std::max(c_end, s[(2 * 16 + i) * 2 + 1]); }
        printf("ramp of %d dependent loads, %s: %.2f us per chain (with the cold-cache predecessor %.2f, that kernel alone %.2f); A's last end -> B's first work %.2f us, "
               "B's end -> C's first work %.2f us; first work -> last end %.2f us (the three spins add to %.2f); stale reads %d\n",
               a.hops, form == 1 ? "three launches" : "one launch, tickets", (ms - ms0) * 1e3 / (CH * REP), ms * 1e3 / (CH * REP), ms0 * 1e3 / (CH * REP),
               (double)(long long)(b_start - a_end) * 0.01, (double)(long long)(c_start - b_end) * 0.01, (double)(long long)(c_end - a_start) * 0.01,
               (a.work[0] + a.work[1] + a.work[2]) * 0.01, bad);
    }
    return 0;
}
