// What does the conv bodies' unit loop cost on this machine, piece by piece?  A stand-alone model of conv_body's steady
// state (csrc/conv_kernel.h): per unit and wave 36 x v_mfma_f32_16x16x4_f32 fed by one ds_read_b128 per operand and tap
// from a double-buffered LDS tile, the staging of the next unit (BatchNorm + ReLU on two float4 items, five
// ds_write_b128), one LDS barrier.  Variants isolate what keeps the matrix pipe at 0.5-0.6 at saturation
// (profiles/r06_sq_saturated.txt):
//
//   0  MFMAs only, operands in registers, ONE accumulator (the dependent chain of the 16-channel tile)
//   1  the same with two accumulators (even / odd k-steps)
//   2  + the LDS fragment reads, software-pipelined one tap ahead (the shipped loop), no staging, no barrier
//   3  + one LDS barrier per unit
//   4  + the staging AFTER the MFMAs (the shipped order)
//   5  the staging INTERLEAVED with the MFMAs (one slice per tap, order pinned with sched_group_barrier)
//   6  as 5 with two accumulators
//   7  as 4, global loads of the next-but-one unit in flight (2 x 16 B per thread and unit, L2-resident)
//   8  as 5, global loads likewise
//
// Timed with s_memtime (shader cycles) and s_memrealtime (100 MHz): cycles per unit, the clock the chip holds, and
// matrix-pipe occupancy = waves per SIMD x 36 x 32 / cycles per unit.
//   hipcc --offload-arch=gfx950 -O3 mfma_loop_probe.hip -o mfma_loop_probe && ./mfma_loop_probe [units]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int P = 112, R = 18, BI = 576;           // plane stride, row stride (slots), weight items per chunk

__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
}

template <int V>
__global__ __launch_bounds__(256) void probe(float *out, unsigned long long *ticks, const f32x4 *src, int units, int xv) {
    __shared__ f32x4 tile[2][4 * P];
    __shared__ f32x4 wt[2][BI];
    __shared__ float cA[16 * 3];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, li = lane & 15;
    for (int i = tid; i < 2 * 4 * P; i += 256) (&tile[0][0])[i] = f32x4{0.001f * i, 0.5f, -0.25f, 1.f};
    for (int i = tid; i < 2 * BI; i += 256) (&wt[0][0])[i] = f32x4{0.002f * i, -0.5f, 0.25f, 1.f};
    if (tid < 48) cA[tid] = 0.01f * tid;
    __syncthreads();
    const int slot0 = wid * R + li;                  // M-tile `wid`: row wid of the 4 x 16 tile, pixel li
    // staging items of this thread: two tile slots (8 consecutive pixels of one plane in 8 lanes), three weight slots
    const int q = (tid >> 3) & 3;
    int xs[2];
    for (int k = 0; k < 2; ++k) { const int i = tid + k * 256, hp = ((i >> 5) << 3) + (i & 7); xs[k] = q * P + (hp < 108 ? hp : 108 + (tid & 3)); }
    f32x4 xr[2] = {f32x4{0.1f * tid, 0.2f, 0.3f, 0.4f}, f32x4{0.5f, 0.6f * tid, 0.7f, 0.8f}};
    f32x4 br[3] = {f32x4{1.f, 2.f, 3.f, 4.f}, f32x4{5.f, 6.f, 7.f, 8.f}, f32x4{9.f, 1.f, 2.f, 3.f}};
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 ra = {0.1f * lane, 0.2f, 0.3f, 0.4f}, rb = {0.5f, 0.25f * lane, 0.125f, 1.f};
    constexpr bool LDS = V >= 2, BAR = V >= 3, STAGE = V >= 4, INTER = V == 5 || V == 6 || V == 8, TWO = V == 1 || V == 6, GL = V == 7 || V == 8 || V == 10;
    constexpr bool FIRST = V == 9 || V == 10;      // the next unit's staging BEFORE this unit's MFMAs: the LDS writes complete under them
    const f32x4 *gsrc = src + (size_t)blockIdx.x * 512 + tid;
    f32x4 nx[2] = {xr[0], xr[1]};

    // one staging slice: item k's transform and store (k = 0, 1), weight item k - 2 (k = 2 .. 4)
    auto stage_slice = [&](int k, int buf) {
        if (k < 2) {
            float cc[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 3; ++e) cc[j][e] = cA[(q * 4 + j) * 3 + e];
            f32x4 v = xr[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf((v[j] - cc[j][0]) * cc[j][1] + cc[j][2], 0.f);
            tile[buf][xs[k]] = v;
        } else {
            const int i = tid + (k - 2) * 256;
            if (i < BI) wt[buf][i] = br[k - 2];
        }
    };

    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int u = 0; u < units; ++u) {
        const f32x4 *cur = tile[u & 1], *wl = wt[u & 1];
        if (GL) { nx[0] = gsrc[(u & 7) * 65536]; nx[1] = gsrc[(u & 7) * 65536 + 256]; }
        if (STAGE && FIRST) {
#pragma unroll
            for (int k = 0; k < 5; ++k) stage_slice(k, (u + 1) & 1);
        }
        if (!LDS) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (TWO && (j & 1)) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[j], rb[j], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[j], rb[j], acc0, 0, 0, 0);
                }
        } else {
            f32x4 fa[2], fb[2];
            auto frag = [&](int tap, f32x4 &a, f32x4 &b) {
                const int dy = tap / 3, dx = tap - dy * 3;
                b = wl[(tap * 4 + g) * 16 + li];
                a = cur[g * P + slot0 + dy * R + dx];
            };
            frag(0, fa[0], fb[0]);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) frag(tap + 1, fa[(tap + 1) & 1], fb[(tap + 1) & 1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (TWO && (j & 1)) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[tap & 1][j], fb[tap & 1][j], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[tap & 1][j], fb[tap & 1][j], acc0, 0, 0, 0);
                }
                if (INTER && tap < 5) stage_slice(tap, (u + 1) & 1);
                if (tap + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                if (INTER && tap < 5) {
                    // MFMA, a few VALU, MFMA, ... : the slice's vector work sits in the MFMAs' shadows
                    if (tap < 2) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x102, 5, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x202, 5, 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x202, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    }
                } else
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }
        asm volatile("s_nop 15");
        if (STAGE && !INTER && !FIRST) {
#pragma unroll
            for (int k = 0; k < 5; ++k) stage_slice(k, (u + 1) & 1);
        }
        if (GL) { xr[0] = nx[0]; xr[1] = nx[1]; }
        else { xr[0][0] += 1.f; xr[1][1] += 1.f; }
        // `xv` further vector instructions per unit (four independent chains): the real bodies execute 100-200 per unit
        // (address arithmetic, masks, the tile epilogue) where this model's staging has 31
        for (int i = 0; i < xv; i += 4) {
            xr[0][2] = __builtin_fmaf(xr[0][2], 1.0001f, 0.5f); xr[0][3] = __builtin_fmaf(xr[0][3], 0.9999f, 0.25f);
            xr[1][2] = __builtin_fmaf(xr[1][2], 1.0002f, 0.125f); xr[1][3] = __builtin_fmaf(xr[1][3], 0.9998f, 0.75f);
        }
        if (BAR) lds_barrier();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { ticks[blockIdx.x * 2] = c1 - c0; ticks[blockIdx.x * 2 + 1] = r1 - r0; }
    const f32x4 a = acc0 + acc1;
    out[(size_t)blockIdx.x * 256 + tid] = a[0] + a[1] + a[2] + a[3] + xr[0][2];
}

template <int V> static void run(int units, int wpc, float *out, unsigned long long *ticks, const f32x4 *src, int xv = 0) {
    const int grid = 256 * wpc;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe<V>, dim3(grid), dim3(256), 0, 0, out, ticks, src, units, xv);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(grid * 2);
    hipMemcpy(h.data(), ticks, grid * 16, hipMemcpyDeviceToHost);
    std::vector<double> cyc, ghz;
    for (int i = 0; i < grid; ++i) { cyc.push_back((double)h[2 * i] / units); ghz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] * 10.0)); }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    const double c = cyc[grid / 2], f = ghz[grid / 2];
    printf("variant %d xv %3d  %d WG/CU: %7.0f cycles per unit (median; max %7.0f), clock %.2f GHz, pipe %.3f, %.1f TFLOP/s\n", V, xv, wpc, c, cyc.back(), f,
           wpc * 36 * 32.0 / c, 256.0 * wpc * 4 * 36 * 2048.0 / (c / (f * 1e9)) / 1e12);
}

int main(int argc, char **argv) {
    const int units = argc > 1 ? atoi(argv[1]) : 400;
    float *out; unsigned long long *ticks; f32x4 *src;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&ticks, 1024 * 16); hipMalloc(&src, (size_t)(1024 * 512 + 8 * 65536 + 512) * 16);
    hipMemset(src, 0, (size_t)(1024 * 512 + 8 * 65536 + 512) * 16);
    for (int wpc = 1; wpc <= 3; ++wpc) {
        run<0>(units, wpc, out, ticks, src); run<1>(units, wpc, out, ticks, src); run<2>(units, wpc, out, ticks, src);
        run<3>(units, wpc, out, ticks, src); run<4>(units, wpc, out, ticks, src); run<5>(units, wpc, out, ticks, src);
        run<6>(units, wpc, out, ticks, src); run<7>(units, wpc, out, ticks, src); run<8>(units, wpc, out, ticks, src);
        run<9>(units, wpc, out, ticks, src); run<10>(units, wpc, out, ticks, src);
        for (int xv : {40, 80, 120, 160, 240}) { run<4>(units, wpc, out, ticks, src, xv); run<7>(units, wpc, out, ticks, src, xv); run<10>(units, wpc, out, ticks, src, xv); }
    }
    return 0;
}
