// Shared device helpers for the multipath-nn MI355X (gfx950) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "mpnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Minimum waves per SIMD requested from the register allocator for the grouped (latency-bound)
// kernels: 4 workgroups of 256 threads per CU.
#ifndef MPNN_OCC
#define MPNN_OCC 4
#endif

#define MPNN_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) return (int)e_; } while (0)

// ---------------------------------------------------------------------------
// BatchNorm coefficients for one channel (reference: layer_types.py:231-238).
// Batch mode: mean/biased variance from the fp64 sums the producing conv
// accumulated; moving mode: m_avg / v_avg.
// ---------------------------------------------------------------------------
struct BnC { float m, rstd, gamma, beta; };

// Sum of one statistic over the replicated slots.
__device__ __forceinline__ double slot_sum(const double *base, int C2, int idx, int nslot) {
    double t = 0.0;
#pragma unroll 4
    for (int s = 0; s < nslot; ++s) t += base[s * C2 + idx];
    return t;
}

__device__ __forceinline__ BnC bn_coef(const mpnn_act &b, int c) {
    BnC k;
    k.gamma = b.gamma[c];
    k.beta = b.beta[c];
    if (b.mode == MPNN_ACT_BN_BATCH) {
        const double inv = 1.0 / (double)b.cnt;
        const double mean = slot_sum(b.sum, 2 * b.C, c, b.nslot) * inv;
        double var = slot_sum(b.sum, 2 * b.C, b.C + c, b.nslot) * inv - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        k.m = (float)mean;
        k.rstd = rsqrtf((float)var + b.eps);
    } else {
        k.m = b.m_avg[c];
        k.rstd = rsqrtf(b.v_avg[c] + b.eps);
    }
    return k;
}

// Workgroup barrier that orders LDS traffic only: waits for this wave's LDS operations
// (lgkmcnt(0)), NOT for its global loads.  __syncthreads() also drains vmcnt, which would wait for
// the prefetch loads that are deliberately left in flight across pipeline steps.
__device__ __forceinline__ void lds_barrier() {
#ifdef MPNN_SAFE_BARRIER
    __syncthreads();
    return;
#endif
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0) only
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Drain the MFMA pipe before the accumulators are read by VALU code.  hipcc covers the
// MFMA-write -> VALU-read hazard by counting the instructions in between as wait states; when
// those are mostly SALU instructions (branch conditions of the persistent loop) the last
// accumulator register written by the last MFMA was observed to be read STALE on gfx950
// (fwd_group_k: rows 4g+3 of a tile missed the final MFMA, run-to-run nondeterministic).
// One s_nop 15 (16 wait states) after every unit's MFMA block, against >1000 cycles of MFMAs per unit.
__device__ __forceinline__ void mfma_drain() {
    __builtin_amdgcn_sched_barrier(0);
#ifndef MPNN_DRAIN_NOPS
#define MPNN_DRAIN_NOPS 1
#endif
#pragma unroll
    for (int k = 0; k < MPNN_DRAIN_NOPS; ++k) asm volatile("s_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// fp64 butterfly over the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48).
__device__ __forceinline__ double reduce_g4(double v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
