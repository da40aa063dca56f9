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
