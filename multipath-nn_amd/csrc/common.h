// Shared device helpers for the multipath-nn MI355X (gfx950) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "mpnn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Minimum waves per SIMD requested from the register allocator for the grouped (latency-bound)
// kernels: 4 workgroups of 256 threads per CU.
#ifndef MPNN_OCC
#define MPNN_OCC 4
#endif

#define MPNN_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) return (int)e_; } while (0)

// Workgroups of `kernel` (256 threads, `dyn_lds` bytes of dynamic LDS) that are resident on the
// device at once (occupancy x compute units), cached per kernel instantiation.  The persistent grids
// are sized to this: workgroups beyond it only start when earlier ones exit, which serialises the
// bodies of a fused launch (measured: the weight-gradient workgroups of bwd_scale started 8-13 us late).
// Compute units the persistent grids leave free (mpnn_set_reserved_cus, misc.hip): under data parallelism RCCL's
// workgroups run beside the backward launches, and a grid fitted to EVERY resident slot then has workgroups that only
// start once others exit -- which doubles a launch whose workgroups all finish together.
extern int mpnn_reserved_cus_g;

static int resident_slots(const void *kernel, int dyn_lds, int threads = 256, int max_per_cu = 0) {
    struct Entry { const void *fn; int lds, per_cu, cus, cap; };      // (a kernel is always queried with the same block size)
    static Entry cache[64];
    static int n_cached = 0;
    int per_cu = 0, cus = 256;
    bool hit = false;
    for (int i = 0; i < n_cached && !hit; ++i)
        if (cache[i].fn == kernel && cache[i].lds == dyn_lds && cache[i].cap == max_per_cu) {
            per_cu = cache[i].per_cu; cus = cache[i].cus; hit = true;
        }
    if (!hit) {
        int dev = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, (size_t)dyn_lds) != hipSuccess || per_cu < 1)
            per_cu = 2;
        // The occupancy query was seen to allow two 82 KB workgroups on a 160 KB CU (bwd_scale_k<2,4,2>:
        // half of the grid then started 11-23 us late): bound it by the LDS arithmetic as well.
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, kernel) == hipSuccess) {
            const size_t lds = fa.sharedSizeBytes + (size_t)dyn_lds;
            if (lds > 0) {
                const int by_lds = (int)((160u * 1024u) / ((lds + 1279) / 1280 * 1280));     // 1280-byte allocation granules
                if (by_lds >= 1 && by_lds < per_cu) per_cu = by_lds;
            }
        }
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        }
        if (max_per_cu > 0 && per_cu > max_per_cu) per_cu = max_per_cu;      // (the caller wants fewer, larger shares)
        {   // experiment: MPNN_SLOTS_PER_CU=<n> overrides the answer (does the chip really host n workgroups of this kernel?)
            static const int force = [] { const char *e = getenv("MPNN_SLOTS_PER_CU"); return e ? atoi(e) : 0; }();
            if (force > 0) per_cu = force;
        }
        if (n_cached < 64) cache[n_cached++] = Entry{kernel, dyn_lds, per_cu, cus, max_per_cu};
    }
    const int free_cus = cus - mpnn_reserved_cus_g;
    return per_cu * (free_cus < 8 ? 8 : free_cus);
}

// ---------------------------------------------------------------------------
// Phase trace (profiling aid; off unless mpnn_debug_set_trace() installed a buffer).  Thread 0 of a
// workgroup stamps the 100 MHz constant clock into buf[wg * MPNN_TRACE_SLOTS + k]; the host tool
// (tools/trace_phases.py) turns the stamps into a per-workgroup timeline of a single launch.
// Every translation unit has its own copy of the pointer (no relocatable device code).
// ---------------------------------------------------------------------------
// Compiled in only with -DMPNN_TRACE (make trace -> libmpnn_hip_trace.so): the stamps cost registers.
#define MPNN_TRACE_SLOTS 12
#ifdef MPNN_TRACE
static __device__ unsigned long long *mpnn_trace_buf_ = nullptr;
__device__ __forceinline__ void trace_stamp(int k) {
    unsigned long long *b = mpnn_trace_buf_;
    if (b && threadIdx.x == 0)
        b[(size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * MPNN_TRACE_SLOTS + k] = __builtin_amdgcn_s_memrealtime();
}
__device__ __forceinline__ void trace_note(int k, unsigned long long v) {
    unsigned long long *b = mpnn_trace_buf_;
    if (b && threadIdx.x == 0) b[(size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * MPNN_TRACE_SLOTS + k] = v;
}
static inline int mpnn_trace_install(void *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(mpnn_trace_buf_), &buf, sizeof(buf));
}
#else
__device__ __forceinline__ void trace_stamp(int) {}
__device__ __forceinline__ void trace_note(int, unsigned long long) {}
static inline int mpnn_trace_install(void *) { return MPNN_E_ARG; }
#endif

// ---------------------------------------------------------------------------
// BatchNorm coefficients for one channel (reference: layer_types.py:231-238).
// Batch mode: mean/biased variance from the fp64 sums the producing conv
// accumulated; moving mode: m_avg / v_avg.
// ---------------------------------------------------------------------------
struct BnC { float m, rstd, gamma, beta; };

// Sums of TWO statistics over the replicated slots in one loop: the loads of both are in flight
// together, 8 slots per round trip (16 slots = 2 dependent round trips instead of 8).
// (Addressing: a UNIFORM row pointer per slot plus the lane's 32-bit byte offset -- the load takes the scalar-base +
// vector-offset form and costs no vector arithmetic; `base[s * C2 + i]` with 64-bit index arithmetic was six
// instructions per load, and this code is the first thing every workgroup of a launch executes.)
__device__ __forceinline__ double ld_slot(const double *base, int C2, int s, unsigned off) {
    const char *row = (const char *)(base + (size_t)s * C2);             // uniform
    return *(const double *)(row + off);
}
__device__ __forceinline__ void slot_sum2(const double *base, int C2, int i0, int i1, int nslot, double &a, double &b) {
    double t0 = 0.0, t1 = 0.0;
    const unsigned o0 = (unsigned)i0 * 8u, o1 = (unsigned)i1 * 8u;
#pragma unroll 8
    for (int s = 0; s < nslot; ++s) { t0 += ld_slot(base, C2, s, o0); t1 += ld_slot(base, C2, s, o1); }
    a = t0; b = t1;
}
__device__ __forceinline__ double slot_sum(const double *base, int C2, int idx, int nslot) {
    double t = 0.0;
    const unsigned o = (unsigned)idx * 8u;
#pragma unroll 8
    for (int s = 0; s < nslot; ++s) t += ld_slot(base, C2, s, o);
    return t;
}

__device__ __forceinline__ BnC bn_coef(const mpnn_act &b, int c) {
    BnC k;
    k.gamma = b.mode == MPNN_ACT_RELU ? 1.f : b.gamma[c];
    k.beta = b.mode == MPNN_ACT_RELU ? 0.f : b.beta[c];
    if (b.mode == MPNN_ACT_BN_BATCH) {
        const double inv = 1.0 / (double)b.cnt;
        double s1, s2;
        slot_sum2(b.sum, 2 * b.C, c, b.C + c, b.nslot, s1, s2);
        const double mean = s1 * inv;
        double var = s2 * inv - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        k.m = (float)mean;
        k.rstd = rsqrtf((float)var + b.eps);
    } else if (b.mode == MPNN_ACT_RELU) {          // plain ReLU: identity coefficients (gamma / beta are not read)
        k.m = 0.f;  k.rstd = 1.f;  k.gamma = 1.f;  k.beta = 0.f;
    } else {
        k.m = b.m_avg[c];
        k.rstd = rsqrtf(b.v_avg[c] + b.eps);
    }
    return k;
}

// The five BatchNorm-backward coefficients of one channel -- mean, rstd, gamma * rstd and the two backward reductions
// scaled by 1 / count (or beta, 0 with want_beta) -- with EVERY input requested before the first is used: gamma, beta,
// the slot sums and the slot reductions are ONE memory round trip (bn_coef followed by slot_sum2 was two to three
// dependent ones, and the coefficient tables are the first thing every backward workgroup waits for).
// Same summation order as bn_coef / slot_sum2 (slots ascending from 0.0): identical bits.  Batch-statistics mode only
// (the callers fall back to bn_coef otherwise).
// (Splitting this into a request half and a finish half with the first tile's loads issued in between -- so that the
// tables cost no round trip of their own -- was measured SLOWER, 493 -> 498 us per step: 66 more live registers in the
// prologue, spills in the 64-channel weight-gradient variants, and the prologue is bound by its instruction count, not
// by the round trip.)
__device__ __forceinline__ void bn_bwd_row(const mpnn_act &b, const double *red, int red_nslot, int c, bool want_beta, float *e) {
    const int C2 = 2 * b.C, ns = b.nslot, rn = red ? red_nslot : 0;
    const float gamma = b.gamma[c], beta = b.beta[c];
    const unsigned o0 = (unsigned)c * 8u, o1 = (unsigned)(b.C + c) * 8u;
    double s1 = 0.0, s2 = 0.0, r0 = 0.0, r1 = 0.0;
    if (ns == 8 && (rn == 8 || rn == 0)) {             // (uniform; what the engine uses: nothing predicated, ~100 instructions)
        double a0[8], a1[8], q0[8], q1[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) { a0[s] = ld_slot(b.sum, C2, s, o0); a1[s] = ld_slot(b.sum, C2, s, o1); }
        if (rn) {
#pragma unroll
            for (int s = 0; s < 8; ++s) { q0[s] = ld_slot(red, C2, s, o0); q1[s] = ld_slot(red, C2, s, o1); }
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) { q0[s] = 0.0; q1[s] = 0.0; }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) { s1 += a0[s]; s2 += a1[s]; }
        if (rn) {
#pragma unroll
            for (int s = 0; s < 8; ++s) { r0 += q0[s]; r1 += q1[s]; }
        }
    } else {
        slot_sum2(b.sum, C2, c, b.C + c, ns, s1, s2);
        if (rn) slot_sum2(red, C2, c, b.C + c, rn, r0, r1);
    }
    const double inv = 1.0 / (double)b.cnt;
    const double mean = s1 * inv;
    double var = s2 * inv - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = rsqrtf((float)var + b.eps);
    e[0] = (float)mean; e[1] = rstd; e[2] = gamma * rstd;
    if (want_beta) { e[3] = beta; e[4] = 0.f; }
    else { e[3] = (float)(r0 * inv); e[4] = (float)(r1 * inv); }
}

// One BatchNorm of mpnn_bn_finalize (table record t: see misc.hip): moving averages from the forward
// sums, dgamma / dbeta from the backward reductions.  Shared by bn_finalize_k and backward_finish_k.
struct BnNoHook { __device__ __forceinline__ void operator()(int, float, int, float) const {} };
// on_grad(beta offset, dbeta, gamma offset, dgamma): called for every channel whose gradients were written (the fused
// end of the backward pass applies the parameter update there)
template <class F = BnNoHook>
__device__ __forceinline__ void bn_finalize_body(double *__restrict__ sums, double *__restrict__ reds,
                                                 float *__restrict__ state, float *__restrict__ grads,
                                                 const int *__restrict__ t, float decay, int n_img,
                                                 double *__restrict__ sums_keep, F on_grad = F()) {
    const int C = t[3], ns = t[7];
    const double inv = 1.0 / ((double)t[4] * (double)n_img);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double mean = slot_sum(sums + t[0], 2 * C, c, ns) * inv;
        double var = slot_sum(sums + t[0], 2 * C, C + c, ns) * inv - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        // A BatchNorm whose output nobody consumes (t[5] < 0: a scale the child block drops and no exit
        // reads) keeps its moving averages: in the reference the two tf.assign hang off the OUTPUT by
        // control dependency (layer_types.py:233-236) and TensorFlow never executes an unfetched output.
        if (t[5] >= 0) {
            float *m = state + t[1] + c, *v = state + t[2] + c;
            *m = decay * *m + (1.f - decay) * (float)mean;
            *v = decay * *v + (1.f - decay) * (float)var;
        }
        if (reds && grads && t[5] >= 0) {
            const float dbeta = (float)slot_sum(reds + t[0], 2 * C, c, ns);            // dbeta  = sum dz
            const float dgamma = (float)slot_sum(reds + t[0], 2 * C, C + c, ns);       // dgamma = sum dz * xhat
            grads[t[6] + c] = dbeta;
            grads[t[5] + c] = dgamma;
            on_grad(t[6] + c, dbeta, t[5] + c, dgamma);
        }
        // This is the LAST reader of the step's slot sums: with sums_keep set it leaves them cleared for the next
        // step (the forward convs and the backward epilogues add to them with atomics), so that a training step
        // needs no clearing launch of its own.
        // (sums_keep: a copy of the forward sums for whoever wants to look at the step's batch statistics afterwards)
        if (!sums_keep) continue;
        for (int s = 0; s < ns; ++s) {
            const int i0 = t[0] + s * 2 * C + c, i1 = i0 + C;
            sums_keep[i0] = sums[i0];  sums_keep[i1] = sums[i1];
            sums[i0] = 0.0;  sums[i1] = 0.0;
            if (reds) { reds[i0] = 0.0;  reds[i1] = 0.0; }
        }
    }
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
// Wave priority around a unit's MFMA block (build-time experiment, -DMPNN_MFMA_PRIO=<1..3>): the waves of different
// workgroups that share a SIMD run the same program and fall into step -- all in their MFMA block, then all in their
// staging code, the matrix pipe idle meanwhile (profiles/r06_sq_saturated.txt).  With the MFMA block at a raised
// priority the wave that reaches it first keeps the pipe until its unit is done and its partners' vector work fills
// the issue slots between its MFMAs.
#ifndef MPNN_MFMA_PRIO
#define MPNN_MFMA_PRIO 0
#endif
__device__ __forceinline__ void mfma_prio_on() {
#if MPNN_MFMA_PRIO
    __builtin_amdgcn_s_setprio(MPNN_MFMA_PRIO);
#endif
}
__device__ __forceinline__ void mfma_prio_off() {
#if MPNN_MFMA_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}
__device__ __forceinline__ void mfma_drain() {
    __builtin_amdgcn_sched_barrier(0);
#ifndef MPNN_DRAIN_NOPS
#define MPNN_DRAIN_NOPS 1
#endif
#pragma unroll
    for (int k = 0; k < MPNN_DRAIN_NOPS; ++k) asm volatile("s_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// conv_first.hip: launches the record if it is the first conv of a net (0), or declines (1)
int mpnn_first_conv_launch(const mpnn_conv_fwd_args *a, hipStream_t st);

// XCD-aware tile order of the conv bodies (ConvP::xcd): on unless MPNN_XCD=0 (A/B measurements).
static inline int xcd_env() {
    static const int v = [] { const char *e = getenv("MPNN_XCD"); return e ? atoi(e) : 1; }();
    return v;
}
// workgroups per row for an XCD-aware launch: a multiple of 8 once there are at least 16
static inline int xcd_round(int g) { return (xcd_env() && g >= 16) ? (g & ~7) : g; }

// fp64 butterfly over the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48).
__device__ __forceinline__ double reduce_g4(double v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// Wave-wide sums, result in every lane.  Inside a 16-lane row the butterfly runs on DPP operands (quad
// swaps, then the half-row and row mirrors: ALU latency), the four row sums are then read as scalars.
// (The __shfl_xor butterfly these replace goes through ds_bpermute: six dependent LDS round trips per sum,
// ~600 cycles -- most of the tail of the one-or-two-workgroup kernels that use them.)
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
#define MPNN_DPP_QUAD_XOR1 0xB1      // quad_perm [1,0,3,2]
#define MPNN_DPP_QUAD_XOR2 0x4E      // quad_perm [2,3,0,1]
#define MPNN_DPP_HALF_MIRROR 0x141   // lane i <-> 7-i inside each half row
#define MPNN_DPP_ROW_MIRROR 0x140    // lane i <-> 15-i inside each row

__device__ __forceinline__ float wave_sum_f(float v) {
    v += __int_as_float(dpp_i<MPNN_DPP_QUAD_XOR1>(__float_as_int(v)));
    v += __int_as_float(dpp_i<MPNN_DPP_QUAD_XOR2>(__float_as_int(v)));
    v += __int_as_float(dpp_i<MPNN_DPP_HALF_MIRROR>(__float_as_int(v)));
    v += __int_as_float(dpp_i<MPNN_DPP_ROW_MIRROR>(__float_as_int(v)));
    const int b = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i<CTRL>((int)b), hi = dpp_i<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ double lane_d(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_d<MPNN_DPP_QUAD_XOR1>(v);
    v += dpp_d<MPNN_DPP_QUAD_XOR2>(v);
    v += dpp_d<MPNN_DPP_HALF_MIRROR>(v);
    v += dpp_d<MPNN_DPP_ROW_MIRROR>(v);
    return (lane_d(v, 0) + lane_d(v, 16)) + (lane_d(v, 32) + lane_d(v, 48));
}
