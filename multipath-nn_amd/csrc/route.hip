// mpnn_route: the data-dependent router -- routing probabilities, expected
// costs and their gradients for SR / actor / critic nets, on device, one
// thread per sample, no host sync.
//
// Restates (reference: scripts/lib/net_types.py)
//   _route / _route_sinks_stat / _route_sinks_dyn         :108-131, :193-243
//   ActorNet cost assembly                                :165-177
//   CriticNet c_ev / c_opt / c_cre and cost assembly      :204-243, :273-280
//   SRNet loss                                            :93-95
// with the algebraic identity  p_tr(child_i) = (p_tr - eps_l) * softmax(r/tau)_i
// + eps_child_i  (eps_x = eps * n_leaves(x) / n_leaves(root)), which is
// p_tr * pi_tr[:, i] of :123-126 multiplied out.
//
// Actor gradient (hand-derived; checked against autograd of the restated
// graph in tests): with V(l) = dL_n/dp_tr(l) = c_err(l) + k_cpt*ops(l)
// + sum_i softmax_i * V(child_i), u_i = (p_tr - eps_l) * V(child_i):
//   dL/dr_j = [ softmax_j * (u_j - sum_i softmax_i u_i) / tau
//               + 2 * k_dec * p_tr * r_j ] / n
//   dL/dc_err(l) = p_tr(l) / n
// Critic: every cost is weighted by stop_gradient(p_tr), so only c_cre
// reaches the router: dL/dr_i = p_tr * 2 * k_cre * (r_i + target_i) / n.
//
// Organisation (this launch is on the step's critical path): all 256 threads
// preload the node table and every per-sample input into LDS in one burst;
// then one wave walks the tree top-down (p_tr, p_ev) and bottom-up, each node
// PUSHING its value into its parent's accumulator (no child searches); all
// per-thread arrays have compile-time bounds so nothing lives in scratch.
#include "common.h"

#define MSK MPNN_MAX_SINKS

// RB = samples per workgroup (the LDS rows): 64 for the chains; 32 / 16 for trees whose tables
// (47 blocks + 47 leaves, 39 switches: arch_and_hypers.py:99-127) would not fit 160 KB at 64.  The
// working wave always has 64 lanes: lanes >= RB repeat lane RB-1's sample (same inputs, same
// arithmetic, identical LDS writes) and keep out of every global write and sum.
template <int RB>
__global__ __launch_bounds__(256) void route_k(const mpnn_route_args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n = a.n, NN = a.n_nodes, MS = a.max_sinks;
    trace_stamp(0); trace_note(6, 14);
    float *P = lds;                               // p_tr per node            [NN][RB]
    float *V = P + NN * RB;                       // actor: V ; critic: c_ev  [NN][RB]
    float *O = V + NN * RB;                       // critic: c_opt            [NN][RB]
    float *RIN = O + NN * RB;                     // router outputs           [n_switches*MS][RB]
    float *CE = RIN + a.n_switches * MS * RB;     // leaf c_err               [n_leaves][RB]
    float *DC = CE + a.n_leaves * RB;             // leaf delta_cor           [n_leaves][RB]
    int *ND = (int *)(DC + a.n_leaves * RB);      // node table               [NN][8]
    float *OPS = (float *)(ND + NN * 8);          // node ops                 [NN]
    float *SM = OPS + ((NN + 3) & ~3);            // softmax(r / tau)         [n_switches*MS][RB]
    int *ARG = (int *)(SM + a.n_switches * MS * RB);   // arg-max sink       [n_switches][RB]
    int *SWN = ARG + a.n_switches * RB;           // sinks of each switch     [n_switches]
    int *LV = SWN + a.n_switches;                 // nodes in (depth, preorder) order   [NN]
    int *LS = LV + NN;                            // first LV slot of each depth level   [NN + 2]
    {
        const int s0 = blockIdx.x * RB;
        const int rows_r = a.n_switches * MS, rows = rows_r + 2 * a.n_leaves;
        // Eight loads in flight per thread, from clamped addresses with no branch around them (a rolled
        // loop with one conditional load per iteration is one dependent memory round trip per iteration:
        // 8 of them = most of this kernel's time before).
        const int total = rows * RB;
        for (int base = 0; base < total; base += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + threadIdx.x + u * 256;
                const int ic = i < total ? i : 0;
                const int row = ic / RB, t = ic - row * RB, s = s0 + t;
                const bool ok = i < total && s < n;
                const int sc = ok ? s : 0;
                const float *src = row < rows_r ? a.r + ((size_t)(row / MS) * n + sc) * MS + (row % MS)
                                 : row < rows_r + a.n_leaves ? a.c_err + (size_t)(row - rows_r) * n + sc
                                                             : a.d_cor + (size_t)(row - rows_r - a.n_leaves) * n + sc;
                v[u] = *src;
                v[u] = ok ? v[u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + threadIdx.x + u * 256;
                if (i < total) RIN[i] = v[u];      // RIN, CE, DC are contiguous
            }
        }
        for (int i = threadIdx.x; i < NN * 8; i += 256) ND[i] = a.nodes[i];
        for (int i = threadIdx.x; i < NN; i += 256) {
            const int sw = a.nodes[i * 8 + 3];
            if (sw >= 0) SWN[sw] = a.nodes[i * 8 + 2];
        }
        for (int i = threadIdx.x; i < NN; i += 256) OPS[i] = a.node_ops[i];
        // level lists from the table's depth (nd[6]) and rank in (depth, preorder) order (nd[7])
        for (int i = threadIdx.x; i < NN + 2; i += 256) LS[i] = NN;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NN; i += 256) {
        const int d = a.nodes[i * 8 + 6], rk = a.nodes[i * 8 + 7];
        LV[rk] = i;
        atomicMin(&LS[d], rk);
    }
    __syncthreads();
    trace_stamp(1);
    // softmax(r / tau) and arg-max (first index on ties) of EVERY (switch, sample), by all four waves:
    // the serial tree walks below only read them (they used to recompute them up to three times per
    // node inside the one working wave).
    {
        const float inv_tau0 = 1.f / a.hyp[MPNN_HYP_TAU];
        for (int i = threadIdx.x; i < a.n_switches * RB; i += 256) {
            const int sw = i / RB, tt = i - sw * RB;
            const int ns = SWN[sw];
            const float *r = RIN + (sw * MS) * RB + tt;
            float mx = r[0];
            int arg = 0;
#pragma unroll
            for (int k = 1; k < MSK; ++k)
                if (k < ns) { if (r[k * RB] > r[arg * RB]) arg = k; mx = fmaxf(mx, r[k * RB]); }
            float e[MSK], den = 0.f;
#pragma unroll
            for (int k = 0; k < MSK; ++k) { e[k] = k < ns ? expf((r[k * RB] - mx) * inv_tau0) : 0.f; den += e[k]; }
            const float inv = 1.f / den;
#pragma unroll
            for (int k = 0; k < MSK; ++k) if (k < MS) SM[(sw * MS + k) * RB + tt] = e[k] * inv;
            ARG[sw * RB + tt] = arg;
        }
    }
    __syncthreads();
    trace_stamp(2);
    const int wave = threadIdx.x >> 6;
    int D = 0;                                     // deepest level
    for (int d = 0; d < NN; ++d) if (LS[d] < NN) D = d;
    const int lane_t = threadIdx.x & 63;
    const int t = lane_t < RB ? lane_t : RB - 1;
    const int s = blockIdx.x * RB + t;
    const bool in_range = s < n, live = in_range && lane_t < RB;
    const float inv_n = 1.f / (float)a.n_total;
    const float tau = a.hyp[MPNN_HYP_TAU], eps = a.hyp[MPNN_HYP_EPS];
    const float k_cpt = a.k_cpt_vec ? (in_range ? a.k_cpt_vec[s] : 0.f) : a.hyp[MPNN_HYP_KCPT];
    const float k_dec = a.hyp[MPNN_HYP_KDEC], k_cre = a.hyp[MPNN_HYP_KCRE];
    const float inv_tau = 1.f / tau;
    const float eps_unit = eps / (float)ND[5];    // eps / n_leaves(root)
    const int type = a.net_type;
    const bool dyn = type != MPNN_NET_SR;
    float l_err = 0.f, l_cpt = 0.f, l_aux = 0.f;

    // softmax / arg-max of switch sw for this sample: precomputed above
    auto soft = [&](int sw, int ns, float *sm, int &arg) {
#pragma unroll
        for (int i = 0; i < MSK; ++i) sm[i] = (i < ns && i < MS) ? SM[(sw * MS + i) * RB + t] : 0.f;
        arg = ARG[sw * RB + t];
    };

    // ---- top-down, one tree LEVEL at a time: p_tr, p_ev; own cost terms ----
    // The nodes of a level only depend on the level above, so the four waves take them in parallel (the
    // chains have two nodes per level: nine steps instead of a 17-node serial walk in one wave).
    for (int d = 0; d <= D; ++d) {
        for (int kq = LS[d] + wave; kq < LS[d + 1]; kq += 4) {
            const int j = LV[kq];
            const int *nd = ND + j * 8;
            const int par = nd[0], si = nd[1], leaf = nd[4];
            float ptr = 1.f, pev = 1.f;
            if (par >= 0) {
                const int *pn = ND + par * 8;
                const int psw = pn[3], pns = pn[2];
                const float pp = P[par * RB + t], ppe = O[par * RB + t];   // O doubles as p_ev storage top-down
                if (dyn && psw >= 0 && pns >= 2) {
                    float sm[MSK]; int arg;
                    soft(psw, pns, sm, arg);
                    float mine = 0.f;
#pragma unroll
                    for (int i = 0; i < MSK; ++i) if (i == si) mine = sm[i];
                    ptr = (pp - eps_unit * (float)pn[5]) * mine + eps_unit * (float)nd[5];
                    pev = arg == si ? ppe : 0.f;
                } else { ptr = pp; pev = ppe; }
            }
            P[j * RB + t] = ptr;
            O[j * RB + t] = pev;
            if (live) { a.p_tr[(size_t)j * n + s] = ptr; a.p_ev[(size_t)j * n + s] = pev; }
            // own terms of the bottom-up recursions
            const float cerr = leaf >= 0 ? CE[leaf * RB + t] : 0.f;
            const float dcor = leaf >= 0 ? DC[leaf * RB + t] : 1.f;
            const float ops = OPS[j];
            const float w = type == MPNN_NET_SR ? 1.f : ptr;
            if (leaf >= 0 && live && a.w_cerr) a.w_cerr[(size_t)leaf * n + s] = w * inv_n;
            l_err += w * cerr;
            if (type == MPNN_NET_ACTOR) { l_cpt += ptr * k_cpt * ops; V[j * RB + t] = cerr + k_cpt * ops; }
            else if (type == MPNN_NET_CRITIC) {
                const float ce = a.use_cls_err ? (leaf >= 0 ? 1.f - dcor : 0.f) : cerr;
                V[j * RB + t] = ce + k_cpt * ops;      // c_ev accumulator (children pushed below)
            }
        }
        __syncthreads();
    }
    trace_stamp(3);
    // ---- node statistics for TALR: sum p_tr, sum p_tr^2 (every wave its share of the nodes) ----
    if (a.node_stat) {
        for (int j = wave; j < NN; j += 4) {
            const float p = live ? P[j * RB + t] : 0.f;
            const float s1 = wave_sum_f(p), s2 = wave_sum_f(p * p);
            if (lane_t == 0) { atomicAdd(a.node_stat + j * 2, s1); atomicAdd(a.node_stat + j * 2 + 1, s2); }
        }
    }
    if (type == MPNN_NET_CRITIC) {
        for (int j = wave; j < NN; j += 4) O[j * RB + t] = V[j * RB + t];     // c_opt starts from the same own term
        __syncthreads();
    }

    // ---- bottom-up, deepest level first; each node pushes into its (static) parent ----
    if (type != MPNN_NET_SR) {
        for (int d = D; d >= 0; --d) {
            for (int kq = LS[d] + wave; kq < LS[d + 1]; kq += 4) {
                const int j = LV[kq];
                const int *nd = ND + j * 8;
                const int par = nd[0], ns = nd[2], sw = nd[3];
                const float p = P[j * RB + t];
                if (sw >= 0 && ns >= 2) {              // dynamic switch: children are final
                    const float *r = RIN + (sw * MS) * RB + t;
                    const int *kids = a.sw_children + sw * MS;
                    float sm[MSK]; int arg;
                    soft(sw, ns, sm, arg);
                    if (type == MPNN_NET_ACTOR) {
                        const float eps_l = eps_unit * (float)nd[5];
                        float u[MSK], ubar = 0.f, r2 = 0.f, v = V[j * RB + t];
#pragma unroll
                        for (int i = 0; i < MSK; ++i) {
                            u[i] = 0.f;
                            if (i < ns) {
                                const float vc = V[kids[i] * RB + t];
                                v += sm[i] * vc;
                                u[i] = (p - eps_l) * vc;
                                ubar += sm[i] * u[i];
                                r2 += r[i * RB] * r[i * RB];
                            }
                        }
                        V[j * RB + t] = v;
                        l_aux += p * k_dec * r2;
                        if (a.want_grad && live) {
#pragma unroll
                            for (int i = 0; i < MSK; ++i)
                                if (i < ns)
                                    a.dr[((size_t)sw * n + s) * MS + i] =
                                        (sm[i] * (u[i] - ubar) * inv_tau + 2.f * k_dec * p * r[i * RB]) * inv_n;
                        }
                    } else {
                        float cev = V[j * RB + t], mn = 0.f, cre = 0.f;
#pragma unroll
                        for (int i = 0; i < MSK; ++i) {
                            if (i < ns) {
                                const float kev = V[kids[i] * RB + t], kopt = O[kids[i] * RB + t];
                                mn = i == 0 ? kopt : fminf(mn, kopt);
                                if (i == arg) cev += kev;
                                const float dd = r[i * RB] + (a.optimistic ? kopt : kev);
                                cre += dd * dd;
                                if (a.want_grad && live) a.dr[((size_t)sw * n + s) * MS + i] = p * 2.f * k_cre * dd * inv_n;
                            }
                        }
                        V[j * RB + t] = cev;
                        O[j * RB + t] += mn;
                        l_aux += p * k_cre * cre;
                    }
                }
                // push into a STATIC parent (a dynamic parent reads its children itself; a static node has
                // at most one sink, so nobody else writes the parent's slot in this level)
                if (par >= 0) {
                    const int *pn = ND + par * 8;
                    if (!(pn[3] >= 0 && pn[2] >= 2)) {
                        V[par * RB + t] += V[j * RB + t];
                        if (type == MPNN_NET_CRITIC) O[par * RB + t] += O[j * RB + t];
                    }
                }
            }
            __syncthreads();
        }
    }

    trace_stamp(4);
    if (a.loss) {
        const double e = wave_sum_d(live ? (double)l_err : 0.0), c = wave_sum_d(live ? (double)l_cpt : 0.0);
        const double x = wave_sum_d(live ? (double)l_aux : 0.0), cnt = wave_sum_d((live && wave == 0) ? 1.0 : 0.0);
        if (lane_t == 0) {
            atomicAdd(a.loss, e); atomicAdd(a.loss + 1, c); atomicAdd(a.loss + 2, x);
            if (wave == 0) atomicAdd(a.loss + 3, cnt);
        }
    }
    trace_stamp(5);
}

int mpnn_trace_install_route(void *buf) { return mpnn_trace_install(buf); }

extern "C" int mpnn_route(const mpnn_route_args *args, void *stream) {
    if (!args || !args->nodes || !args->p_tr || !args->p_ev) return MPNN_E_ARG;
    if (args->n_nodes > MPNN_MAX_NODES || args->max_sinks > MPNN_MAX_SINKS) return MPNN_E_SHAPE;
    if (args->n <= 0) return 0;
    const size_t per = (size_t)(3 * args->n_nodes + 2 * args->n_switches * args->max_sinks + args->n_switches + 2 * args->n_leaves) * 4;
    const size_t fix = (size_t)(args->n_nodes * 11 + 8 + args->n_switches) * 4;
    const size_t cap = 160 * 1024;
    const hipStream_t st = (hipStream_t)stream;
    const int n = args->n;
    static bool raised = false;
    if (!raised) {                                  // (more than the default 64 KB of dynamic LDS)
        hipFuncSetAttribute((const void *)route_k<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);
        hipFuncSetAttribute((const void *)route_k<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);
        hipFuncSetAttribute((const void *)route_k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);
        raised = true;
    }
    if (per * 64 + fix <= cap)      hipLaunchKernelGGL(route_k<64>, dim3((n + 63) / 64), dim3(256), per * 64 + fix, st, *args);
    else if (per * 32 + fix <= cap) hipLaunchKernelGGL(route_k<32>, dim3((n + 31) / 32), dim3(256), per * 32 + fix, st, *args);
    else if (per * 16 + fix <= cap) hipLaunchKernelGGL(route_k<16>, dim3((n + 15) / 16), dim3(256), per * 16 + fix, st, *args);
    else return MPNN_E_SHAPE;
    MPNN_LAUNCH_CHECK();
    return 0;
}
