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
#include "common.h"

#define RB 64     // threads (samples) per workgroup

__global__ __launch_bounds__(RB) void route_k(const mpnn_route_args a) {
    __shared__ float P[MPNN_MAX_NODES * RB];      // p_tr per node
    __shared__ float V[MPNN_MAX_NODES * RB];      // actor: V ; critic: c_ev
    __shared__ float O[MPNN_MAX_NODES * RB];      // critic: c_opt
    const int t = threadIdx.x;
    const int s = blockIdx.x * RB + t;
    const bool live = s < a.n;
    const int n = a.n, NN = a.n_nodes, MS = a.max_sinks;
    const float inv_n = 1.f / (float)a.n_total;
    const float tau = a.hyp[MPNN_HYP_TAU], eps = a.hyp[MPNN_HYP_EPS];
    const float k_cpt = a.k_cpt_vec ? (live ? a.k_cpt_vec[s] : 0.f) : a.hyp[MPNN_HYP_KCPT];
    const float k_dec = a.hyp[MPNN_HYP_KDEC], k_cre = a.hyp[MPNN_HYP_KCRE];
    const float root_leaves = (float)a.nodes[5];
    const bool dyn = a.net_type != MPNN_NET_SR;
    float l_err = 0.f, l_cpt = 0.f, l_aux = 0.f;

    // ---- top-down: p_tr, p_ev (DFS preorder: parents first) ----
    for (int j = 0; j < NN; ++j) {
        const int *nd = a.nodes + j * 8;
        const int par = nd[0], si = nd[1];
        float ptr = 1.f, pev = 1.f;
        if (par >= 0 && live) {
            const int *pn = a.nodes + par * 8;
            const int psw = pn[3], pns = pn[2];
            const float pp = P[par * RB + t], ppe = a.p_ev[(size_t)par * n + s];
            if (dyn && psw >= 0 && pns >= 2) {
                const float *r = a.r + ((size_t)psw * n + s) * MS;
                float mx = r[0]; int arg = 0;
                for (int i = 1; i < pns; ++i) { if (r[i] > r[arg]) arg = i; mx = fmaxf(mx, r[i]); }
                float den = 0.f, num = 0.f;
                for (int i = 0; i < pns; ++i) { const float e = expf((r[i] - mx) / tau); den += e; if (i == si) num = e; }
                const float eps_p = eps * (float)pn[5] / root_leaves, eps_c = eps * (float)nd[5] / root_leaves;
                ptr = (pp - eps_p) * (num / den) + eps_c;
                pev = arg == si ? ppe : 0.f;
            } else { ptr = pp; pev = ppe; }
        }
        P[j * RB + t] = ptr;
        if (live) { a.p_tr[(size_t)j * n + s] = ptr; a.p_ev[(size_t)j * n + s] = pev; }
    }

    // ---- node statistics for TALR: sum p_tr, sum p_tr^2 ----
    if (a.node_stat) {
        for (int j = 0; j < NN; ++j) {
            const float p = live ? P[j * RB + t] : 0.f;
            const float s1 = wave_sum_f(p), s2 = wave_sum_f(p * p);
            if (t == 0) { atomicAdd(a.node_stat + j * 2, s1); atomicAdd(a.node_stat + j * 2 + 1, s2); }
        }
    }

    // ---- bottom-up (reverse preorder: children first) ----
    for (int j = NN - 1; j >= 0; --j) {
        const int *nd = a.nodes + j * 8;
        const int ns = nd[2], sw = nd[3], leaf = nd[4];
        const float p = P[j * RB + t];
        float cerr = 0.f, dcor = 1.f;
        if (leaf >= 0 && live) { cerr = a.c_err[(size_t)leaf * n + s]; dcor = a.d_cor[(size_t)leaf * n + s]; }
        const float ops = a.node_ops[j];
        if (leaf >= 0 && live && a.w_cerr)
            a.w_cerr[(size_t)leaf * n + s] = (a.net_type == MPNN_NET_SR ? 1.f : p) * inv_n;
        l_err += (a.net_type == MPNN_NET_SR ? 1.f : p) * cerr;
        if (a.net_type == MPNN_NET_ACTOR) l_cpt += p * k_cpt * ops;

        if (a.net_type == MPNN_NET_ACTOR) {
            float v = cerr + k_cpt * ops;
            if (sw >= 0 && ns >= 2) {
                const float *r = a.r + ((size_t)sw * n + s) * MS;
                const int *kids = a.sw_children + sw * MS;
                float sm[MPNN_MAX_SINKS], u[MPNN_MAX_SINKS], mx = live ? r[0] : 0.f, den = 0.f, r2 = 0.f;
                for (int i = 1; i < ns; ++i) mx = fmaxf(mx, live ? r[i] : 0.f);
                for (int i = 0; i < ns; ++i) { sm[i] = expf(((live ? r[i] : 0.f) - mx) / tau); den += sm[i]; }
                const float eps_l = eps * (float)nd[5] / root_leaves;
                float ubar = 0.f;
                for (int i = 0; i < ns; ++i) {
                    sm[i] /= den;
                    const float vc = V[kids[i] * RB + t];
                    v += sm[i] * vc;
                    u[i] = (p - eps_l) * vc;
                    ubar += sm[i] * u[i];
                    r2 += live ? r[i] * r[i] : 0.f;
                }
                l_aux += p * k_dec * r2;
                if (a.want_grad && live)
                    for (int i = 0; i < ns; ++i)
                        a.dr[((size_t)sw * n + s) * MS + i] =
                            (sm[i] * (u[i] - ubar) / tau + 2.f * k_dec * p * r[i]) * inv_n;
            } else {
                for (int k = j + 1; k < NN; ++k) if (a.nodes[k * 8] == j) v += V[k * RB + t];
            }
            V[j * RB + t] = v;
        } else if (a.net_type == MPNN_NET_CRITIC) {
            const float ce = a.use_cls_err ? (leaf >= 0 ? 1.f - dcor : 0.f) : cerr;
            float cev = ce + k_cpt * ops, copt = cev;
            if (sw >= 0 && ns >= 2) {
                const float *r = a.r + ((size_t)sw * n + s) * MS;
                const int *kids = a.sw_children + sw * MS;
                int arg = 0;
                for (int i = 1; i < ns; ++i) if (live && r[i] > r[arg]) arg = i;
                float mn = O[kids[0] * RB + t], cre = 0.f;
                for (int i = 0; i < ns; ++i) {
                    const float kev = V[kids[i] * RB + t], kopt = O[kids[i] * RB + t];
                    mn = fminf(mn, kopt);
                    if (i == arg) cev += kev;
                    const float tgt = a.optimistic ? kopt : kev;
                    const float d = (live ? r[i] : 0.f) + tgt;
                    cre += d * d;
                    if (a.want_grad && live)
                        a.dr[((size_t)sw * n + s) * MS + i] = p * 2.f * k_cre * d * inv_n;
                }
                copt += mn;
                l_aux += p * k_cre * cre;
            } else {
                for (int k = j + 1; k < NN; ++k)
                    if (a.nodes[k * 8] == j) { cev += V[k * RB + t]; copt += O[k * RB + t]; }
            }
            V[j * RB + t] = cev; O[j * RB + t] = copt;
        }
    }

    if (a.loss) {
        const double e = wave_sum_d(live ? (double)l_err : 0.0), c = wave_sum_d(live ? (double)l_cpt : 0.0);
        const double x = wave_sum_d(live ? (double)l_aux : 0.0), cnt = wave_sum_d(live ? 1.0 : 0.0);
        if (t == 0) { atomicAdd(a.loss, e); atomicAdd(a.loss + 1, c); atomicAdd(a.loss + 2, x); atomicAdd(a.loss + 3, cnt); }
    }
}

extern "C" int mpnn_route(const mpnn_route_args *args, void *stream) {
    if (!args || !args->nodes || !args->p_tr || !args->p_ev) return MPNN_E_ARG;
    if (args->n_nodes > MPNN_MAX_NODES || args->max_sinks > MPNN_MAX_SINKS) return MPNN_E_SHAPE;
    if (args->n <= 0) return 0;
    hipLaunchKernelGGL(route_k, dim3((args->n + RB - 1) / RB), dim3(RB), 0, (hipStream_t)stream, *args);
    MPNN_LAUNCH_CHECK();
    return 0;
}
