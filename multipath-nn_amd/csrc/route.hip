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
// Organisation (this launch is on the step's critical path; two workgroups at batch 128, so what
// counts is the length of the dependent chain, not throughput):
//   1. all threads preload every per-sample input into LDS in one burst, pack one 16-byte record
//      per node in topological (depth, preorder) order, and compute softmax / arg-max of every
//      (switch, sample) and the nodes' own cost terms;
//   2. the two tree RECURRENCES run concurrently, each in one wave with no barriers: wave 0 top-down
//      (p_tr, p_ev), wave 1 bottom-up (actor V / critic c_ev, c_opt) -- they do not depend on each
//      other -- each node = one LDS round trip (operands from clamped addresses, no control flow,
//      the next record fetched ahead);
//   3. one barrier, then everything that is merely a FUNCTION of those values (global stores, dL/dr,
//      loss terms, TALR statistics) by all four waves, nodes dealt round-robin.
// The first version did all of a node's work inside the level-by-level walk: ~1 us per tree level,
// 17 of the launch's 22 us.  All per-thread arrays have compile-time bounds: nothing in scratch.
#include "common.h"

#define MSK MPNN_MAX_SINKS

// RB = samples per workgroup (the LDS rows): 64 for the chains; 32 / 16 for trees whose tables
// (47 blocks + 47 leaves, 39 switches: arch_and_hypers.py:99-127) would not fit 160 KB at 64.  The
// working wave always has 64 lanes: lanes >= RB repeat lane RB-1's sample (same inputs, same
// arithmetic, identical LDS writes) and keep out of every global write and sum.
#define RT_THREADS 512       // eight waves: the preload and the read-only phase scale with them
#define RT_WAVES (RT_THREADS / 64)

__device__ __forceinline__ int4 uniform4(int4 v) {   // a record is the same in every lane: keep it in SGPRs
    return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y),
                     __builtin_amdgcn_readfirstlane(v.z), __builtin_amdgcn_readfirstlane(v.w));
}

// blk = workgroup index within the net's launch (RB samples each)
// The deterministic statistics exchange: partials leave with write-through (system-scope) stores and are read back
// uncached, so that what the last-arriving workgroup adds never depends on another XCD's L2.
template <class T> __device__ __forceinline__ void st_part(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
template <class T> __device__ __forceinline__ T ld_part(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

template <int RB>
__device__ __forceinline__ void route_body(const mpnn_route_args &a, const int blk, const int nblk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n = a.n, NN = a.n_nodes, MS = a.max_sinks;
    const bool det_stat = a.node_stat && a.stat_part && nblk > 2;     // (uniform; <= 2 workgroups: two atomics onto a cleared sum commute)
    trace_stamp(0); trace_note(6, 14);
    float *P = lds;                               // p_tr per node            [NN][RB]
    float *PE = P + NN * RB;                      // p_ev per node            [NN][RB]
    float *V = PE + NN * RB;                      // actor: V ; critic: c_ev  [NN][RB]
    float *O = V + NN * RB;                       // critic: c_opt            [NN][RB]
    float *RIN = O + NN * RB;                     // router outputs           [n_switches*MS][RB]
    float *CE = RIN + a.n_switches * MS * RB;     // leaf c_err               [n_leaves][RB]
    float *DC = CE + a.n_leaves * RB;             // leaf delta_cor           [n_leaves][RB]
    float *SM = DC + a.n_leaves * RB;             // softmax(r / tau)         [n_switches*MS][RB]
    int *ARG = (int *)(SM + a.n_switches * MS * RB);   // arg-max sink       [n_switches][RB]
    int4 *REC = (int4 *)(ARG + a.n_switches * RB);     // node records by rank   [NN + 1]  (16-byte aligned: RB % 4 == 0)
    float *OPS = (float *)(REC + NN + 1);         // node ops                 [NN]
    int *SWN = (int *)(OPS + NN);                 // sinks of each switch     [n_switches]
    const int s0 = blk * RB;
    const int wave = threadIdx.x >> 6;
    constexpr int BT = RT_THREADS - 64;            // threads of the input burst; the last wave packs the records meanwhile
    if (wave < RT_WAVES - 1) {
        const int rows_r = a.n_switches * MS, rows = rows_r + 2 * a.n_leaves;
        // Eight loads in flight per thread, from clamped addresses with no branch around them (a rolled
        // loop with one conditional load per iteration is one dependent memory round trip per iteration:
        // 8 of them = most of this kernel's time before).
        const int total = rows * RB;
        for (int base = 0; base < total; base += BT * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + threadIdx.x + u * BT;
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
                const int i = base + threadIdx.x + u * BT;
                if (i < total) RIN[i] = v[u];      // RIN, CE, DC are contiguous
            }
        }
    } else {
        // One record per node, stored at its rank nd[7] in (depth, preorder) order -- parents before
        // children -- with everything the walks need about the node AND its parent:
        //   x = par+1 | si<<8 | ns<<12 | (sw+1)<<16 | (leaf+1)<<24
        //   y = n_leaves | n_leaves(par)<<8 | (sw(par)+1)<<16 | ns(par)<<24
        //   z = the children of the node's switch, 8 bits each        w = node id
        for (int i = threadIdx.x - BT; i < NN; i += 64) {
            const int *nd = a.nodes + i * 8;
            const int par = nd[0], ns = nd[2], sw = nd[3];
            const int *pn = a.nodes + (par >= 0 ? par : 0) * 8;
            int kids = 0;
#pragma unroll
            for (int k = 0; k < MSK; ++k)
                if (sw >= 0 && k < ns && k < MS) kids |= (a.sw_children[sw * MS + k] & 255) << (8 * k);
            REC[nd[7]] = make_int4((par + 1) | (nd[1] << 8) | (ns << 12) | ((sw + 1) << 16) | ((nd[4] + 1) << 24),
                                   nd[5] | (pn[5] << 8) | ((par >= 0 ? pn[3] + 1 : 0) << 16) | ((par >= 0 ? pn[2] : 0) << 24),
                                   kids, i);
            if (sw >= 0) SWN[sw] = ns;
            OPS[i] = a.node_ops[i];
        }
        if (threadIdx.x == BT) REC[NN] = make_int4(0, 0, 0, 0);   // (the walks fetch one record ahead)
    }
    __syncthreads();
    trace_stamp(1);
    const int type = a.net_type;
    const bool dyn = type != MPNN_NET_SR;
    // softmax(r / tau) and arg-max (first index on ties) of EVERY (switch, sample), and every node's own
    // term of the bottom-up recursions, by all four waves: the walks below only read them.
    {
        const float inv_tau0 = 1.f / a.hyp[MPNN_HYP_TAU];
        for (int i = threadIdx.x; i < a.n_switches * RB; i += RT_THREADS) {
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
        if (dyn) {
            const int tt = threadIdx.x % RB;           // (RT_THREADS % RB == 0: the same sample every iteration)
            const float kc = a.k_cpt_vec ? (s0 + tt < n ? a.k_cpt_vec[s0 + tt] : 0.f) : a.hyp[MPNN_HYP_KCPT];
            const bool cls = type == MPNN_NET_CRITIC && a.use_cls_err;
            for (int i = threadIdx.x; i < NN * RB; i += RT_THREADS) {
                const int j = i / RB, leaf = a.nodes[j * 8 + 4];
                const int lc = leaf >= 0 ? leaf : 0;
                const float ce = cls ? 1.f - DC[lc * RB + tt] : CE[lc * RB + tt];
                const float own = (leaf >= 0 ? ce : 0.f) + kc * OPS[j];
                V[i] = own;                            // actor: V; critic: c_ev accumulator
                O[i] = own;                            // critic: c_opt starts from the same own term
            }
        }
    }
    __syncthreads();
    trace_stamp(2);
    const int lane_t = threadIdx.x & 63;
    const int t = lane_t < RB ? lane_t : RB - 1;
    const int s = s0 + t;
    const bool in_range = s < n, live = in_range && lane_t < RB;
    const float inv_n = 1.f / (float)a.n_total;
    const float tau = a.hyp[MPNN_HYP_TAU], eps = a.hyp[MPNN_HYP_EPS];
    const float k_cpt = a.k_cpt_vec ? (in_range ? a.k_cpt_vec[s] : 0.f) : a.hyp[MPNN_HYP_KCPT];
    const float k_dec = a.hyp[MPNN_HYP_KDEC], k_cre = a.hyp[MPNN_HYP_KCRE];
    const float inv_tau = 1.f / tau;
    const float eps_unit = eps / (float)a.nodes[5];    // eps / n_leaves(root)
    float l_err = 0.f, l_cpt = 0.f, l_aux = 0.f;

    // p_tr, p_ev of a node from its parent's (operands from clamped addresses, no control flow)
    auto from_parent = [&](const int4 rc, float &ptr, float &pev) {
        const int par = (rc.x & 255) - 1, si = (rc.x >> 8) & 15;
        const int nl = rc.y & 255, pnl = (rc.y >> 8) & 255, psw = ((rc.y >> 16) & 255) - 1, pns = (rc.y >> 24) & 15;
        const int parc = par >= 0 ? par : 0, pswc = psw >= 0 ? psw : 0;
        const float pp = P[parc * RB + t], ppe = PE[parc * RB + t];
        const float mine = SM[(pswc * MS + (si < MS ? si : 0)) * RB + t];
        const int arg = ARG[pswc * RB + t];
        const bool dynp = dyn && psw >= 0 && pns >= 2;
        ptr = dynp ? (pp - eps_unit * (float)pnl) * mine + eps_unit * (float)nl : pp;
        pev = dynp ? (arg == si ? ppe : 0.f) : ppe;
        ptr = par >= 0 ? ptr : 1.f;
        pev = par >= 0 ? pev : 1.f;
    };
    if (wave == 0) {
        // ---- top-down recurrence: p_tr, p_ev of every node THAT HAS CHILDREN, parents first (the
        // childless ones -- half of a chain -- are nobody's operand: phase 3 evaluates them) ----
        int4 rc = uniform4(REC[0]);
        for (int kq = 0; kq < NN; ++kq) {
            const int4 nxv = REC[kq + 1];
            if (((rc.x >> 12) & 15) > 0) {
                float ptr, pev;
                from_parent(rc, ptr, pev);
                P[rc.w * RB + t] = ptr;
                PE[rc.w * RB + t] = pev;
            }
            rc = uniform4(nxv);
        }
    } else if (wave == 1 && dyn) {
        // ---- bottom-up recurrence, children first; a node with a static parent pushes into it (a
        // dynamic parent reads its children itself; a static node has at most one sink).  One loop per
        // net type and only the net's own sink count: this wave's instruction count IS the phase. ----
        const bool actor = type == MPNN_NET_ACTOR;
        int4 rc = uniform4(REC[NN - 1]);
        for (int kq = NN - 1; kq >= 0; --kq) {
            const int4 nxv = REC[kq > 0 ? kq - 1 : NN];       // (made uniform at the bottom: off the critical path)
            const int par = (rc.x & 255) - 1, ns = (rc.x >> 12) & 15, sw = ((rc.x >> 16) & 255) - 1, j = rc.w;
            const int psw = ((rc.y >> 16) & 255) - 1, pns = (rc.y >> 24) & 15;
            const bool is_sw = sw >= 0 && ns >= 2, push = par >= 0 && !(psw >= 0 && pns >= 2);
            if (is_sw || push) {                       // (else: a leaf under a dynamic switch, its own term is final)
                const int parc = par >= 0 ? par : 0, swc = sw >= 0 ? sw : 0;
                const float *smp = SM + swc * MS * RB + t;
                const float vj = V[j * RB + t], vp = V[parc * RB + t];
                float kev[MSK], sm[MSK];
#pragma unroll
                for (int i = 0; i < MSK; ++i)
                    if (i < MS) { kev[i] = V[((rc.z >> (8 * i)) & 255) * RB + t]; sm[i] = smp[i * RB]; }
                float v_new = vj;
                if (actor) {
                    if (is_sw) {
#pragma unroll
                        for (int i = 0; i < MSK; ++i)
                            if (i < MS) v_new += i < ns ? sm[i] * kev[i] : 0.f;
                        V[j * RB + t] = v_new;
                    }
                    if (push) V[parc * RB + t] = vp + v_new;
                } else {
                    const float oj = O[j * RB + t], op = O[parc * RB + t];
                    const int arg = ARG[swc * RB + t];
                    float kopt[MSK];
#pragma unroll
                    for (int i = 0; i < MSK; ++i)
                        if (i < MS) kopt[i] = O[((rc.z >> (8 * i)) & 255) * RB + t];
                    float o_new = oj;
                    if (is_sw) {
                        float mn = kopt[0];
#pragma unroll
                        for (int i = 0; i < MSK; ++i)
                            if (i < MS) {
                                v_new += (i < ns && i == arg) ? kev[i] : 0.f;
                                mn = i < ns ? fminf(mn, kopt[i]) : mn;
                            }
                        o_new += mn;
                        V[j * RB + t] = v_new;
                        O[j * RB + t] = o_new;
                    }
                    if (push) { V[parc * RB + t] = vp + v_new; O[parc * RB + t] = op + o_new; }
                }
            }
            rc = uniform4(nxv);
        }
    }
    __syncthreads();
    trace_stamp(3);

    // ---- everything that only READS the recurrences: nodes dealt to the four waves ----
    for (int kq = wave; kq < NN; kq += RT_WAVES) {
        const int4 rc = uniform4(REC[kq]);
        const int ns = (rc.x >> 12) & 15, sw = ((rc.x >> 16) & 255) - 1, leaf = ((rc.x >> 24) & 255) - 1, j = rc.w;
        const int nl = rc.y & 255;
        const int swc = sw >= 0 ? sw : 0, leafc = leaf >= 0 ? leaf : 0;
        float ptr = P[j * RB + t], pev = PE[j * RB + t];
        const float cerr_l = CE[leafc * RB + t], ops = OPS[j];
        float ptr_c, pev_c;
        from_parent(rc, ptr_c, pev_c);
        ptr = ns > 0 ? ptr : ptr_c;                // (childless: not stored by the recurrence)
        pev = ns > 0 ? pev : pev_c;
        float sm[MSK], rr[MSK], kev[MSK], kopt[MSK];
#pragma unroll
        for (int i = 0; i < MSK; ++i) {
            sm[i] = rr[i] = kev[i] = kopt[i] = 0.f;
            if (i < MS && dyn) {
                const int kid = (rc.z >> (8 * i)) & 255;
                sm[i] = SM[(swc * MS + i) * RB + t];
                rr[i] = RIN[(swc * MS + i) * RB + t];
                kev[i] = V[kid * RB + t];
                kopt[i] = O[kid * RB + t];
            }
        }
        const float cerr = leaf >= 0 ? cerr_l : 0.f;
        const float w = type == MPNN_NET_SR ? 1.f : ptr;
        if (live) {
            a.p_tr[(size_t)j * n + s] = ptr; a.p_ev[(size_t)j * n + s] = pev;
            if (leaf >= 0 && a.w_cerr) a.w_cerr[(size_t)leaf * n + s] = w * inv_n;
        }
        l_err += w * cerr;
        if (type == MPNN_NET_ACTOR) l_cpt += ptr * k_cpt * ops;
        if (a.node_stat) {                         // TALR: sum p_tr, sum p_tr^2
            const float pl = live ? ptr : 0.f;
            const float s1 = wave_sum_f(pl), s2 = wave_sum_f(pl * pl);
            if (lane_t == 0) {
                if (det_stat) { st_part(a.stat_part + ((size_t)blk * NN + j) * 2, s1); st_part(a.stat_part + ((size_t)blk * NN + j) * 2 + 1, s2); }
                else { atomicAdd(a.node_stat + j * 2, s1); atomicAdd(a.node_stat + j * 2 + 1, s2); }
            }
        }
        if (dyn && sw >= 0 && ns >= 2) {
            const float p = ptr;
            if (type == MPNN_NET_ACTOR) {
                const float eps_l = eps_unit * (float)nl;
                float u[MSK], ubar = 0.f, r2 = 0.f;
#pragma unroll
                for (int i = 0; i < MSK; ++i) {
                    const bool on = i < ns;
                    u[i] = on ? (p - eps_l) * kev[i] : 0.f;
                    ubar += on ? sm[i] * u[i] : 0.f;
                    r2 += on ? rr[i] * rr[i] : 0.f;
                }
                l_aux += p * k_dec * r2;
                if (a.want_grad && live) {
#pragma unroll
                    for (int i = 0; i < MSK; ++i)
                        if (i < ns)
                            a.dr[((size_t)sw * n + s) * MS + i] = (sm[i] * (u[i] - ubar) * inv_tau + 2.f * k_dec * p * rr[i]) * inv_n;
                }
            } else {
                float cre = 0.f;
#pragma unroll
                for (int i = 0; i < MSK; ++i) {
                    if (i < ns) {
                        const float dd = rr[i] + (a.optimistic ? kopt[i] : kev[i]);
                        cre += dd * dd;
                        if (a.want_grad && live) a.dr[((size_t)sw * n + s) * MS + i] = p * 2.f * k_cre * dd * inv_n;
                    }
                }
                l_aux += p * k_cre * cre;
            }
        }
    }

    trace_stamp(4);
    if (a.loss) {
        const double e = wave_sum_d(live ? (double)l_err : 0.0), c = wave_sum_d(live ? (double)l_cpt : 0.0);
        const double x = wave_sum_d(live ? (double)l_aux : 0.0), cnt = wave_sum_d((live && wave == 0) ? 1.0 : 0.0);
        // the eight waves' sums meet in LDS (fixed order) and leave as ONE atomic per sum and workgroup: same-address
        // atomics serialise across the chip -- with one per wave the launch took 32 us at 4 096 samples (64 workgroups)
        // (in the head of the dynamic LDS area, once every wave is done with the tables: the launch may use all 160 KB)
        double *lsum = (double *)lds;                     // [RT_WAVES][4]
        __syncthreads();
        if (lane_t == 0) { lsum[wave * 4] = e; lsum[wave * 4 + 1] = c; lsum[wave * 4 + 2] = x; lsum[wave * 4 + 3] = wave == 0 ? cnt : 0.0; }
        __syncthreads();
        if (threadIdx.x < 4) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < RT_WAVES; ++w) t += lsum[w * 4 + threadIdx.x];
            if (det_stat) st_part((double *)(a.stat_part + (size_t)nblk * NN * 2) + blk * 4 + threadIdx.x, t);     // (summed below, in order)
            else atomicAdd(a.loss + threadIdx.x, t);
        }
    }
    if (det_stat) {
        // the partial sums of all workgroups, added in workgroup order by whichever finishes last
        int *last_s = (int *)lds + 128;                   // (dynamic LDS, behind the loss sums: the launch may use all 160 KB)
        // RELEASE by every writer: the partials were written by lane 0 of SEVERAL waves with write-through stores
        // (st_part); each wave waits for the acknowledgement of its own before the barrier in front of the ticket.  The
        // barrier alone is a workgroup-scope release (lgkmcnt only) and thread 0's fence below only covers wave 0's
        // stores -- another wave's partial could reach memory after the ticket (csrc/lin.hip uses the same exchange).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            *last_s = atomicAdd(a.stat_ticket, 1) == nblk - 1;
        }
        __syncthreads();
        if (*last_s) {
            __threadfence();
            for (int i = threadIdx.x; i < 2 * NN; i += RT_THREADS) {
                float t = 0.f;
                for (int b = 0; b < nblk; ++b) t += ld_part(a.stat_part + (size_t)b * NN * 2 + i);
                a.node_stat[i] += t;
            }
            if (a.loss && threadIdx.x < 4) {
                const double *lp = (const double *)(a.stat_part + (size_t)nblk * NN * 2);
                double t = 0.0;
                for (int b = 0; b < nblk; ++b) t += ld_part(lp + b * 4 + threadIdx.x);
                a.loss[threadIdx.x] += t;
            }
            if (threadIdx.x == 0) *a.stat_ticket = 0;
        }
    }
    trace_stamp(5);
}

template <int RB>
__global__ __launch_bounds__(RT_THREADS) void route_k(const mpnn_route_args a) { route_body<RB>(a, blockIdx.x, gridDim.x); }

// Several nets of one tree shape in one launch (co-training, lib/_co.py): wpn workgroups per net, net r's record tab[r].
template <int RB>
__global__ __launch_bounds__(RT_THREADS) void route_multi_k(const mpnn_route_args *__restrict__ tab, const int wpn) {
    const int net = blockIdx.x / wpn;
    const mpnn_route_args a = tab[net];            // (by value: every field's scalar load in the entry block)
    route_body<RB>(a, blockIdx.x - net * wpn, wpn);
}

int mpnn_trace_install_route(void *buf) { return mpnn_trace_install(buf); }

extern "C" int mpnn_route(const mpnn_route_args *args, void *stream) {
    if (!args || !args->nodes || !args->p_tr || !args->p_ev || (args->stat_part && !args->stat_ticket)) return MPNN_E_ARG;
    if (args->n_nodes > MPNN_MAX_NODES || args->max_sinks > MPNN_MAX_SINKS) return MPNN_E_SHAPE;
    if (args->n <= 0) return 0;
    const size_t per = (size_t)(4 * args->n_nodes + 2 * args->n_switches * args->max_sinks + args->n_switches + 2 * args->n_leaves) * 4;
    const size_t fix = (size_t)(args->n_nodes * 5 + 8 + args->n_switches) * 4;
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
    if (per * 64 + fix <= cap)      hipLaunchKernelGGL(route_k<64>, dim3((n + 63) / 64), dim3(RT_THREADS), per * 64 + fix, st, *args);
    else if (per * 32 + fix <= cap) hipLaunchKernelGGL(route_k<32>, dim3((n + 31) / 32), dim3(RT_THREADS), per * 32 + fix, st, *args);
    else if (per * 16 + fix <= cap) hipLaunchKernelGGL(route_k<16>, dim3((n + 15) / 16), dim3(RT_THREADS), per * 16 + fix, st, *args);
    else return MPNN_E_SHAPE;
    MPNN_LAUNCH_CHECK();
    return 0;
}

// mpnn_route for `count` nets with the same tree shape and batch size (host_table: the records, to size the launch;
// dev_table: the same records in device memory).
extern "C" int mpnn_route_multi(const mpnn_route_args *host_table, const mpnn_route_args *dev_table, int count, void *stream) {
    if (count <= 0) return 0;
    if (!host_table || !dev_table) return MPNN_E_ARG;
    const mpnn_route_args *args = host_table;
    for (int k = 0; k < count; ++k) {
        const mpnn_route_args &b = host_table[k];
        if (!b.nodes || !b.p_tr || !b.p_ev || (b.stat_part && !b.stat_ticket)) return MPNN_E_ARG;
        if (b.n != args->n || b.n_nodes != args->n_nodes || b.n_switches != args->n_switches || b.n_leaves != args->n_leaves ||
            b.max_sinks != args->max_sinks) return MPNN_E_ARG;
    }
    if (args->n_nodes > MPNN_MAX_NODES || args->max_sinks > MPNN_MAX_SINKS) return MPNN_E_SHAPE;
    if (args->n <= 0) return 0;
    const size_t per = (size_t)(4 * args->n_nodes + 2 * args->n_switches * args->max_sinks + args->n_switches + 2 * args->n_leaves) * 4;
    const size_t fix = (size_t)(args->n_nodes * 5 + 8 + args->n_switches) * 4;
    const size_t cap = 160 * 1024;
    const hipStream_t st = (hipStream_t)stream;
    const int n = args->n;
    static bool raised = false;
    if (!raised) {
        hipFuncSetAttribute((const void *)route_multi_k<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);
        hipFuncSetAttribute((const void *)route_multi_k<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);
        hipFuncSetAttribute((const void *)route_multi_k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);
        raised = true;
    }
    if (per * 64 + fix <= cap)      { const int w = (n + 63) / 64; hipLaunchKernelGGL(route_multi_k<64>, dim3(w * count), dim3(RT_THREADS), per * 64 + fix, st, dev_table, w); }
    else if (per * 32 + fix <= cap) { const int w = (n + 31) / 32; hipLaunchKernelGGL(route_multi_k<32>, dim3(w * count), dim3(RT_THREADS), per * 32 + fix, st, dev_table, w); }
    else if (per * 16 + fix <= cap) { const int w = (n + 15) / 16; hipLaunchKernelGGL(route_multi_k<16>, dim3(w * count), dim3(RT_THREADS), per * 16 + fix, st, dev_table, w); }
    else return MPNN_E_SHAPE;
    MPNN_LAUNCH_CHECK();
    return 0;
}
