// mpnn_msconv_bwd_level: one DEPENDENCY LEVEL of the backward pass as one launch.
//
// g(b, S) (block b, absolute scale S) is final once B(b+1, S) (dgrad-horz of the child block) and
// B(b, S+1) (dgrad-vert of the coarser scale) have run, so the launches B(b, S) = {dgrad-horz, dgrad-vert,
// weight gradients of g(b, S)} with equal (depth from the end) + (scales from the coarsest) do not depend
// on each other: the 20 backward launches of the 8-block chain are 11 levels.  A level's small members
// (4x4 / 8x8 maps: a few dozen workgroups that are nothing but their ~8 us prologue chain) run beside the
// large one instead of after it.  Autodiff of layer_types.py:181-185 as in wgrad.hip; the bodies are the
// same (conv_kernel.h, bwd_bodies.h).
//
// Members (<= MPNN_BWD_LEVEL_MAX) are records in DEVICE memory (a run-time index into a by-value argument
// array would put the whole block in scratch); a workgroup finds its member from the first-workgroup table,
// then its body as in bwd_scale_k.  The HOST decides how many workgroups every body gets (the caller's
// budget: lib/_plan.py shares the resident slots of the variant between the members); nothing here is
// sized implicitly.  One instantiation per set of map geometries present (GKMASK) and per set of
// weight-gradient tile widths (OTMASK: bit 0 = 16-channel groups, bit 1 = 64-channel groups), so that a
// level only pays the registers of the bodies it runs.
#include "bwd_level_k.h"

// ------------------------------- host ----------------------------------------
template <int OTMASK>
static LevelKern level_kernel_ot(int gkmask) {
    switch (gkmask) {
        case 1: return bwd_level_k<1, OTMASK, false>;
        case 2: return bwd_level_k<2, OTMASK, false>;
        case 3: return bwd_level_k<3, OTMASK, false>;
        case 4: return bwd_level_k<4, OTMASK, false>;
        case 5: return bwd_level_k<5, OTMASK, false>;
        case 6: return bwd_level_k<6, OTMASK, false>;
        case 7: return bwd_level_k<7, OTMASK, false>;
    }
    return nullptr;
}
LevelKern mpnn_level_kernel_smallc(int gkmask);          // bwd_level_small.hip
static LevelKern level_kernel(int gkmask, int otmask, bool smallc = false) {
    // (block 0's members only ever share a level with 16-channel members of block 1: no SMALLC variants of the levels with
    // 64-channel weight-gradient groups -- such a level would run the general body, which is correct for any C)
    if (smallc && MPNN_WG_SMALLC && !(otmask & 2)) return mpnn_level_kernel_smallc(gkmask);
    return (otmask & 2) ? level_kernel_ot<3>(gkmask) : level_kernel_ot<1>(gkmask);
}

static int geom_kind(int H, int W) {
    if (W >= 16 && (W % 16) == 0 && (H % 4) == 0) return 0;
    if (W == 8 && H == 8) return 1;
    if (W == 4 && H == 4) return 2;
    return -1;
}

int mpnn_trace_install_level(void *buf) { return mpnn_trace_install(buf); }

extern "C" int mpnn_msconv_bwd_level_record_size(void) { return (int)sizeof(BwdRec); }

extern "C" int mpnn_msconv_bwd_level_slots(const int *H, const int *W, const int *Cout, int count) {
    if (!H || !W || !Cout || count < 1 || count > MPNN_BWD_LEVEL_MAX) return MPNN_E_ARG;
    int gkmask = 0, otmask = 0;
    for (int k = 0; k < count; ++k) {
        const int gk = geom_kind(H[k], W[k]);
        if (gk < 0 || (Cout[k] % 16)) return MPNN_E_SHAPE;
        gkmask |= 1 << gk;
        otmask |= (Cout[k] % 64) == 0 ? 2 : 1;
    }
    // (levels with a 64-channel weight-gradient group: at most THREE workgroups per CU.  Rounds 2-5 capped them at two --
    // fewer, larger shares --; with three the co-trained joint step of 8 nets takes 1 983 instead of 2 002 us and the single
    // net's step is unchanged (485.1 / 485.5 us): profiles/r06_level_cap.txt.  MPNN_LEVEL_WIDE_CAP overrides, 0 = whatever fits)
    static const int wide_cap = [] { const char *e = getenv("MPNN_LEVEL_WIDE_CAP"); return e ? atoi(e) : 3; }();
    return resident_slots((const void *)level_kernel(gkmask, otmask), 0, 256, (otmask & 2) ? wide_cap : 0);
}

// records + first-workgroup table + kernel variant of a level; total = workgroups of the launch
// smallc (out): some member's operand A is a 1- / 3-channel image -- the kernel variant with the SMALLC weight-gradient body.
// (An out-parameter, not file-scope state: two engines building programs from different host threads would overwrite each
// other's answer between level_build and the launch.)
static int level_build(const mpnn_bwd_member *mem, int count, BwdRec *recs, BwdLevelQ &lq, int &gkmask, int &otmask, int &total,
                       bool &smallc) {
    if (!mem || count < 1 || count > MPNN_BWD_LEVEL_MAX) return MPNN_E_ARG;
    gkmask = otmask = total = 0;
    lq.n = count;
    smallc = false;
    BwdRec scratch;
    for (int k = 0; k < count; ++k) {
        const mpnn_bwd_member &m = mem[k];
        if (!m.wgrad || m.wgrad->n <= 0) return MPNN_E_ARG;
        BwdRec &r = recs ? recs[k] : scratch;
        r = BwdRec{};
        int split = 1;
        const int rc = mpnn_fill_bwd_scale(m.horz, m.vert, m.wgrad, r.q, split);
        if (rc) return rc;
        const mpnn_wgrad_args *w = m.wgrad;
        r.gk = geom_kind(w->H, w->W);
        if (r.gk < 0) return MPNN_E_SHAPE;
        r.wide = (w->Cout % 64) == 0;
        const int tiles = mpnn_wgrad_tiles(w->n, w->H, w->W);
        auto clampx = [&](int g) { return g < 1 ? 1 : (g > tiles ? tiles : g); };
        BwdScaleP &q = r.q;
        q.gyh = m.horz ? q.h.Cout / 16 : 0;
        q.gyv = m.vert ? q.v.Cout / 16 : 0;
        q.gxh = m.horz ? clampx(m.wg_horz) : 0;
        q.gxv = m.vert ? clampx(m.wg_vert) : 0;
        q.h.n_tiles = q.v.n_tiles = tiles;
        q.h.xcd = q.v.xcd = q.w.c.xcd = xcd_env();
        q.gxw = split;
        q.nchw = ((q.w.c.a.C + 15) >> 4) + (q.w.c.v ? ((q.w.c.Cv + 15) >> 4) : 0);
        const int gyw = q.nchw * (w->Cout / (r.wide ? 64 : 16));
        lq.w0[k] = total;
        total += q.gyh * q.gxh + q.gyv * q.gxv + gyw * q.gxw;
        gkmask |= 1 << r.gk;
        otmask |= r.wide ? 2 : 1;
        if (!r.wide && w->a.C <= 3) smallc = true;
    }
    return 0;
}

extern "C" int mpnn_msconv_bwd_level_prepare(const mpnn_bwd_member *members, int count, void *host_records) {
    if (!host_records) return MPNN_E_ARG;
    BwdLevelQ lq = {};
    int gkmask, otmask, total;
    bool smallc;
    return level_build(members, count, (BwdRec *)host_records, lq, gkmask, otmask, total, smallc);
}

// The same level for `reps` nets of one architecture in ONE launch (co-training, lib/_co.py): members / records hold
// reps * count entries, net r's at [r * count, (r + 1) * count) -- identical shapes and workgroup budgets, the buffers of
// net r.  The caller's budget is per net (mpnn_msconv_bwd_level_slots / reps).
static int level_build_rep(const mpnn_bwd_member *mem, int count, int reps, BwdRec *recs, BwdLevelQ &lq, int &gkmask, int &otmask, int &total,
                           bool &smallc) {
    if (reps < 1) return MPNN_E_ARG;
    int rc = level_build(mem, count, recs, lq, gkmask, otmask, total, smallc);
    if (rc) return rc;
    for (int r = 1; r < reps; ++r) {
        BwdLevelQ l2 = {};
        BwdRec tmp[MPNN_BWD_LEVEL_MAX];
        int g2, o2, t2;
        bool s2;
        if ((rc = level_build(mem + r * count, count, recs ? recs + r * count : tmp, l2, g2, o2, t2, s2))) return rc;
        if (g2 != gkmask || o2 != otmask || t2 != total || s2 != smallc) return MPNN_E_ARG;      // (identical shapes: see above)
        for (int k = 0; k < count; ++k) if (l2.w0[k] != lq.w0[k]) return MPNN_E_ARG;
    }
    lq.reps = reps;  lq.wpr = total;
    // the `_rep` forms never use the XCD-aware tile order (see fwd_group_launch in conv_fwd.hip); reps = 1 is the launch of
    // one net of a group stepping by itself with the group's grids
    if (recs) for (int k = 0; k < count * reps; ++k) recs[k].q.h.xcd = recs[k].q.v.xcd = recs[k].q.w.c.xcd = 0;
    return 0;
}

extern "C" int mpnn_msconv_bwd_level_prepare_rep(const mpnn_bwd_member *members, int count, int reps, void *host_records) {
    if (!host_records) return MPNN_E_ARG;
    BwdLevelQ lq = {};
    int gkmask, otmask, total;
    bool smallc;
    return level_build_rep(members, count, reps, (BwdRec *)host_records, lq, gkmask, otmask, total, smallc);
}

extern "C" int mpnn_msconv_bwd_level_rep(const mpnn_bwd_member *members, int count, int reps, const void *dev_records, void *stream) {
    if (!dev_records) return MPNN_E_ARG;
    BwdLevelQ lq = {};
    int gkmask, otmask, total;
    bool smallc;
    const int rc = level_build_rep(members, count, reps, nullptr, lq, gkmask, otmask, total, smallc);
    if (rc) return rc;
    LevelKern kern = level_kernel(gkmask, otmask, smallc);
    if (!kern) return MPNN_E_SHAPE;
    hipLaunchKernelGGL(kern, dim3(total * reps), dim3(256), 0, (hipStream_t)stream, (const BwdRec *)dev_records, lq);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_msconv_bwd_level(const mpnn_bwd_member *members, int count, const void *dev_records, void *stream) {
    if (!dev_records) return MPNN_E_ARG;
    BwdRec recs[MPNN_BWD_LEVEL_MAX];
    BwdLevelQ lq = {};
    int gkmask, otmask, total;
    bool smallc;
    const int rc = level_build(members, count, recs, lq, gkmask, otmask, total, smallc);
    if (rc) return rc;
    LevelKern kern = level_kernel(gkmask, otmask, smallc);
    if (!kern) return MPNN_E_SHAPE;
    hipLaunchKernelGGL(kern, dim3(total), dim3(256), 0, (hipStream_t)stream, (const BwdRec *)dev_records, lq);
    MPNN_LAUNCH_CHECK();
    return 0;
}
