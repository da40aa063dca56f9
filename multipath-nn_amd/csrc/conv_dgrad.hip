// Input-gradient convolutions of a MultiscaleConvMax scale (see conv_kernel.h).
#include "conv_kernel.h"

// `g` holds dz: apply the BatchNorm backward (mpnn_bn_bwd_apply) while staging operand A.
static int fill_g_ctx(const mpnn_bn_ctx *c, ConvP &p) {
    if (!c) return 0;
    if (!c->s || !c->red) return MPNN_E_ARG;
    p.ga_on = 1;  p.ga_s = c->s;  p.ga_bn = c->bn;  p.ga_red = c->red;
    p.ga_nslot = c->red_nslot < 1 ? 1 : c->red_nslot;
    return 0;
}

int mpnn_fill_dgrad_horz(const mpnn_dgrad_horz_args *a, ConvP &p) {
    if (!a || !a->g || !a->w_pack || !a->out) return MPNN_E_ARG;
    p.a.x = a->g;  p.a.C = a->Cg;  p.a.mode = MPNN_ACT_IDENTITY;  p.a.shift = 0;
    p.wa = a->w_pack;
    p.n = a->n;  p.H = a->H;  p.W = a->W;  p.Cout = a->Cout;
    p.extra = a->dy_extra;  p.out = a->out;  p.acc_out = a->accumulate ? 1 : 0;
    if (a->prev) {
        if (!a->prev->s || !a->red_out) return MPNN_E_ARG;
        p.sprev = a->prev->s;  p.pbn = a->prev->bn;  p.red_out = a->red_out;
        p.out_nslot = a->prev->red_nslot < 1 ? 1 : a->prev->red_nslot;
    }
    return fill_g_ctx(a->g_ctx, p);
}

int mpnn_fill_dgrad_vert(const mpnn_dgrad_vert_args *a, ConvP &p) {
    if (!a || !a->g || !a->w_pack || !a->fine || !a->fine->s || !a->dz_g_fine) return MPNN_E_ARG;
    p.a.x = a->g;  p.a.C = a->Cg;  p.a.mode = MPNN_ACT_IDENTITY;  p.a.shift = 0;
    p.wa = a->w_pack;
    p.n = a->n;  p.H = a->H;  p.W = a->W;  p.Cout = a->Cout;
    p.out = a->dz_g_fine;  p.sprev = a->fine->s;  p.pbn = a->fine->bn;
    p.red = a->fine_has_dz ? a->fine->red : nullptr;  p.has_dz = a->fine_has_dz;
    p.red_nslot = a->fine->red_nslot < 1 ? 1 : a->fine->red_nslot;
    return fill_g_ctx(a->g_ctx, p);
}

// Both input gradients of one scale (they read the same g) in ONE launch.
int mpnn_trace_install_dgrad(void *buf) { return mpnn_trace_install(buf); }

extern "C" int mpnn_msconv_dgrad_pair(const mpnn_dgrad_horz_args *h, const mpnn_dgrad_vert_args *v, void *stream) {
    ConvP ph = {}, pv = {};
    int rc = mpnn_fill_dgrad_horz(h, ph);
    if (rc) return rc;
    rc = mpnn_fill_dgrad_vert(v, pv);
    if (rc) return rc;
    if (!h->prev) return MPNN_E_ARG;               // the paired form always carries the producer's BN backward
    return conv_launch_pair<EPI_DGH_BN, EPI_DGV>(ph, pv, (hipStream_t)stream);
}

extern "C" int mpnn_msconv_dgrad_horz(const mpnn_dgrad_horz_args *a, void *stream) {
    ConvP p = {};
    const int rc = mpnn_fill_dgrad_horz(a, p);
    if (rc) return rc;
    if (a->prev) return conv_launch<EPI_DGH_BN>(p, (hipStream_t)stream);
    return conv_launch<EPI_DGH_RAW>(p, (hipStream_t)stream);
}

extern "C" int mpnn_msconv_dgrad_vert(const mpnn_dgrad_vert_args *a, void *stream) {
    ConvP p = {};
    const int rc = mpnn_fill_dgrad_vert(a, p);
    if (rc) return rc;
    return conv_launch<EPI_DGV>(p, (hipStream_t)stream);
}
