// mpnn_msconv_fwd: one scale of MultiscaleConvMax forward (see conv_kernel.h).
#include "conv_kernel.h"

extern "C" int mpnn_msconv_fwd(const mpnn_conv_fwd_args *a, void *stream) {
    if (!a || !a->a.x || !a->wa_pack || !a->out || !a->bias) return MPNN_E_ARG;
    if (a->v && !a->wv_pack) return MPNN_E_ARG;
    ConvP p = {};
    p.a = a->a;
    p.v = a->v;  p.Cv = a->v ? a->Cv : 0;
    p.wa = a->wa_pack;  p.wv = a->wv_pack;
    p.n = a->n;  p.H = a->H;  p.W = a->W;  p.Cout = a->Cout;
    p.bias = a->bias;  p.out = a->out;  p.out_sum = a->out_sum;
    p.out_nslot = a->out_nslot < 1 ? 1 : (a->out_nslot > MPNN_BN_SLOTS ? MPNN_BN_SLOTS : a->out_nslot);
    return conv_launch<EPI_FWD>(p, (hipStream_t)stream);
}
