#!/usr/bin/env python3
"""Where do the vector instructions of the forward conv body go?  Single-op forward convs (mpnn_msconv_fwd, BatchNorm-on-load in
batch mode, statistics + pooling epilogue) over a grid of (images, input channels, map size): under `rocprofv3 --pmc
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES` the executed instruction counts are linear in the number of workgroups,
tiles and units,  VALU = P x workgroups + T x tiles + U x units  (per WAVE: a workgroup is four) -- a least-squares fit over
the launches gives the prologue, per-tile and per-unit cost of the body.

    python tools/valu_regression.py run            # launches the convs (under rocprofv3), prints one line per launch
    python tools/valu_regression.py fit <dir>      # joins the counter CSV with that list (dispatch order) and fits
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multipath-nn_amd'), os.path.join(ROOT, 'tests')]
import numpy as np

CASES = [(H, n, cin, cout) for H in (8, 4, 16) for cin in (16, 32, 64, 128) for n in (256, 512, 1024) for cout in (16,)
         if not (H == 16 and cin > 32)]


def geometry(H, n, cin, cout):
    tiles = n * (H // 16) * (H // 4) if H >= 16 else (n if H == 8 else (n + 3) // 4)
    gy = cout // 16
    cap = max(1024 // gy, 64)
    gx = min(tiles, cap)
    upt = cin // 16
    return gx * gy, tiles * gy, tiles * gy * upt


if sys.argv[1] == 'run':
    import torch
    import hiputil as U
    from lib import _hip
    rng = np.random.default_rng(0)
    for H, n, cin, cout in CASES:
        x = rng.standard_normal((n, H, H, cin)).astype(np.float32)
        w = (rng.standard_normal((3, 3, cin, cout)) * 0.1).astype(np.float32)
        b = np.zeros(cout, np.float32)
        bn, cnt = U.bn_dict(x, np.ones(cin, np.float32), np.zeros(cin, np.float32))
        U.conv_fwd(x, w, b, bn=bn, mode=_hip.ACT_BN_BATCH, bn_cnt=cnt, want_pool=H > 4)
        print('case\t%d\t%d\t%d\t%d\t%d\t%d\t%d' % ((H, n, cin, cout) + geometry(H, n, cin, cout)), flush=True)
else:
    import csv, glob, re
    root = sys.argv[2]
    cases = [tuple(int(v) for v in ln.split('\t')[1:]) for ln in open(os.path.join(root, 'cases.txt')) if ln.startswith('case')]
    rows = {}
    for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'conv_k<' in r['Kernel_Name']:
                rows.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
    disp = [rows[k] for k in sorted(rows)]
    assert len(disp) == len(cases), (len(disp), len(cases))
    for geo in (8, 4, 16):
        A, yv, ys, ym = [], [], [], []
        for c, d in zip(cases, disp):
            if c[0] != geo:
                continue
            wgs, tiles, units = c[4:]
            A.append([4 * wgs, 4 * tiles, 4 * units])
            yv.append(d['SQ_INSTS_VALU'] - d['SQ_INSTS_MFMA']); ys.append(d['SQ_INSTS_SALU']); ym.append(d['SQ_INSTS_MFMA'])
        A = np.array(A, float)
        for name, y in (('vector (MFMAs not counted)', yv), ('scalar', ys), ('MFMA', ym)):
            coef, res, _, _ = np.linalg.lstsq(A, np.array(y), rcond=None)
            fit = A @ coef
            err = np.abs(fit - np.array(y)).max() / np.array(y).max()
            print('H = %2d  %-27s per wave: prologue %8.1f   per tile %7.1f   per unit %7.1f    (worst residual %.1f %%)' % (geo, name, coef[0], coef[1], coef[2], 100 * err))
