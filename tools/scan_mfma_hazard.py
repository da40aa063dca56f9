"""Scan the gfx950 ISA of every kernel for MFMA-result reads that are covered only by SALU
instructions.

Background ("MFMA -> AGPR copy hazard"): hipcc satisfies the MFMA-write -> VALU-read
hazard by counting ANY instruction between the two as a wait state.  On MI355X a run of scalar
instructions (the branch conditions of a persistent loop) does not take that long, and the
accumulator register written last by the last MFMA was read stale.  The kernels therefore end every
MFMA block with mfma_drain() (an explicit s_nop 15); this script checks, in text order, that between
each v_mfma and the next v_accvgpr_read/v_accvgpr_mov there is an s_nop or >= 11 vector
instructions.

    python tools/scan_mfma_hazard.py            # compiles csrc/*.hip to ISA with hipcc -S
"""
import glob, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'multipath-nn_amd', 'csrc')


def scan(asm_path):
    kern, last, sites = None, None, []
    n_valu = n_salu = 0
    nop = False
    for ln, raw in enumerate(open(asm_path), 1):
        t = raw.strip()
        if t.startswith('_Z') and ':' in t and ' ' not in t.split(':')[0]:
            kern, last = t.split(':')[0], None
        if not t or t[0] in ';.' or t.endswith(':'):
            continue
        op = t.split()[0]
        if op.startswith('v_mfma'):
            last, n_valu, n_salu, nop = ln, 0, 0, False
        elif last is not None:
            if op.startswith('v_accvgpr_read') or op.startswith('v_accvgpr_mov'):
                if not nop and n_valu < 11:
                    sites.append((kern, ln, n_valu, n_salu))
                last = None
            elif op == 's_nop':
                nop = True
            elif op.startswith('s_'):
                n_salu += 1
            else:
                n_valu += 1
    return sites


def main():
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(glob.glob(os.path.join(CSRC, '*.hip'))):
            out = os.path.join(tmp, os.path.basename(src) + '.s')
            subprocess.check_call(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'),
                                   '-munsafe-fp-atomics', '--cuda-device-only', '-S', src, '-o', out],
                                  cwd=CSRC, stderr=subprocess.DEVNULL)
            sites = scan(out)
            print('%-16s %d suspect site(s)' % (os.path.basename(src), len(sites)))
            for k, ln, nv, ns in sites[:8]:
                print('    %s  line %d: %d VALU + %d SALU, no s_nop' % (k[:60], ln, nv, ns))
            bad += len(sites)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
