"""Scan the gfx950 ISA of every kernel for MFMA-result reads that are covered only by SALU
instructions.

Background ("MFMA -> AGPR copy hazard"): hipcc satisfies the MFMA-write -> VALU-read
hazard by counting ANY instruction between the two as a wait state.  On MI355X a run of scalar
instructions (the branch conditions of a persistent loop) does not take that long, and the
accumulator register written last by the last MFMA was read stale.  The kernels therefore end every
MFMA block with mfma_drain() (an explicit s_nop 15); this script checks, in text order, that between
each v_accvgpr_read/v_accvgpr_mov and the LAST MFMA THAT WROTE THE REGISTER IT READS there is an s_nop
or >= 11 vector-instruction slots (an MFMA in between counts as 8).

    python tools/scan_mfma_hazard.py            # compiles csrc/*.hip to ISA with hipcc -S
"""
import glob, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'multipath-nn_amd', 'csrc')


def scan(asm_path):
    """[(kernel, line, vector instructions since the write, SALU since the write)] for every accumulator
    read whose LAST WRITING MFMA is closer than 11 vector-instruction slots with no s_nop in between
    (another MFMA in between counts as 8 slots: it occupies the pipe for at least 8 passes)."""
    import re
    kern, sites = None, []
    writes = {}                      # accumulator register -> (valu, salu, nops) counters at its last MFMA write
    valu = salu = nops = 0
    dst = re.compile(r'^v_mfma\S*\s+a\[(\d+):(\d+)\]')
    for ln, raw in enumerate(open(asm_path), 1):
        t = raw.strip()
        if t.startswith('_Z') and ':' in t and ' ' not in t.split(':')[0]:
            kern, writes = t.split(':')[0], {}
        if not t or t[0] in ';.' or t.endswith(':'):
            continue
        op = t.split()[0]
        if op.startswith('v_mfma'):
            valu += 8
            m = dst.match(t)
            if m:
                for r in range(int(m.group(1)), int(m.group(2)) + 1):
                    writes[r] = (valu, salu, nops)
        elif op.startswith('v_accvgpr_read') or op.startswith('v_accvgpr_mov'):
            m = re.search(r'\ba(\d+)\b', t.split(',', 1)[1] if ',' in t else '')
            if m and int(m.group(1)) in writes:
                v0, s0, n0 = writes[int(m.group(1))]
                if nops == n0 and valu - v0 < 11:
                    sites.append((kern, ln, valu - v0, salu - s0))
            valu += 1
        elif op == 's_nop':
            nops += 1
        elif op.startswith('s_'):
            salu += 1
        else:
            valu += 1
    return sites


def main():
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(glob.glob(os.path.join(CSRC, '*.hip'))):
            out = os.path.join(tmp, os.path.basename(src) + '.s')
            subprocess.check_call(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'),
                                   '-munsafe-fp-atomics', '-mllvm', '-amdgpu-kernarg-preload-count=16', '--cuda-device-only', '-S', src, '-o', out],
                                  cwd=CSRC, stderr=subprocess.DEVNULL)
            sites = scan(out)
            print('%-16s %d suspect site(s)' % (os.path.basename(src), len(sites)))
            for k, ln, nv, ns in sites[:8]:
                print('    %s  line %d: %d VALU + %d SALU, no s_nop' % (k[:60], ln, nv, ns))
            bad += len(sites)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
