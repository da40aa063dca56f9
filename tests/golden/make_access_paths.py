"""Generates tests/golden/access_paths.json: the dictionary ACCESS PATHS the reference's consumers
take into the files this build writes -- data extracted with ``ast`` from the reference tree (run in
the build container, where /root/reference exists), no source text:

  stats : subscript chains with constant keys in scripts/make-routing-hists, make-acc-eff-plots,
          make-nlds, make-pres-figs, make-videos (readers of nets/<expt>/<i>-stats.npy and
          <i>-stats/<t>.npy, written by scripts/train-nets:149-153 from scripts/lib/desc.py:24-36)
  net   : the record keys scripts/lib/serdes.py:21-60 reads back (decode_layer, load_params, decode_net)

    python tests/golden/make_access_paths.py     # rewrites access_paths.json
"""
import ast
import json
import os

REF = '/root/reference/scripts'
HERE = os.path.dirname(os.path.abspath(__file__))


def chains(path, min_len=1):
    tree = ast.parse(open(path, encoding='utf-8').read())
    inner = set()
    out = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Subscript):
            keys, cur = [], node
            while isinstance(cur, ast.Subscript):
                sl = cur.slice
                if isinstance(sl, ast.Constant) and isinstance(sl.value, (str, int)):
                    keys.append(sl.value)
                else:
                    keys.append(None)
                if cur is not node:
                    inner.add(id(cur))
                cur = cur.value
            keys.reverse()
            if isinstance(cur, ast.Name):
                out.add((id(node), cur.id, tuple(keys)))
    # keep maximal chains only, all-constant, containing at least one string key
    res = set()
    for nid, root, keys in out:
        if nid in inner or None in keys or len(keys) < min_len:
            continue
        if any(isinstance(k, str) for k in keys):
            res.add((root, keys))
    return sorted(res, key=str)


def main():
    stats = {}
    for name in ('make-routing-hists', 'make-acc-eff-plots', 'make-nlds', 'make-pres-figs', 'make-videos'):
        known = {'root', 'sinks', 'stats_ts', 'stats_tr', 'name', 'type'}
        picked = []
        for root, keys in chains(os.path.join(REF, name)):
            if any(k in known for k in keys if isinstance(k, str)):
                picked.append({'var': root, 'keys': list(keys)})
        stats[name] = picked
    net = [{'var': r, 'keys': list(k)} for r, k in chains(os.path.join(REF, 'lib', 'serdes.py'))
           if r in ('record', 'desc')]
    json.dump({'stats': stats, 'net': net}, open(os.path.join(HERE, 'access_paths.json'), 'w'), indent=1, ensure_ascii=False)
    print({k: len(v) for k, v in stats.items()}, len(net))


if __name__ == '__main__':
    main()
