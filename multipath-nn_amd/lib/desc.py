"""Network descriptors: dataset-averaged statistics in the reference's schema.

Counterpart of ``scripts/lib/desc.py``: ``net_desc`` returns
``{type, stats_tr, stats_ts, root: {name, stats_tr, stats_ts, sinks: [...]}}`` with every
statistic of ``state_tensors`` (scripts/train-nets:117-130) averaged over the FULL training
and test sets in evaluation mode (desc.py:10-36), so the reference's figure scripts can
read the ``*-stats.npy`` files this build writes.  ``render_net_desc`` is a plain-text
summary (cosmetic; not byte-compatible with the reference's box drawing).
"""
import numpy as np

__all__ = ['net_desc', 'render_net_desc']


def mean_net_state(net, data, hypers, routed=False):
    """One evaluation pass per batch; per-sample statistics are summed on the device and
    divided by the sample count at the end (desc.py:10-22).  In 'ev' mode every sample is
    independent, so the batch size only sets how much work a launch carries: the passes feed
    thousands of images at a time (the reference's 128 would be 469 latency-bound passes).

    routed=True evaluates each block only on the samples routed to it (Net.eval): acc, moc, p_cor,
    p_inc and the *_by_cls statistics -- all the reference's figure scripts read -- are unchanged;
    the log-only per-leaf c_err / p_tr and per-switch x_rte then average over the samples that
    REACH the node (0 for the others) instead of over all samples."""
    sums, count = None, 0
    for x0, y in data:
        net.eval({net.x0: x0, net.y: y, **hypers}, routed=routed)
        sums_of = getattr(net.engine(), 'state_sums', None)
        if sums_of is not None:                 # (the multiscale engine: the sums in a dozen batched device operations)
            part = sums_of()
        else:
            part = {k: v.sum(0).double() for k, v in net.state().items()}
        sums = part if sums is None else {k: sums[k] + part[k] for k in part}
        count += len(x0)
    if sums is None:
        return {}
    return {k: (v / count).cpu().numpy().tolist() for k, v in sums.items()}


def layer_desc(ℓ, stats_tr, stats_ts):
    pick = lambda st: {k: v for (t, k), v in st.items() if t is ℓ}
    return {'name': ℓ.name, 'stats_tr': pick(stats_tr), 'stats_ts': pick(stats_ts),
            'sinks': [layer_desc(s, stats_tr, stats_ts) for s in ℓ.sinks]}


def net_desc(net, dataset, hypers={}, state=None, batch=4096, routed=False):
    stats_tr = mean_net_state(net, dataset.training_set(batch), hypers, routed)
    stats_ts = mean_net_state(net, dataset.test_set(batch), hypers, routed)
    pick = lambda st: {k: v for (t, k), v in st.items() if t is net}
    return {'type': type(net).__name__, 'stats_tr': pick(stats_tr), 'stats_ts': pick(stats_ts),
            'root': layer_desc(net.root, stats_tr, stats_ts)}


def _scalars(stats):
    items = sorted((k, v) for k, v in stats.items() if np.ndim(v) == 0)
    return '(' + '; '.join('%s=%.3g' % kv for kv in items) + ')' if items else ''


def _render_layer(d, key, depth):
    lines = ['%s%s %s' % ('  ' * depth, d['name'], _scalars(d[key]))]
    for s in d['sinks']:
        lines += _render_layer(s, key, depth + 1)
    return lines


def render_net_desc(desc, name='Network'):
    out = ['== %s ==' % name]
    for title, key in (('Training Set', 'stats_tr'), ('Test Set', 'stats_ts')):
        out += ['%s: [%s] %s' % (title, desc['type'], _scalars(desc[key]))]
        out += _render_layer(desc['root'], key, 1)
    return '\n'.join(out)
