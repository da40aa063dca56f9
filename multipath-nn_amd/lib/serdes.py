"""Network serialisation in the reference's record format.

Counterpart of ``scripts/lib/serdes.py``: ``write_net`` stores ``np.save`` of the nested dict
``{type, root: {type, name, hypers, params{name: ndarray}, sinks, comps, router}, hypers,
params}`` (serdes.py:13-19,40-44); BatchNorm moving averages travel with the parameters
(layer_types.py:229-230).  Unlike the reference (whose ``read_net`` nothing calls and which
drops the optimizer state), ``write_net(..., with_optimizer=True)`` also stores the momentum
accumulators so training can resume exactly.
"""
import numpy as np

import lib.layer_types
import lib.net_types

__all__ = ['encode_net', 'decode_net', 'write_net', 'read_net']


def encode_layer(layer, accum=False):
    if layer is None:
        return None
    rec = dict(type=type(layer).__name__, name=layer.name, hypers=dict(vars(layer.hypers)),
               params={k: v.numpy() for k, v in vars(layer.params).items()},
               sinks=[encode_layer(s, accum) for s in layer.sinks],
               comps=[encode_layer(c, accum) for c in layer.comps],
               router=encode_layer(layer.router, accum))
    if accum:
        rec['accum'] = {k: v.accum.cpu().numpy().reshape(v.shape) for k, v in vars(layer.params).items()
                        if v.trainable}
    return rec


def decode_layer(rec):
    if rec is None:
        return None
    cls = getattr(lib.layer_types, rec['type'])
    comps = [] if rec['type'] == 'MultiscaleBatchNorm' else [decode_layer(c) for c in rec['comps']]
    return cls(name=rec['name'], router=decode_layer(rec['router']),
               sinks=[decode_layer(s) for s in rec['sinks']], comps=comps, **rec['hypers'])


def load_params(layer, rec):
    if layer is None:
        return
    load_params(layer.router, rec['router'])
    for ℓ, r in zip(layer.comps, rec['comps']):
        load_params(ℓ, r)
    for ℓ, r in zip(layer.sinks, rec['sinks']):
        load_params(ℓ, r)
    for k, v in rec['params'].items():
        getattr(layer.params, k).assign(v)
    for k, v in rec.get('accum', {}).items():
        p = getattr(layer.params, k)
        import torch
        p.accum.copy_(torch.as_tensor(np.asarray(v, np.float32).reshape(-1)).to(p.accum.device))


def encode_net(net, with_optimizer=False):
    net.engine()
    return dict(type=type(net).__name__, root=encode_layer(net.root, with_optimizer),
                hypers=dict(vars(net.hypers)), params={})


def decode_net(rec):
    net = getattr(lib.net_types, rec['type'])(root=decode_layer(rec['root']), **rec['hypers'])
    net.engine()
    load_params(net.root, rec['root'])
    return net


def write_net(path, net, with_optimizer=False):
    np.save(path, encode_net(net, with_optimizer))


def read_net(path):
    return decode_net(np.load(path, allow_pickle=True)[()])
