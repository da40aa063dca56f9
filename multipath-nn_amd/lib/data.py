"""Datasets and host-side augmentation.

Counterpart of the reference's ``scripts/lib/data.py``: same ``Dataset`` API, same
archive format (a pickled dict in ``arr_0`` of an ``.npz``; prep-data:60,134-136,191)
and the SAME random-number law, draw for draw, on ``numpy.random``'s global stream:
per sample ``randint(0, N)``, then ``rand() < 0.5`` for a flip if the class is
symmetric, then ``randint(-r, r + 1, 2)`` for the shift (data.py:10-34).  Seeding
numpy therefore reproduces the reference's batches bit for bit
(tests/golden/data_aug_golden.npz was produced by the reference module).

Unlike the reference's per-sample copy loop, the pixels are moved with one
vectorised gather per batch; the mean fill (data.py:19) is per image and channel.
"""
import numpy as np
import numpy.random as rand

__all__ = ['Dataset']


def _sym_of_sources(y, m_sym):
    """Per source image: is its class mirror-symmetric (data.py:27)?  A list, for fast scalar lookups."""
    return np.asarray(m_sym)[np.argmax(y, axis=1)].tolist()


def _draw_augmentation(n, n_src, y, m_sym, r_shift, sym_src=None):
    """The reference's RNG call sequence (data.py:24-34), without touching pixels: per sample
    randint(0, N), then rand() if the class is symmetric, then randint(-r, r + 1, 2) -- the same calls
    in the same order on numpy's global stream.  (The loop is the input pipeline's serial part:
    ~0.3 ms per batch of 128 with the class lookups hoisted out of it.)"""
    if sym_src is None:
        sym_src = _sym_of_sources(y, m_sym)
    randint, rnd = rand.randint, rand.rand
    lo, hi = -r_shift, r_shift + 1
    j = [0] * n; flip = [False] * n; sh = [None] * n
    for i in range(n):
        ji = randint(0, n_src)
        j[i] = ji
        if sym_src[ji]:
            flip[i] = not (rnd() < 0.5)               # rand_flip keeps `a` when rand() < 0.5
        sh[i] = randint(lo, hi, 2)
    return np.array(j, np.int64), np.array(flip, bool), np.array(sh, np.int64).reshape(n, 2)


def augmented_batch(x0, y, n, m_sym, r_shift):
    j, flip, sh = _draw_augmentation(n, len(x0), y, m_sym, r_shift)
    h, w = x0.shape[1:3]
    src = x0[j].astype(np.float64)
    src[flip] = src[flip][:, :, ::-1]
    u = np.arange(h)[None, :] + sh[:, 0:1]             # b[u, v] = a[u + du, v + dv]
    v = np.arange(w)[None, :] + sh[:, 1:2]
    ok = ((u >= 0) & (u < h))[:, :, None] & ((v >= 0) & (v < w))[:, None, :]
    out = src[np.arange(n)[:, None, None], np.clip(u, 0, h - 1)[:, :, None], np.clip(v, 0, w - 1)[:, None, :]]
    fill = src.mean(axis=(1, 2))
    out = np.where(ok[..., None], out, fill[:, None, None, :])
    return out, y[j].astype(np.float64)


def batch(x0, y, n):
    i = rand.randint(0, len(x0), n)
    return np.take(x0, i, axis=0), np.take(y, i, axis=0)


def full_set(x0, y, n):
    for i in range(0, len(x0), n):
        yield x0[i:i + n], y[i:i + n]


class Dataset:
    def __init__(self, path=None, arrays=None):
        archive = arrays if arrays is not None else np.load(path, allow_pickle=True)['arr_0'][()]
        self.x0_tr, self.x0_ts = archive['x0_tr'], archive['x0_ts']
        self.y_tr, self.y_ts = archive['y_tr'], archive['y_ts']
        self.m_sym = archive['m_sym']
        self.x0_vl, self.y_vl = self.x0_tr[:0], self.y_tr[:0]

    @classmethod
    def synthetic(cls, n_tr=1024, n_ts=256, shape=(32, 32, 3), n_cls=10, seed=0):
        """Stand-in with the statistics the benchmark uses (uniform pixels, uniform labels)."""
        g = np.random.default_rng(seed)
        mk = lambda n: (g.random((n,) + tuple(shape), dtype=np.float32),
                        np.eye(n_cls, dtype=np.float32)[g.integers(0, n_cls, n)])
        (x_tr, y_tr), (x_ts, y_ts) = mk(n_tr), mk(n_ts)
        return cls(arrays=dict(x0_tr=x_tr, y_tr=y_tr, x0_ts=x_ts, y_ts=y_ts, m_sym=np.ones(n_cls, bool)))

    @property
    def x0_shape(self):
        return self.x0_tr.shape[1:]

    @property
    def y_shape(self):
        return self.y_tr.shape[1:]

    def augmented_training_batch(self, n=128, r_shift=4):
        return augmented_batch(self.x0_tr, self.y_tr, n, self.m_sym, r_shift)

    # ---- device-resident path (mpnn_augment_batch): the dataset is uploaded once, a step costs a
    # 2 KB upload of the draws and one gather launch; the draws are the reference's, call for call.
    def to_device(self, device='cuda:0'):
        import torch
        self._dev = device
        self._x_dev = torch.from_numpy(np.ascontiguousarray(self.x0_tr, dtype=np.float32)).to(device)
        self._y_dev = torch.from_numpy(np.ascontiguousarray(self.y_tr, dtype=np.float32)).to(device)
        self._x_ts_dev = torch.from_numpy(np.ascontiguousarray(self.x0_ts, dtype=np.float32)).to(device)
        self._y_ts_dev = torch.from_numpy(np.ascontiguousarray(self.y_ts, dtype=np.float32)).to(device)
        self._draw_host = None
        return self

    def augmented_training_batch_device(self, n=128, r_shift=4, x_out=None, y_out=None):
        """Same batch as augmented_training_batch (same numpy RNG draws), assembled on the GPU in fp32.
        Returns (x, y) device tensors; pass x_out / y_out (e.g. the engine's input buffers) to fill them."""
        import torch
        from . import _hip
        if getattr(self, '_x_dev', None) is None:
            raise _hip.HipError('Dataset.to_device() first: the augmentation kernel gathers from device memory')
        lib = _hip.load()
        if getattr(self, '_sym_src', None) is None:
            self._sym_src = _sym_of_sources(self.y_tr, self.m_sym)
        j, flip, sh = _draw_augmentation(n, len(self.x0_tr), self.y_tr, self.m_sym, r_shift, self._sym_src)
        draw = np.stack([j, flip.astype(np.int64), sh[:, 0], sh[:, 1]], 1).astype(np.int32)
        d = torch.from_numpy(draw).to(self._dev, non_blocking=False)
        h, w, c = self.x0_tr.shape[1:]
        n_cls = self.y_tr.shape[1]
        if x_out is None:
            x_out = torch.empty((n, h, w, c), device=self._dev)
        if y_out is None:
            y_out = torch.empty((n, n_cls), device=self._dev)
        _hip.check(lib.mpnn_augment_batch(self._x_dev.data_ptr(), self._y_dev.data_ptr(), d.data_ptr(),
                                          x_out.data_ptr(), y_out.data_ptr(), n, h, w, c, n_cls,
                                          torch.cuda.current_stream().cuda_stream), 'augment_batch')
        self._keep = d                                  # until the stream has consumed it
        return x_out, y_out

    def training_batch(self, n=128):
        return batch(self.x0_tr, self.y_tr, n)

    def test_batch(self, n=128):
        return batch(self.x0_ts, self.y_ts, n)

    def training_set(self, n=128):
        """Full training set in batches of n (scripts/lib/data.py:72-73); device slices once
        to_device() has made the set resident (no re-upload per statistics pass)."""
        if getattr(self, '_x_dev', None) is not None:
            yield from full_set(self._x_dev, self._y_dev, n)
        else:
            yield from full_set(self.x0_tr, self.y_tr, n)

    def test_set(self, n=128):
        if getattr(self, '_x_ts_dev', None) is not None:
            yield from full_set(self._x_ts_dev, self._y_ts_dev, n)
        else:
            yield from full_set(self.x0_ts, self.y_ts, n)
