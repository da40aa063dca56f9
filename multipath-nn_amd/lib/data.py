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

__all__ = ['Dataset', 'DrawStream']


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


def _draw_augmentation_fast(n, n_src, sym_u8, r_shift, out=None, all_sym=False):
    """The same draws as _draw_augmentation -- same values, same position of numpy's global stream afterwards --
    without the per-sample Python calls: RAW 32-bit outputs are pulled from numpy's own generator
    (randint(0, 2**32, size, uint32) returns consecutive MT19937 words) and the library's host function
    mpnn_draw_augmentation consumes them exactly as randint / rand would (numpy's legacy masked-rejection bounded
    integers, random_sample's two-word double).  Rejection sampling makes the number of words a batch needs
    unknowable in advance, and the stream must not be over-drawn (other code draws from it between two batches):
    each round fetches exactly the MINIMUM the remaining draws need, the function consumes all of it and reports the
    new minimum (640 -> 240 -> 88 -> ... words for a batch of 128).  ~20 us per batch instead of 0.3-0.4 ms.
    Returns an int32 [n, 4] array (j, flip, du, dv): the record mpnn_augment_batch reads."""
    import ctypes as C
    from . import _hip
    lib = _hip.load()
    if out is None:
        out = np.empty((n, 4), np.int32)
    state = np.zeros(4, np.int64)
    state[3] = 1 if all_sym else 0                    # (every source is mirror-symmetric: two more words per sample are certain)
    p_sym, p_out, p_state = (a.ctypes.data_as(C.c_void_p) for a in (sym_u8, out, state))
    need = lib.mpnn_draw_augmentation(None, 0, n, n_src, p_sym, r_shift, p_out, p_state)
    while need > 0:
        raw = rand.randint(0, 2 ** 32, size=need, dtype=np.uint32)
        need = lib.mpnn_draw_augmentation(raw.ctypes.data_as(C.c_void_p), need, n, n_src, p_sym, r_shift, p_out, p_state)
    if need < 0:
        raise _hip.HipError('mpnn_draw_augmentation: status %d' % need)
    return out


class DrawStream:
    """A private copy of numpy's legacy random stream (MT19937: key[624] + position) that only ever serves augmentation
    draws, stepped by the library's host function (mpnn_draw_augmentation_mt): no Python per sample, any number of
    batches per call.  `DrawStream(seed)` starts where `numpy.random.seed(seed)` starts; `skip(batches, ...)` advances
    over whole batches.  `serial_positions` hands every net of an experiment the stream position it has in the
    reference's serial loop (scripts/train-nets:159-164: ONE stream, the nets one after another), so that nets trained
    side by side train on exactly the batches they see there."""
    def __init__(self, seed=None, state=None):
        st = np.random.RandomState(seed).get_state() if state is None else state
        self.key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
        import ctypes as C
        self.pos = C.c_int(int(st[2]))

    def copy(self):
        return DrawStream(state=('MT19937', self.key, self.pos.value))

    def state(self):
        """As numpy.random.RandomState.set_state takes it."""
        return ('MT19937', self.key.copy(), int(self.pos.value), 0, 0.0)

    def draw(self, n, n_src, sym_u8, r_shift, out=None, batches=1):
        """`batches` batches of n records (j, flip, du, dv) into out ([batches, n, 4] or [n, 4] int32; None: skip them)."""
        import ctypes as C
        from . import _hip
        lib = _hip.load()
        if out is not None and not (out.dtype == np.int32 and out.flags.c_contiguous and out.size == batches * n * 4):
            raise ValueError('out: a C-contiguous int32 array of batches * n * 4 entries')
        rc = lib.mpnn_draw_augmentation_mt(self.key.ctypes.data_as(C.c_void_p), C.byref(self.pos), batches, n, n_src,
                                           None if sym_u8 is None else sym_u8.ctypes.data_as(C.c_void_p), r_shift,
                                           None if out is None else out.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise _hip.HipError('mpnn_draw_augmentation_mt: status %d' % rc)
        return out

    skip = lambda self, batches, n, n_src, sym_u8, r_shift: self.draw(n, n_src, sym_u8, r_shift, None, batches)


def augmented_batch(x0, y, n, m_sym, r_shift, draws=None):
    """draws: an [n, 4] record array (j, flip, du, dv) from _draw_augmentation_fast instead of drawing here."""
    if draws is None:
        j, flip, sh = _draw_augmentation(n, len(x0), y, m_sym, r_shift)
    else:
        j, flip, sh = draws[:, 0].astype(np.int64), draws[:, 1].astype(bool), draws[:, 2:4].astype(np.int64)
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

    # ---- device-resident path (mpnn_augment_batch): the dataset is uploaded once; a step costs the draws (the
    # reference's, call for call, replayed by the library's host function: ~20 us), a 2 KB asynchronous upload through
    # a ring of pinned buffers and one gather launch -- which, once bind_engine() has installed it, is part of the
    # training step's hipGraph.
    RING = 8
    SLOTS = 8           # record slots per consumer: Engine.STEPS_MAX training steps in one hipGraph

    def to_device(self, device='cuda:0'):
        import torch
        self._dev = device
        self._x_dev = torch.from_numpy(np.ascontiguousarray(self.x0_tr, dtype=np.float32)).to(device)
        self._y_dev = torch.from_numpy(np.ascontiguousarray(self.y_tr, dtype=np.float32)).to(device)
        self._x_ts_dev = torch.from_numpy(np.ascontiguousarray(self.x0_ts, dtype=np.float32)).to(device)
        self._y_ts_dev = torch.from_numpy(np.ascontiguousarray(self.y_ts, dtype=np.float32)).to(device)
        self._sym_u8 = np.ascontiguousarray(_sym_of_sources(self.y_tr, self.m_sym), dtype=np.uint8)
        self._all_sym = bool(self._sym_u8.all())
        self._bufs = {}
        return self

    def _draw_buffers(self, n, eng=None):
        """The static device record buffer + its ring of pinned upload buffers for one consumer (eng None: the dataset's
        own; an engine bound with bind_engine has its own set, so that several engines -- the nets of a co-trained
        group -- each read the batch staged for THEM).  An engine's set is stored ON the engine (per dataset), so it
        lives exactly as long as its consumer: a table keyed by id(eng) outlived collected engines and could hand a NEW
        engine (CPython reuses ids) the buffers of a dead one."""
        import torch
        from . import _hip
        if getattr(self, '_x_dev', None) is None:
            raise _hip.HipError('Dataset.to_device() first: the augmentation kernel gathers from device memory')
        if eng is None:
            if self.__dict__.get('_bound_engines'):
                raise ValueError('an engine is bound to this dataset (bind_engine): stage its batches with '
                                 'stage_training_draws(n, eng=eng) -- the dataset\'s own record buffer is not what its step reads')
            holder = self.__dict__.setdefault('_bufs', {})
            key = None
        else:
            holder = eng.__dict__.setdefault('_draw_bufs', {})
            key = id(self)
        b = holder.get(key)
        if b is None or b['ring'][0][0].shape[0] < n:
            # dev: SLOTS record buffers -- slot j feeds step j of a K-step graph (Engine.run_steps), slot 0 the one-step graph
            b = holder[key] = dict(ring=[(torch.zeros((n, 4), dtype=torch.int32).pin_memory(), None) for _ in range(self.RING * self.SLOTS)],
                                   slot=-1, dev=torch.zeros((self.SLOTS, n, 4), dtype=torch.int32, device=self._dev))
        return b

    def stage_training_draws(self, n=128, r_shift=4, eng=None, slot=0):
        """Draw one batch's augmentation records -- (j, flip, du, dv) per sample, the reference's numpy.random call
        sequence (scripts/lib/data.py:24-34) -- and queue their upload into the static device buffer the augmentation
        launch reads (eng: the buffer of that bound engine; slot: for step `slot` of a K-step graph).  Asynchronous: a ring of pinned host buffers, each reused only
        after the event behind its last copy has completed (under hipGraph replay the host runs several steps ahead of the
        stream)."""
        import torch
        b = self._draw_buffers(n, eng)
        k = b['slot'] = (b['slot'] + 1) % len(b['ring'])
        buf, ev = b['ring'][k]
        if ev is not None:
            ev.synchronize()
        _draw_augmentation_fast(n, len(self.x0_tr), self._sym_u8, r_shift, out=buf.numpy()[:n], all_sym=self._all_sym)
        b['dev'][slot, :n].copy_(buf[:n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        b['ring'][k] = (buf, ev)
        return b['dev'][slot]

    def stage_training_draws_k(self, K, n=128, r_shift=4, eng=None, stream=None, between=None):
        """stage_training_draws for the K steps of one K-step graph replay (record slots 0 .. K-1, drawn in step order from
        the one numpy stream) with ONE upload: a small copy on the compute stream in front of a replay costs ~8 us.
        between(j) runs right after step j's draws (the caller's own per-iteration draws keep their place in the stream)."""
        import torch
        b = self._draw_buffers(n, eng)
        if K > self.SLOTS:
            raise ValueError('at most %d record slots' % self.SLOTS)
        if 'ringk' not in b:
            b['ringk'], b['slotk'] = [(torch.zeros((self.SLOTS, n, 4), dtype=torch.int32).pin_memory(), None) for _ in range(self.RING)], -1
        k = b['slotk'] = (b['slotk'] + 1) % len(b['ringk'])
        buf, ev = b['ringk'][k]
        if ev is not None:
            ev.synchronize()
        out = buf.numpy()
        if stream is not None and out.shape[1] == n and between is None:
            stream.draw(n, len(self.x0_tr), self._sym_u8, r_shift, out=out[:K], batches=K)
        else:
            for j in range(K):
                if stream is not None:
                    stream.draw(n, len(self.x0_tr), self._sym_u8, r_shift, out=out[j, :n])
                else:
                    _draw_augmentation_fast(n, len(self.x0_tr), self._sym_u8, r_shift, out=out[j, :n], all_sym=self._all_sym)
                if between is not None:
                    between(j)             # (whatever else the loop draws per iteration, in the loop's order: train-adaptive-nets' k_cpt choice)
        b['dev'][:K, :n].copy_(buf[:K, :n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        b['ringk'][k] = (buf, ev)

    def _augment_launch(self, n, x_out, y_out, stream, draws=None):
        from . import _hip
        h, w, c = self.x0_tr.shape[1:]
        draws = self._draw_buffers(n)['dev'][0] if draws is None else draws
        _hip.check(_hip.load().mpnn_augment_batch(self._x_dev.data_ptr(), self._y_dev.data_ptr(), draws.data_ptr(),
                                                  x_out.data_ptr(), y_out.data_ptr(), n, h, w, c, self.y_tr.shape[1], stream),
                   'augment_batch')

    def augmented_training_batch_device(self, n=128, r_shift=4, x_out=None, y_out=None):
        """Same batch as augmented_training_batch (same numpy RNG draws), assembled on the GPU in fp32.
        Returns (x, y) device tensors; pass x_out / y_out (e.g. the engine's input buffers) to fill them."""
        import torch
        self.stage_training_draws(n, r_shift)
        h, w, c = self.x0_tr.shape[1:]
        if x_out is None:
            x_out = torch.empty((n, h, w, c), device=self._dev)
        if y_out is None:
            y_out = torch.empty((n, self.y_tr.shape[1]), device=self._dev)
        self._augment_launch(n, x_out, y_out, torch.cuda.current_stream().cuda_stream)
        return x_out, y_out

    def bind_engine(self, eng, n=128):
        """Make the batch assembly the FIRST launch of the engine's training step (part of its hipGraph): every
        net.train.run then gathers the batch described by the latest stage_training_draws() straight into the
        engine's input buffers.  Returns the (x0, y) feed values (markers that name the engine's buffers)."""
        from ._plan import BoundInput
        if getattr(self, '_x_dev', None) is None:
            self.to_device(str(eng.dev))
        eng.ensure_capacity(n)
        # the engine's own draw buffer and upload ring, allocated WITHOUT drawing: a draw here would consume a batch of the
        # numpy stream and offset every later batch against the reference's call sequence (scripts/lib/data.py:24-34).
        # Stage every step's records with stage_training_draws(n, eng=eng).
        draws = self._draw_buffers(n, eng)['dev']
        self.__dict__['_bound_engines'] = self.__dict__.get('_bound_engines', 0) + 1
        # the engine's buffers are resolved when the launch is issued (eagerly or into a capture; the engine drops its
        # graphs whenever it reallocates them), never held as views: see _plan.BoundInput
        eng.set_prologue(lambda stream: self._augment_launch(n, eng.x0[:n], eng.y[:n], stream, draws[0]),
                         lambda stream, j: self._augment_launch(n, eng.x0[:n], eng.y[:n], stream, draws[j]))
        return BoundInput(eng, 'x0', n), BoundInput(eng, 'y', n)

    def bind_cotrainer(self, co, n=128):
        """bind_engine for every net of a co-trained group (lib/_co.py), plus ONE gather launch for the whole group at the
        head of the joint step (mpnn_augment_batch_multi: K x n workgroups instead of K launches of n) reading ONE
        record buffer [K, n, 4] that stage_cotrainer_draws fills with one upload per step.  Returns the list of (x0, y)
        feed values, one pair per net."""
        import torch
        from . import _hip
        bound = [self.bind_engine(e, n) for e in co.engs]
        K = len(co.engs)
        h, w, c = self.x0_tr.shape[1:]
        # dev: SLOTS record buffers [K, n, 4] -- slot j feeds step j of a K-step joint graph (CoTrainer.run_steps), slot 0
        # the one-step graph.  The group's buffers live on the co-trainer (as an engine's on the engine).
        g = dict(n=n, dev=torch.zeros((self.SLOTS, K, n, 4), dtype=torch.int32, device=self._dev), slot=-1,
                 ring=[(torch.zeros((self.SLOTS, K, n, 4), dtype=torch.int32).pin_memory(), None) for _ in range(self.RING)])
        co.__dict__.setdefault('_draw_group', {})[id(self)] = g
        state = {}

        def launch_slot(stream, j):
            key = tuple(e.generation for e in co.engs)              # (the engines' input buffers may have been reallocated)
            if state.get('key') != key:
                tabs = []
                for jj in range(self.SLOTS):
                    recs = []
                    for k, e in enumerate(co.engs):
                        d = _hip.AugmentDst()
                        d.draw, d.x_out, d.y_out = g['dev'][jj, k].data_ptr(), e.x0.data_ptr(), e.y.data_ptr()
                        recs.append(d)
                    tabs.append(_hip.to_device_table(recs, self._dev))
                state['key'], state['tabs'] = key, tabs
            _hip.check(_hip.load().mpnn_augment_batch_multi(self._x_dev.data_ptr(), self._y_dev.data_ptr(), state['tabs'][j].data_ptr(),
                                                             K, n, h, w, c, self.y_tr.shape[1], stream),
                       'augment_batch_multi')
        co.set_prologue(lambda stream: launch_slot(stream, 0), launch_slot)
        return bound

    def serial_positions(self, positions, iters, n=128, r_shift=4, seed=0):
        """One DrawStream per entry of `positions` (ascending): where the reference's serial experiment loop (one numpy
        stream seeded with `seed`, `iters` batches of n per net, net after net) starts its net number positions[q].  The
        nets in between are skipped by the library at a few us per batch."""
        if getattr(self, '_sym_u8', None) is None:
            self._sym_u8 = np.ascontiguousarray(_sym_of_sources(self.y_tr, self.m_sym), dtype=np.uint8)
        s, at, out = DrawStream(seed), 0, []
        for q in positions:
            if q < at:
                raise ValueError('positions must ascend')
            s.skip(iters * (q - at), n, len(self.x0_tr), self._sym_u8, r_shift)
            at = q
            out.append(s.copy())
        return out

    def stage_cotrainer_draws(self, co, r_shift=4, streams=None):
        """One step's augmentation records of EVERY net of a bound group: the reference's draws, net after net, from the one
        numpy stream -- or, with streams (one DrawStream per net), every net from its own; one asynchronous upload for the
        whole group."""
        self.stage_cotrainer_draws_k(co, 1, r_shift, streams)

    def stage_cotrainer_draws_k(self, co, S, r_shift=4, streams=None):
        """stage_cotrainer_draws for the S steps of one K-step joint graph replay (record slots 0 .. S-1) with ONE upload.
        Draw order: step after step, within a step net after net (the order of S single-step calls); with streams every
        net draws its S batches from its own DrawStream -- net r's j-th batch is the j-th batch it sees in the serial loop."""
        import torch
        g = co.__dict__['_draw_group'][id(self)]
        if S > self.SLOTS:
            raise ValueError('at most %d record slots' % self.SLOTS)
        k = g['slot'] = (g['slot'] + 1) % len(g['ring'])
        buf, ev = g['ring'][k]
        if ev is not None:
            ev.synchronize()
        out = buf.numpy()
        for j in range(S):
            for r in range(out.shape[1]):
                if streams is not None:            # (net r's own stream: DrawStream, e.g. its position in the serial loop)
                    streams[r].draw(g['n'], len(self.x0_tr), self._sym_u8, r_shift, out=out[j, r])
                else:
                    _draw_augmentation_fast(g['n'], len(self.x0_tr), self._sym_u8, r_shift, out=out[j, r], all_sym=self._all_sym)
        g['dev'][:S].copy_(buf[:S], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        g['ring'][k] = (buf, ev)

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
