"""Generates tests/golden/data_aug_golden.npz by IMPORTING the reference's only importable
module (scripts/lib/data.py, pure NumPy) in the build container.  Run from the repo root:
    python tests/golden/make_data_golden.py
The fixture holds inputs and expected outputs only (no reference source)."""
import importlib.util
import os

import numpy as np

REF = '/root/reference/scripts/lib/data.py'
spec = importlib.util.spec_from_file_location('ref_data', REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

g = np.random.default_rng(0)
x0 = g.random((24, 8, 8, 3))
y = np.eye(4)[g.integers(0, 4, 24)]
m_sym = np.array([True, False, True, False])
out = {'x0': x0, 'y': y, 'm_sym': m_sym}
for k, (seed, n, r) in enumerate([(123, 16, 2), (7, 9, 4), (99, 5, 0)]):
    np.random.seed(seed)
    xb, yb = ref.augmented_batch(x0, y, n, m_sym, r)
    out['case%d_args' % k] = np.array([seed, n, r])
    out['case%d_x' % k], out['case%d_y' % k] = xb, yb
# KA7: rand_shift on a 4x4 ramp with seed 1
np.random.seed(1)
out['ka7'] = ref.rand_shift(np.arange(16.0).reshape(4, 4, 1), 1)
np.random.seed(5)
out['batch_idx_x'], out['batch_idx_y'] = ref.batch(x0, y, 7)
np.savez_compressed(os.path.join(os.path.dirname(__file__), 'data_aug_golden.npz'), **out)
print('wrote', {k: np.shape(v) for k, v in out.items()})
