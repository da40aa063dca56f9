import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'multipath-nn_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'calibrate: leaves the stream calibration of lib/_co.py on')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _deterministic_group_plans(request, monkeypatch):
    """lib/_co.py: CoGroups.plan MEASURES which streams run side by side (a timing test with spin kernels) and splits the nets
    accordingly; the tests that compare a plan's results with expected group sizes, or two plans with each other, must not
    depend on that measurement's outcome on a loaded box: they take the first streams unmeasured.  The measurement itself
    has its own test (tests/test_cotrain.py::test_stream_calibration), marked `calibrate`."""
    if 'calibrate' not in request.keywords:
        monkeypatch.setenv('MPNN_CO_CALIBRATE', '0')
