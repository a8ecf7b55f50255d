import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the oracle's float64 CPU passes are many small ops: on the GPU box (128 intra-op threads by default) the full-size C2
    # parity test takes 47 s with 64 threads and 20 s with 32 (measured) - the checker, not the product, is what this speeds up
    import torch
    torch.set_num_threads(min(32, os.cpu_count() or 1))


# PBSED_EMULATE=1: the `-m gpu` tests on a machine WITHOUT a GPU - the library's entry points served by the emulated translation units
# of tests/emu (tests/emu/cpu_device.py), the tests' device 'cpu'.  Orders of magnitude slower than the GPU (a tool for the times the
# GPU pool is closed, e.g. `PBSED_EMULATE=1 pytest tests/test_gpu_ops.py -m gpu -n 4`; tests that time kernels or need the real
# device's memory model have nothing to say there); never on by default, and pointless where a GPU is.
EMULATE = os.environ.get('PBSED_EMULATE') == '1'


@pytest.fixture(scope='session', autouse=EMULATE)
def _emulated_device(tmp_path_factory):
    from tests.emu import cpu_device
    mp = pytest.MonkeyPatch()
    with cpu_device.emulated_device(mp, cpu_device.EmulatedLibrary(tmp_path_factory.mktemp('emu_whole'))) as library:
        yield library
    mp.undo()


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    if EMULATE:
        for item in items:
            if getattr(item.module, 'DEV', None) == 'cuda:0':
                item.module.DEV = 'cpu'
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load
