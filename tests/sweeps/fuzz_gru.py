"""Randomised (B, H, T) sweep of the two-layer forward + reversed GRU stacks (persistent granule scans and the
launch-per-step fallback) against torch.nn.GRU through the oracle wrapper: reuses the body of
tests/test_gpu_ops.py::test_gru_stack_wavefront_vs_torch.  Usage: fuzz_gru.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pytest
from tests import test_gpu_ops as T
from pb_sed_amd import ops as _ops
T.ops_check = _ops.check_gru_sync        # a failed case must not leave the time-out flag set for the next one

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    b = int(rng.choice([1, 2, 7, 15, 16, 17, 31, 32, 33, 48, 49, 64, 65, 100, 128]))   # > 32: two batch tiles per block (forward) / chain by chain
    h = int(rng.choice([64, 128, 256, 512]))
    t = int(rng.choice([1, 2, 3, 9, 40, 130]))
    ragged = bool(rng.random() < .6) and t > 1
    persist = '2' if rng.random() < .75 else '0'
    mp = pytest.MonkeyPatch()
    try:
        T.test_gru_stack_wavefront_vs_torch(b, h, t, ragged, persist, mp)
        print(f'case {case}: B{b} H{h} T{t} ragged={int(ragged)} persist={persist} ok')
    except Exception as ex:
        bad += 1
        print(f'case {case}: B{b} H{h} T{t} ragged={int(ragged)} persist={persist} FAILED {type(ex).__name__}: {str(ex)[:160]}')
    finally:
        mp.undo()
        try:
            T.ops_check()
        except Exception:
            pass
print('failed cases:', bad, 'of', n_cases)
