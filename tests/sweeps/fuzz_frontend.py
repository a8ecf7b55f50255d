"""Randomised clip-length sweep of the fused STFT -> mel -> log front-end against the oracle (reuses
tests/test_gpu_ops.py::test_logmel_vs_oracle): lengths from less than one frame shift to 12.5 s, batch 1..3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import test_gpu_ops as T

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    n = int(rng.choice([rng.integers(1, 1200), rng.integers(1200, 20000), rng.integers(20000, 200000)]))
    b = int(rng.integers(1, 4))
    try:
        T.test_logmel_vs_oracle(n, b)
        print(f'case {case}: n={n} b={b} ok')
    except Exception as ex:
        bad += 1
        print(f'case {case}: n={n} b={b} FAILED {type(ex).__name__}: {str(ex)[:150]}')
print('failed cases:', bad, 'of', n_cases)
