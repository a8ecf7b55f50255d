"""Randomised (T, B, G, K) sweep of the batched GRU weight-gradient GEMMs (producer / consumer kernel, slots or atomics, one or
several residency rounds, row / column tails) against fp64: reuses tests/test_gpu_ops.py::test_gru_wgrad_vs_torch, which runs
three GEMMs of widths K, K, 2 K with the three row shifts in one launch.  Usage: fuzz_gru_wgrad.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import test_gpu_ops as T

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    t = int(rng.choice([1, 2, 5, 33, 64, 127, 500]))
    b = int(rng.choice([1, 3, 16, 31, 32, 48]))
    g = int(rng.choice([4, 12, 100, 192, 384, 768, 1536]))
    k = int(rng.choice([4, 8, 36, 128, 252, 256, 512]))
    # bf16 operands: the test's tolerance is 2e-2 sqrt(T B) for the MAXIMUM over G x K entries - too tight for a handful of rows
    precision = 'f32' if rng.random() < .7 or t * b < 64 else 'bf16'
    try:
        T.test_gru_wgrad_vs_torch(t, b, g, k, precision)
        print(f'case {case}: T{t} B{b} G{g} K{k} {precision} ok')
    except Exception as ex:
        bad += 1
        print(f'case {case}: T{t} B{b} G{g} K{k} {precision} FAILED {type(ex).__name__}: {str(ex)[:160]}')
print('failed cases:', bad, 'of', n_cases)
