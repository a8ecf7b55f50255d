"""Randomised sweep of the GPU post-processing kernels against the oracle's restatement of the reference filters
(bit-exact for the median filter, 1e-15 for the f64 boundaries filter): random [clips, classes, frames] shapes and
per-class filter lengths incl. lengths above the sequence length.  Usage: fuzz_postproc.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import postproc as pp
from pb_sed_amd import ops, inference as inf

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to('cuda')
bad = 0
for case in range(n_cases):
    n, k, t = int(rng.integers(1, 9)), int(rng.choice([1, 3, 10, 17])), int(rng.choice([1, 2, 5, 47, 100, 500, 1001]))
    x = rng.random((n, k, t)).astype(np.float32)
    if rng.random() < .3:
        x = np.round(x * 8) / 8                      # ties
    med = rng.choice([1, 3, 5, 11, 41, 101, 301], size=k)
    step = rng.choice([0, 2, 4, 10, 20, 60], size=k)
    tag = f'case {case}: [{n},{k},{t}]'
    try:
        a = inf.filtering(dev(x), ops.medfilt, med).cpu().numpy()
        b = pp.filtering(x.copy(), pp.medfilt, med)
        ok1 = np.array_equal(a, b)
        a2 = inf.filtering(dev(x), ops.boundariesfilt, step).cpu().numpy()
        b2 = pp.filtering(x.copy(), pp.boundariesfilt, step)
        ok2 = np.allclose(a2, b2, rtol=0, atol=1e-6)
        n1 = int(med[0])
        ok3 = np.array_equal(ops.medfilt(dev(x), n1).cpu().numpy(), pp.medfilt(x.copy(), n1))
        ok = ok1 and ok2 and ok3
        bad += not ok
        print(tag, 'med', med.tolist(), 'step', step.tolist(), 'ok' if ok else f'BAD med={ok1} bnd={ok2} scalar={ok3}')
    except Exception as ex:
        bad += 1
        print(tag, 'EXCEPTION', type(ex).__name__, str(ex)[:140])
print('bad cases:', bad, 'of', n_cases)
