"""Short fixed-seed runs of the randomised shape sweeps in tests/sweeps/ (separate processes: the library reads its switches
once per process and a failed scan must not leak state into other tests)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('tool,args', [('fuzz_conv.py', ['100', '11']), ('fuzz_gru.py', ['16', '12']),
                                       ('fuzz_postproc.py', ['48', '13']), ('fuzz_frontend.py', ['24', '14']),
                                       ('fuzz_model.py', ['20', '15']), ('fuzz_gru_wgrad.py', ['60', '16'])])
def test_randomised_sweep(tool, args):
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS', '32'))      # the float64 references: see conftest.py
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'sweeps', tool), *args], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=env)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-2000:]
    flagged = [line for line in text.splitlines() if ' BAD' in line or 'EXCEPTION' in line or 'FAILED' in line]
    assert not flagged, '\n'.join(flagged[:10])
