"""Static check of the persistent GRU scans' exchange protocol (CPU; SURVEY.md section 5: race detection).

The scans of ``pb_sed_amd/csrc/gru_stack.hip`` hand their step outputs from workgroup to workgroup as tagged words in a
workspace; a wrong index map does not crash, it times out or feeds a consumer another element's word.
``oracle/scan_protocol_check.cpp`` compiles the very text the kernels compile (``gru_granule_map.h`` /
``gru_granule_role.inc``) with g++ and enumerates it: one publisher per word, every (chain, layer, step, row, unit) has a
word, every polled word is the element the consumer's MFMA operand order assumes, a ring's workgroups share an XCD, every
XCD can hold its share of the grid, nothing leaves the workspace.  Reference op site of the scans:
pb_sed/models/weak_label/crnn.py:61-67 (torch.nn.GRU inside padertorch's wrapper).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'pb_sed_amd', 'csrc')
SRC = os.path.join(ROOT, 'oracle', 'scan_protocol_check.cpp')


def _build(out, header_dir=CSRC):
    """g++ the checker against the index maps under ``header_dir`` (the tree's, or a mutated copy)."""
    src = open(SRC).read().replace('#include "../pb_sed_amd/csrc/gru_granule_map.h"', '#include "gru_granule_map.h"')
    tmp_src = out + '.cpp'
    with open(tmp_src, 'w') as f:
        f.write(src)
    subprocess.run(['g++', '-O2', '-std=c++17', '-I', header_dir, '-o', out, tmp_src], check=True)
    return out


@pytest.fixture(scope='module')
def checker(tmp_path_factory):
    return _build(str(tmp_path_factory.mktemp('scanchk') / 'scan_protocol_check'))


def _run(binary, *args):
    r = subprocess.run([binary] + [str(a) for a in args], capture_output=True, text=True)
    return r.returncode, r.stdout.strip()


# (direction, chains, layers, B, H, tiles per block): the shapes the configurations launch, and ragged ones
SHAPES = [
    ('fwd', 2, 2, 32, 256, 1), ('bwd', 2, 2, 32, 256, 1),            # BASELINE configs[1]: FBCRNN, two 2-layer stacks
    ('fwd', 2, 1, 32, 256, 1), ('bwd', 2, 1, 32, 256, 1),            # configs[2]: the BiGRU layers of the BiCRNN
    ('fwd', 2, 2, 64, 256, 2), ('fwd', 6, 1, 64, 256, 2),            # configs[4]: 64 clips, two tiles per block; three detectors' layer
    ('bwd', 1, 2, 64, 256, 1),                                        # BPTT above 32 clips: chain by chain
    ('fwd', 2, 2, 32, 512, 2), ('bwd', 1, 2, 32, 512, 1),            # net_config 'deep'
    ('fwd', 2, 2, 16, 512, 1),                                        # the shape that falls back to the 3-D grid (fuzz_gru.py)
    ('fwd', 2, 2, 1, 64, 1), ('bwd', 2, 3, 7, 128, 1), ('fwd', 1, 4, 17, 128, 1), ('fwd', 2, 2, 33, 256, 2), ('bwd', 2, 2, 20, 64, 1),
]


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: '-'.join(str(x) for x in s))
def test_exchange_maps_are_consistent(checker, shape):
    d, nc, nl, b, h, nb = shape
    rc, out = _run(checker, d, nc, nl, b, h, 3, nb, 256)
    assert rc == 0 and out.startswith('OK'), out
    if (d, nc, nl, b, h) == ('fwd', 2, 2, 32, 256):
        assert '192 blocks (1-D XCD-aware grid, 24 per XCD of 32 CUs)' in out          # 192 co-resident workgroups


def test_a_ring_pinned_to_a_full_xcd_falls_back_to_the_3d_grid(checker):
    rc, out = _run(checker, 'fwd', 2, 2, 16, 512, 3, 1, 256)      # 2 x 2 rings of 32 blocks + projections: 40 slots per XCD > 32 CUs
    assert rc == 0 and '3-D grid' in out, out


def test_a_scan_that_cannot_be_resident_is_refused(checker):
    rc, out = _run(checker, 'fwd', 2, 2, 32, 512, 3, 1, 256)      # 384 blocks on 256 CUs: the launcher takes two tiles per block instead
    assert rc == 1 and 'co-resident' in out, out


MUTANTS = [
    # (file, original text, mutated text, what breaks)
    ('gru_granule_map.h', '(k0 / 16) * 256 + lr * 16 + lq * 4', '(k0 / 16) * 256 + lr * 16 + lq * 8', 'a lane polls another unit'),
    ('gru_granule_map.h', '(size_t)nb * (H / 16) * 256 + (tid & 255)', '(size_t)nb * (H / 16) * 128 + (tid & 255)', "a block's second tile overwrites words"),
    ('gru_granule_map.h', '((size_t)TILE0 * (H / 16) + bx) * 256', '((size_t)TILE0 * (H / 16) + bx) * 128', 'two producers share words'),
    ('gru_granule_role.inc', 'unit = (slot / nj) * 8 + x;', 'unit = (slot / nj) * 8 + ((x + slot) & 7);', 'a ring leaves its XCD'),
]


@pytest.mark.parametrize('mutant', MUTANTS, ids=[m[3] for m in MUTANTS])
def test_the_checker_finds_broken_maps(tmp_path, mutant):
    """The checker is only worth something if it fails on a wrong map: four single-token mutations of the shared text."""
    fname, old, new, _ = mutant
    d = tmp_path / 'csrc'
    d.mkdir()
    for f in ('gru_granule_map.h', 'gru_granule_role.inc'):
        shutil.copy(os.path.join(CSRC, f), d / f)
    text = (d / fname).read_text()
    assert text.count(old) >= 1
    (d / fname).write_text(text.replace(old, new))
    binary = _build(str(tmp_path / 'mutant'), str(d))
    results = [_run(binary, *s[:5], 3, s[5], 256) for s in SHAPES[:6]]
    assert any(rc == 1 and out.startswith('FAIL') for rc, out in results), results


def test_the_kernels_compile_the_checked_text():
    """gru_stack.hip must take its role map and exchange indices from the shared text, not from a private copy."""
    src = open(os.path.join(CSRC, 'gru_stack.hip')).read()
    assert '#include "gru_granule_role.inc"' in src and '#include "gru_granule_map.h"' in src
    for macro, n in (('PBSED_GM_RING_BASE(', 2), ('PBSED_GM_POLL_OFFSET0(', 2), ('PBSED_GM_RING_WORD', 3)):
        assert src.count(macro) >= n, macro
    # no second, hand-written form of the indices left behind
    assert not re.search(r'\(k0 / 16\) \* 256 \+ lr \* 16', src)
    assert 'slot / nj' not in src
