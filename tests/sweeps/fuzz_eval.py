"""Randomised sweep of the FBCRNN inference heads (tagging, boundaries detection, windowed sound event detection with
scalar and per-class window lengths) against the oracle in eval mode: random batch, clip length, window length / shift."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import frontend as ofe, models as om
from pb_sed_amd.models import weak_label
from tests import test_gpu_model as T

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DEV = 'cuda'
bad = 0
for case in range(n_cases):
    b = int(rng.choice([1, 2, 4, 7])); n = int(rng.integers(20, 120)) * 320 + int(rng.integers(0, 320))
    hidden = int(rng.choice([64, 128]))
    torch.manual_seed(case)
    kw = dict(num_events=10, number_of_filters=128, hidden_size=hidden, num_layers=2, net=T.TINY)
    ref = om.FBCRNN.build(**kw).eval()
    with torch.no_grad():
        for name, buf in ref.named_buffers():
            if name.endswith('running_mean'): buf.normal_(0, .2)
            elif name.endswith('running_power'): buf.uniform_(.8, 1.5)
        ref.feature_extractor.mean.fill_(-7.); ref.feature_extractor.inv_std.fill_(.4)
    model = weak_label.CRNN.build(**kw)
    T._copy_weights(model, ref)
    model.to(DEV).eval()
    wav, seq, *_ = T.synth_batch(b, n, 10, seed=case)
    inp_ref = {'stft': ofe.stft(wav), 'seq_len': seq.tolist()}
    inp = {'audio_data': wav.to(DEV), 'seq_len': seq.tolist()}
    wl, ws = int(rng.choice([3, 5, 9, 21])), int(rng.choice([1, 2, 4]))
    per_class = [[int(v) for v in rng.choice([3, 5, 9], 10)], [int(v) for v in rng.choice([5, 7], 10)]]
    tag = f'case {case}: B{b} n{n} H{hidden} window {wl}/{ws}'
    try:
        with torch.no_grad():
            worst = 0.
            for method, kwargs in [('tagging', {}), ('boundaries_detection', {}),
                                   ('sound_event_detection', dict(window_length=wl, window_shift=ws)),
                                   ('sound_event_detection', dict(window_length=per_class, window_shift=1))]:
                y_ref, sl_ref = getattr(ref, method)(dict(inp_ref), **kwargs)
                y, sl = getattr(model, method)(dict(inp), **kwargs)
                assert np.array_equal(sl, sl_ref) and y.shape == y_ref.shape, (method, y.shape, y_ref.shape)
                worst = max(worst, (y.cpu() - y_ref).abs().max().item())
        ok = worst < 1e-4
        bad += not ok
        print(tag, f'max |dy| {worst:.1e}', '' if ok else 'BAD')
    except Exception as ex:
        bad += 1
        print(tag, 'EXCEPTION', type(ex).__name__, str(ex)[:150])
print('bad cases:', bad, 'of', n_cases)
