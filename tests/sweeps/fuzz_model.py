"""Randomised (batch, clip length) sweep of one FBCRNN / tag-conditioned BiCRNN train step (tiny net configuration) against the oracle on the
CPU: features, both score tensors, loss and the gradient in the L2 sense.  Clip lengths give odd frame counts (T not a
multiple of 4 or of the kernels' tiles).  Usage: fuzz_model.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import frontend as ofe, models as om
from pb_sed_amd.models import weak_label, strong_label
from tests import test_gpu_model as T

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
DEV = 'cuda'
bad = 0
for case in range(n_cases):
    b = int(rng.choice([1, 2, 3, 5, 8, 17]))
    n = int(rng.integers(25, 160)) * 320 + int(rng.integers(0, 320))
    if os.environ.get('FUZZ_LONG'):                # long clips: 20 .. 60 s
        n = int(rng.integers(1000, 3000)) * 320 + int(rng.integers(0, 320)); b = min(b, 3)
    hidden = int(rng.choice([64, 128]))
    ragged = bool(rng.random() < .7)
    if rng.random() < .4:                          # tag-conditioned BiCRNN (bidirectional GRU layers as one-layer scans)
        torch.manual_seed(case)
        kw = dict(num_events=10, number_of_filters=128, hidden_size=hidden, num_layers=2, net=T.TINY, tag_conditioning=True)
        ref = om.BiCRNN.build(**kw)
        model = strong_label.CRNN.build(**kw)
        T._copy_weights(model, ref)
        model.to(DEV)
        wav, seq, weak, strong, t = T.synth_batch(b, n, 10, ragged=ragged, seed=case)
        cond = (weak > .99).float()
        ref.train()
        inp_ref = {'stft': ofe.stft(wav), 'seq_len': seq.tolist(), 'weak_targets': weak, 'strong_targets': strong, 'tag_condition': cond}
        out_ref = ref(inp_ref)
        loss_ref = ref.review(inp_ref, out_ref)['loss']
        loss_ref.backward()
        model.train()
        inp = {'audio_data': wav.to(DEV), 'seq_len': seq.tolist(), 'weak_targets': weak.to(DEV), 'strong_targets': strong.to(DEV),
               'tag_condition': cond.to(DEV)}
        model.flat_parameters()[1].zero_()
        tag = f'case {case}: BiCRNN B{b} n{n} (T{t}) H{hidden} ragged={int(ragged)}'
        try:
            out = model(dict(inp))
            loss = model.review(inp, out)['loss']
            loss.backward()
            torch.cuda.synchronize()
            e_y = (out[0].cpu() - out_ref[0]).abs().max().item()
            e_loss = abs(loss.item() - loss_ref.item()) / abs(loss_ref.item())
            refp = dict(ref.named_parameters())
            g = torch.cat([p.grad.cpu().reshape(-1) for _, p in model.named_parameters()])
            gr = torch.cat([refp[nm].grad.reshape(-1) for nm, _ in model.named_parameters()])
            e_g = ((g - gr).norm() / gr.norm()).item()
            ok = e_y < 2e-4 and e_loss < 2e-4 and e_g < 2e-2
            bad += not ok
            print(tag, f'scores {e_y:.1e} loss {e_loss:.1e} grad(L2) {e_g:.1e}', '' if ok else 'BAD')
        except Exception as ex:
            bad += 1
            print(tag, 'EXCEPTION', type(ex).__name__, str(ex)[:140])
        continue
    torch.manual_seed(case)
    kw = dict(num_events=10, number_of_filters=128, hidden_size=hidden, num_layers=2, net=T.TINY)
    ref = om.FBCRNN.build(**kw)
    with torch.no_grad():
        for name, p in ref.named_parameters():
            if name.endswith('gamma'): p.uniform_(.7, 1.3)
            elif name.endswith('beta') or name.endswith('conv.bias'): p.normal_(0, .1)
        ref.feature_extractor.mean.fill_(-7.); ref.feature_extractor.inv_std.fill_(.4)
    model = weak_label.CRNN.build(**kw)
    T._copy_weights(model, ref)
    model.to(DEV)
    wav, seq, weak, bnd, t = T.synth_batch(b, n, 10, ragged=ragged, seed=case)
    ref.train()
    inputs_ref = {'stft': ofe.stft(wav), 'seq_len': seq.tolist(), 'weak_targets': weak, 'boundary_targets': bnd}
    out_ref = ref(inputs_ref)
    rev_ref = ref.review(inputs_ref, out_ref)
    rev_ref['loss'].backward()
    model.train()
    inputs = {'audio_data': wav.to(DEV), 'seq_len': seq.tolist(), 'weak_targets': weak.to(DEV), 'boundary_targets': bnd.to(DEV)}
    _, flat_grad = model.flat_parameters(); flat_grad.zero_()
    tag = f'case {case}: B{b} n{n} (T{t}) H{hidden} ragged={int(ragged)}'
    try:
        out = model(dict(inputs))
        rev = model.review(inputs, out)
        rev['loss'].backward()
        torch.cuda.synchronize()
        e_feat = (out[3].cpu() - out_ref[3]).abs().max().item()
        e_y = max((out[0].cpu() - out_ref[0]).abs().max().item(), (out[1].cpu() - out_ref[1]).abs().max().item())
        e_loss = abs(rev['loss'].item() - rev_ref['loss'].item()) / abs(rev_ref['loss'].item())
        refp = dict(ref.named_parameters())
        g = torch.cat([p.grad.cpu().reshape(-1) for _, p in model.named_parameters()])
        gr = torch.cat([refp[nm].grad.reshape(-1) for nm, _ in model.named_parameters()])
        e_g = ((g - gr).norm() / gr.norm()).item()
        ok = e_feat < 2e-4 and e_y < 2e-4 and e_loss < 2e-4 and e_g < 2e-2
        bad += not ok
        print(tag, f'features {e_feat:.1e} scores {e_y:.1e} loss {e_loss:.1e} grad(L2) {e_g:.1e}', '' if ok else 'BAD')
    except Exception as ex:
        bad += 1
        print(tag, 'EXCEPTION', type(ex).__name__, str(ex)[:140])
print('bad cases:', bad, 'of', n_cases)
