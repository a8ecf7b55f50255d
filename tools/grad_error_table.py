"""Per-tensor gradient error of one 'shallow' FBCRNN train step (B clips of 10 s) against the float64 CPU oracle, next to
the error of the fp32 CPU oracle itself, under several kernel selections (env switches are read once per process, so each
variant runs in its own subprocess).  Usage: python tools/grad_error_table.py [B]   (needs the GPU)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CACHE = '/tmp/pbsed_grad_oracle.pt'


KIND = os.environ.get('GRAD_TABLE_MODEL', 'fbcrnn')


def oracle(b):
    import copy
    import torch
    from oracle import frontend as ofe, models as om
    from tests.test_gpu_configs import _randomise, _sorted_batch
    torch.manual_seed(0)
    ref = om.FBCRNN.build(num_events=10) if KIND == 'fbcrnn' else om.BiCRNN.build(num_events=10, tag_conditioning=True)
    _randomise(ref)
    state = copy.deepcopy(ref.state_dict())
    ref64 = copy.deepcopy(ref).double().train()
    wav, seq, weak, bnd, t = _sorted_batch(b, 160000, seed=21)
    ref.train()
    tkey = 'boundary_targets' if KIND == 'fbcrnn' else 'strong_targets'
    inp = {'stft': ofe.stft(wav), 'seq_len': seq.tolist(), 'weak_targets': weak, tkey: bnd, 'tag_condition': (weak > .99).float()}
    out = ref(inp)
    ref.review(inp, out)['loss'].backward()
    in64 = {'stft': ofe.stft(wav).double(), 'seq_len': seq.tolist(), 'weak_targets': weak.double(), tkey: bnd.double(),
            'tag_condition': (weak > .99).double()}
    out64 = ref64(in64)
    ref64.review(in64, out64)['loss'].backward()
    torch.save({'state': state, 'wav': wav, 'seq': seq, 'weak': weak, 'bnd': bnd,
                'g32': {n: p.grad for n, p in ref.named_parameters()}, 'g64': {n: p.grad for n, p in ref64.named_parameters()},
                'y32': out[0].detach(), 'y64': out64[0].detach()}, CACHE)


def variant():
    import torch
    from pb_sed_amd.models import strong_label, weak_label
    d = torch.load(CACHE, weights_only=False)
    model = weak_label.CRNN.build(num_events=10) if KIND == 'fbcrnn' else strong_label.CRNN.build(num_events=10, tag_conditioning=True)
    model.load_state_dict(d['state'])
    model.to('cuda:0').train()
    tkey = 'boundary_targets' if KIND == 'fbcrnn' else 'strong_targets'
    inp = {'audio_data': d['wav'].cuda(), 'seq_len': d['seq'].tolist(), 'weak_targets': d['weak'].cuda(), tkey: d['bnd'].cuda(),
           'tag_condition': (d['weak'] > .99).float().cuda()}
    model.flat_parameters()[1].zero_()
    out = model(dict(inp))
    model.review(inp, out)['loss'].backward()
    torch.cuda.synchronize()
    print(f"scores: |hip-cpu64| {(out[0].cpu().double() - d['y64']).abs().max():.2e}  |cpu32-cpu64| {(d['y32'].double() - d['y64']).abs().max():.2e}")
    rows = []
    for n, p in model.named_parameters():
        g64 = d['g64'][n]
        sc = g64.abs().max().item()
        if sc < 1e-9:
            continue
        e = (p.grad.cpu().double() - g64).abs().max().item() / sc
        e32 = (d['g32'][n].double() - g64).abs().max().item() / sc
        l2 = ((p.grad.cpu().double() - g64).norm() / g64.norm()).item()
        l2_32 = ((d['g32'][n].double() - g64).norm() / g64.norm()).item()
        where = tuple(int(v) for v in torch.unravel_index((p.grad.cpu().double() - g64).abs().argmax(), g64.shape))
        rows.append((e / max(e32, 1e-9), n, e, e32, l2, l2_32, where, tuple(g64.shape)))
    rows.sort(reverse=True)
    for r, n, e, e32, l2, l2_32, where, shape in rows[:20]:
        print(f'  {n:42s} max-rel {e:.2e} (cpu32 {e32:.2e}, x{r:4.1f})   L2 {l2:.2e} (cpu32 {l2_32:.2e})  at {where} of {shape}')
    print(f'  tensors with err > max(2e-3, 2 x cpu32): {sum(1 for r, n, e, e32, *_ in rows if e > max(2e-3, 2 * e32))} of {len(rows)}')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--variant':
        variant()
    else:
        b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
        oracle(b)
        for env in ({}, {'PBSED_CONV_WINO': '0', 'PBSED_WGRAD_WINO': '0'}, {'PBSED_GRU_PERSIST': '0'}):
            print('==== variant', env or 'default', flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), '--variant'], env=dict(os.environ, **env), cwd=ROOT)
