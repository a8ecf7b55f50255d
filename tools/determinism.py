"""Run the same FBCRNN forward + backward N times from the same state and report, per parameter tensor, the largest
deviation of the gradient from run 0 (float atomics reorder sums: ~1e-6 relative; a race shows up orders above)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pb_sed_amd.models import weak_label
from tests.test_gpu_model import synth_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
DEV = 'cuda'
torch.manual_seed(0)
model = weak_label.CRNN.build(num_events=10).to(DEV).train()
model.feature_extractor.freeze_stats = True        # the cumulative feature statistics would differ from run to run by design
b = 32
wav, seq, weak, bnd, t = synth_batch(b, 160000, 10, ragged=True)
order = np.argsort(-seq, kind='stable')
wav, seq, weak, bnd = wav[order], seq[order], weak[order], bnd[order]
inputs = {'audio_data': wav.to(DEV), 'seq_len': seq.tolist(), 'weak_targets': weak.to(DEV), 'boundary_targets': bnd.to(DEV)}
names, offs = [], []
off = 0
for name, p in model.named_parameters():
    names.append(name); offs.append((off, off + p.numel())); off += p.numel()

def run():
    _, flat_grad = model.flat_parameters()
    flat_grad.zero_()
    for m_ in model.modules():
        if hasattr(m_, 'running_mean') and m_ is not model.feature_extractor:
            m_.running_mean.zero_(), m_.running_power.fill_(1.)
    out = model(dict(inputs))
    rev = model.review(inputs, out)
    rev['loss'].backward()
    torch.cuda.synchronize()
    return out[0].detach().clone(), out[1].detach().clone(), rev['loss'].item(), flat_grad.detach().clone()

y_f, y_b, loss, grad = run()
_, flat_grad = model.flat_parameters()
assert flat_grad.numel() == off, (flat_grad.numel(), off)
bad = 0
for it in range(1, n):
    yf, yb, l, g = run()
    rel = ((g - grad).norm() / grad.norm()).item()
    keep = torch.ones_like(grad, dtype=torch.bool)
    for name, (a0, a1) in zip(names, offs):
        if name.endswith('conv.bias'): keep[a0:a1] = False          # bias in front of a BatchNorm: true gradient 0
    rel_nb = ((g - grad)[keep].norm() / grad[keep].norm()).item()
    bias_norm = grad[~keep].norm().item()
    dy = max((yf - y_f).abs().max().item(), (yb - y_b).abs().max().item())
    flag = rel > 1e-4 or dy > 1e-5
    bad += flag
    print(f'run {it}: loss diff {l - loss:+.2e}  scores max diff {dy:.2e}  grad rel diff {rel:.2e} (without conv biases {rel_nb:.2e}; |bias grads| {bias_norm:.2e} of {grad.norm().item():.2e})' + ('   <-- deviates' if flag else ''))
    if flag:
        worst = []
        for name, (a0, a1) in zip(names, offs):
            d = (g[a0:a1] - grad[a0:a1]).norm().item() / max(grad[a0:a1].norm().item(), 1e-12)
            worst.append((d, name))
        for d, name in [w for w in sorted(worst, reverse=True) if not w[1].endswith('conv.bias')][:4]:
            print(f'      {d:.2e}  {name}')
print('deviating runs:', bad, 'of', n - 1)
