"""Three or four Trainer steps of the real-width FBCRNN beside the oracle + torch.optim.Adam: loss / gradient-norm trajectory and how
many parameter elements are half an Adam step apart after NSTEP steps (the figures quoted in tests/test_gpu_model.py).
Usage (GPU box): NSTEP=3 python tools/micro/traj_debug.py"""
import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
torch.set_num_threads(32)
from oracle import frontend as ofe, models as om
from pb_sed_amd.models import weak_label
from pb_sed_amd.trainer import Trainer
from tests.test_gpu_model import _copy_weights, synth_batch
DEV='cuda:0'
import os
NSTEP=int(os.environ.get('NSTEP','4'))
for lr in (5e-4,):
    torch.manual_seed(1)
    ref = om.FBCRNN.build(num_events=10); model = weak_label.CRNN.build(num_events=10)
    _copy_weights(model, ref); model.to(DEV)
    wav, seq, weak, bnd, t = synth_batch(8, 160000, 10, seed=5)
    inputs_ref = {'stft': ofe.stft(wav), 'seq_len': seq.tolist(), 'weak_targets': weak, 'boundary_targets': bnd}
    inputs = {'audio_data': wav.to(DEV), 'seq_len': seq.tolist(), 'weak_targets': weak.to(DEV), 'boundary_targets': bnd.to(DEV)}
    clip=1.5
    opt = torch.optim.Adam(ref.parameters(), lr=lr); trainer = Trainer(model, lr=lr, gradient_clipping=clip); ref.train()
    for step in range(NSTEP):
        opt.zero_grad(); rev_ref = ref.review(inputs_ref, ref(inputs_ref)); rev_ref['loss'].backward()
        norm_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), clip); opt.step()
        rev = trainer.step(inputs)
        print(f'lr {lr} step {step}: loss {rev["loss"].item():.6f} / {rev_ref["loss"].item():.6f}  norm {rev["scalars"]["grad_norm"].item():.5f} / {norm_ref.item():.5f}')
    refp = dict(ref.named_parameters()); worst=[]
    for name, p in model.named_parameters():
        if refp[name].grad.abs().max().item() < 1e-6: continue
        diff = (p.detach().cpu() - refp[name].detach()).abs()
        worst.append(((diff > 0.5*lr).float().mean().item(), diff.mean().item()/lr, name))
    worst.sort(reverse=True)
    print('lr', lr, 'worst frac>0.5lr, mean/lr:', [(round(a,4), round(b,4), n) for a,b,n in worst[:6]])
