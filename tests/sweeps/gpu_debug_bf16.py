"""Where does the bf16 mode leave the bf16-operand oracle (oracle/bf16emu.py)?  Layer by layer: the raw input of every conv
layer of the HIP run (engine.DECISION_TAP) against the same tensor of the oracle, then the GRU outputs and the logits.
Usage: gpu_debug_bf16.py [batch] [seconds]"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from oracle import bf16emu, frontend as ofe, models as om
from pb_sed_amd import engine
from pb_sed_amd.models import strong_label
from tests.test_gpu_configs import _bicrnn_inputs, _randomise, _sorted_batch
from tests.test_gpu_model import _copy_weights

b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.
torch.manual_seed(2)
ref = om.BiCRNN.build(num_events=10, tag_conditioning=True)
_randomise(ref, 2)
model = strong_label.CRNN.build(num_events=10, tag_conditioning=True)
_copy_weights(model, ref)
model.to('cuda:0').train()
model.conv_precision = 'bf16'
model.keep_logits = True
wav, seq, weak, strong, t = _sorted_batch(b, int(16000 * secs), seed=31)
for mode in ('emu', 'f64'):
    emu = copy.deepcopy(ref).double().train()
    if mode == 'emu':
        bf16emu.enable(emu)
    caps = {}
    from oracle.nn import _ConvLayer
    for name, m in emu.named_modules():
        if isinstance(m, _ConvLayer) and m.pre:          # pre-activation layers: the layer's output is its (pooled) conv output
            m.register_forward_hook(lambda mod, i, o, name=name: caps.__setitem__(name, o.detach()))
    cap_rnn = {}
    emu.rnn.register_forward_hook(lambda mod, i, o: cap_rnn.__setitem__('logit', o[0].detach()))
    in64 = _bicrnn_inputs(wav, seq, weak, strong, dtype=torch.float64)
    emu(in64)

    class Tap:
        def __init__(self):
            self.rows = []
            self.names = {m: n for n, m in model.named_modules()}

        def append(self, e):
            if e[0] == 'layer':
                self.rows.append((self.names[e[1]], e[3].detach().clone()))
    tap = Tap()
    engine.DECISION_TAP = tap
    inp = _bicrnn_inputs(wav, seq, weak, strong, 'cuda:0')
    model(dict(inp))
    engine.DECISION_TAP = None
    torch.cuda.synchronize()
    print(f'--- HIP bf16 mode against the {"bf16-operand oracle" if mode == "emu" else "float64 oracle"}')
    names = [n for n, _ in tap.rows]
    for i, (name, x) in enumerate(tap.rows):
        if i == 0 or names[i - 1].split('.convs.')[0] != name.split('.convs.')[0]:
            continue                                     # first layer of a stack: its input is not a conv output of that stack
        prev = names[i - 1]
        if prev not in caps:
            continue
        want = caps[prev]
        got = x.cpu().double().reshape(want.shape)
        m = (torch.arange(want.shape[-1])[None] < torch.as_tensor(seq)[:, None]).reshape([b] + [1] * (want.dim() - 2) + [-1])
        e = ((got - want) * m).abs().max().item() / want.abs().max().item()
        print(f'  output of {prev:34s} max-abs err / max {e:.2e}')
    logit = model.last_logits[0].cpu().double()
    m = (torch.arange(t)[None] < torch.as_tensor(seq)[:, None])[:, None, :]
    print(f'  logits: {((logit - cap_rnn["logit"]) * m).abs().max().item():.2e} (|logit| max {cap_rnn["logit"].abs().max().item():.2f})')
