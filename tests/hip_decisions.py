"""The discrete decisions of a HIP training pass (ReLU masks, (2,1)-pool rows, the weak loss's max selector) in the form
``oracle/decisions.py::impose`` takes, so that the float64 oracle differentiates the branch the HIP run took.

``with tap_decisions(model) as tap: out = model(inputs)`` collects them through ``pb_sed_amd.engine.DECISION_TAP``:
every kernel of the library decides a ReLU as ``fmaf(x, scale, shift) > 0`` on the layer's raw fp32 input and the fp32
per-channel scale / shift that ``bn_finalize`` wrote; the sign of an fma equals the sign of the exact ``x * scale + shift``,
and a float64 evaluation of that expression (24 x 24-bit product exact, one rounding of the sum) has that sign too - the
masks below are the kernels' decisions, not an approximation of them.  Pool rows are the argmax bytes the forward kernels
stored for their backward pass.
"""
import contextlib

import torch


class _Tap:
    """Receives engine.DECISION_TAP entries and turns each into decision tensors at once (the entry's buffers may be
    overwritten by later launches; the arithmetic below is enqueued on the same stream right behind the producing kernel)."""

    def __init__(self, model):
        self.names = {m: n for n, m in model.named_modules()}
        self.dec = {}

    def _mask(self, x, scale, shift):
        shape = [1, -1] + [1] * (x.dim() - 2)
        return (x.double() * scale.double().reshape(shape) + shift.double().reshape(shape)) > 0

    def append(self, e):
        if e[0] == 'layer':
            _, conv, norm, x, scale, shift, idx = e[:7]
            if norm is not None:
                owner = self.names[norm].rsplit('.', 1)[0]            # 'cnn.cnn_2d.convs.3.norm' -> the layer applying norm + ReLU
                self.dec.setdefault(owner, {})['relu'] = self._mask(x, scale, shift)
            if idx is not None:
                assert int(idx.max()) <= 1
                self.dec.setdefault(self.names[conv], {})['pool'] = idx.bool()
        elif e[0] == 'final':
            _, norm, x, scale, shift = e
            stack = self.names[norm].rsplit('.', 1)[0]                # '....cnn_1d.out_norm' -> the stack
            self.dec.setdefault(stack, {})['out_relu'] = self._mask(x, scale, shift)
        elif e[0] == 'skip':
            _, src, dst, crossed, pidx = e
            stack, i_src = self.names[src].rsplit('.convs.', 1)
            key = (int(i_src), int(self.names[dst].rsplit('.convs.', 1)[1]), int(self.names[crossed].rsplit('.convs.', 1)[1]))
            self.dec.setdefault(stack, {}).setdefault('skip_pool', {})[key] = pidx.bool()
        elif e[0] not in ('out', 'grad', 'grad_in'):
            raise ValueError(e[0])

    def decisions(self, outputs=None):
        """``outputs``: the FBCRNN forward tuple (y_fwd, y_bwd, ...) - adds the selector of max(y_fwd, y_bwd)."""
        dec = dict(self.dec)
        if outputs is not None and len(outputs) == 6 and outputs[1] is not None:
            dec['max_sel'] = (outputs[0] >= outputs[1]).detach()
        return dec


@contextlib.contextmanager
def tap_decisions(model):
    from pb_sed_amd import engine
    tap = _Tap(model)
    engine.DECISION_TAP = tap
    try:
        yield tap
    finally:
        engine.DECISION_TAP = None
