"""Tiny deterministic stand-ins for feature extractor / CNN / RNN used to pin the reference's
inference heads (tests/golden/gen_golden.py) and to replay them against the oracle / product classes."""
import torch


class StubFeatures(torch.nn.Module):
    def forward(self, x, seq_len=None, targets=None):
        return (x, seq_len) if targets is None else (x, seq_len, targets)


class StubCNN(torch.nn.Module):
    conditional_dims = 0

    def forward(self, x, seq_len=None, condition=None):
        return x, seq_len


class StubRNN(torch.nn.Module):
    """y[b,k,t] = sum_c A[k,c] * running-mean of h over the causal (or anti-causal) context."""

    def __init__(self, a, reverse):
        super().__init__()
        self.a = torch.as_tensor(a)
        self.reverse = reverse

    def forward(self, h, seq_len=None):
        t = h.shape[-1]
        n = torch.arange(1, t + 1, dtype=h.dtype)
        if self.reverse:
            ctx = h.flip(-1).cumsum(-1).div(n).flip(-1)
        else:
            ctx = h.cumsum(-1).div(n)
        return torch.einsum('kc,bct->bkt', self.a, ctx), seq_len
