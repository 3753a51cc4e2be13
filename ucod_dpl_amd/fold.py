"""Host side of the LayerNorm fold (include/ucod_dpl.h: ucod_gemm_lnfold): pure torch, runs on any device.

nn.LayerNorm(gamma, beta, eps) followed by nn.Linear(W, b) (transformers modeling_dinov2.py:348-381: norm1 -> query / key / value, norm2 -> fc1) is ONE
matrix product on the un-normalised rows plus two per-row scalars:

    LN(x) W^T + b  =  rstd[m] * ( x W'^T  -  mean[m] * c[n] )  +  b'[n]
    W' = half( q (.) W (.) gamma ),    c[n] = sum_k W'[n][k]  (of the ROUNDED W'),    b' = q (.) (W beta + b)

``q`` is an optional per-output scale applied before the rounding (the softmax pre-scale of the Q rows).  The column sums are taken over the rounded
weights because that is what the MFMA multiplies the rows by: `x W'^T - mean * c` then cancels exactly, whatever the rounding did to W'.
"""
import torch


def fold_layernorm_linear(gamma, beta, w, b, row_scale=None, half=torch.float16):
    """-> (W' in ``half`` [N, K], b' f32 [N], c f32 [N]).  gamma, beta [K]; w [N, K]; b [N]; row_scale [N] or None.  Products and sums in f64, one
    round-to-nearest-even to ``half`` (what the library's cast kernel does)."""
    q = torch.ones(w.shape[0], dtype=torch.float64, device=w.device) if row_scale is None else row_scale.double()
    wf = (w.double() * gamma.double()[None, :] * q[:, None]).float().to(half).contiguous()
    colsum = wf.double().sum(1).float().contiguous()
    bias = ((w.double() @ beta.double() + b.double()) * q).float().contiguous()
    return wf, bias, colsum


def row_stats(x, eps):
    """(rstd, -mean * rstd) per row of x [M, K], biased variance like nn.LayerNorm: what the folded epilogue applies (f64 here: the checker's form)."""
    xd = x.double()
    mean = xd.mean(1)
    rstd = (xd.var(1, unbiased=False) + eps).rsqrt()
    return torch.stack((rstd, -mean * rstd), 1)


def apply_folded(x, wf, bias, colsum, eps, stats=None):
    """The folded product in f64 on the given (already rounded) operands: the exact value the device epilogue approximates."""
    st = row_stats(x, eps) if stats is None else stats.double()
    acc = x.double() @ wf.double().t()
    return st[:, :1] * acc + st[:, 1:] * colsum.double()[None, :] + bias.double()[None, :]
