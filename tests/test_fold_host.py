"""CPU: the host side of the LayerNorm fold (ucod_dpl_amd/fold.py) against nn.LayerNorm -> nn.Linear (transformers modeling_dinov2.py:348-381).

The identity  LN(x) W^T + b = rstd (x W'^T - mean c) + b'  is exact in exact arithmetic for ANY W' as long as c is the column sum of THAT W' and the
weights it is compared with are gamma-folded from the same W'; what the fp16 rounding of W' costs is measured separately."""
import pytest
import torch

from ucod_dpl_amd.fold import fold_layernorm_linear, row_stats, apply_folded

EPS = 1e-6


def _case(M, N, K, seed, massive=0.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g) * (0.5 + 2 * torch.rand(M, 1, generator=g)) + 0.3 * torch.randn(M, 1, generator=g)
    if massive:
        x[:, 3] = massive
        x[::2, K - 2] = -0.75 * massive
    x = x.to(torch.float16)                                    # the residual stream as the device stores it
    gamma, beta = 1 + 0.3 * torch.randn(K, generator=g), 0.2 * torch.randn(K, generator=g)
    w, b = 0.04 * torch.randn(N, K, generator=g), 0.1 * torch.randn(N, generator=g)
    return x, gamma, beta, w, b


@pytest.mark.parametrize("massive", [0.0, 200.0, 3.0e4])
def test_fold_identity_on_the_rounded_weights(massive):
    x, gamma, beta, w, b = _case(64, 48, 256, 1, massive)
    q = torch.cat((torch.full((16,), 0.125 * 1.4426950408889634), torch.ones(32)))
    wf, bf_, c = fold_layernorm_linear(gamma, beta, w, b, row_scale=q)
    assert wf.dtype == torch.float16 and c.dtype == torch.float32
    got = apply_folded(x, wf, bf_, c, EPS)
    # LayerNorm -> Linear with the weight that W' actually is: un-fold gamma and q from the ROUNDED W' (gamma, q != 0)
    w_eff = wf.double() / (gamma.double()[None, :] * q.double()[:, None])
    ln = torch.nn.functional.layer_norm(x.double(), (256,), gamma.double(), beta.double(), EPS)
    ref = (ln @ w_eff.t() + (w.double() @ beta.double() + b.double()) - (w_eff @ beta.double())) * q.double()
    # (bias: b' uses the UNROUNDED W for the beta term, as the engine does; the reference above does the same)
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() < 1e-5 * max(1.0, scale)          # exact up to the f32 storage of b' and c


def test_fold_against_plain_layernorm_linear_costs_one_weight_rounding():
    x, gamma, beta, w, b = _case(128, 96, 768, 2)
    wf, bf_, c = fold_layernorm_linear(gamma, beta, w, b)
    got = apply_folded(x, wf, bf_, c, EPS)
    ref = torch.nn.functional.layer_norm(x.double(), (768,), gamma.double(), beta.double(), EPS) @ w.double().t() + b.double()
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 4e-4, rel                                      # 2^-11 / sqrt(3) per weight, random signs


def test_row_stats_are_layernorm_statistics():
    x = torch.randn(10, 256, generator=torch.Generator().manual_seed(3)).to(torch.float16)
    st = row_stats(x, EPS)
    xd = x.double()
    assert torch.allclose(st[:, 0], (xd.var(1, unbiased=False) + EPS).rsqrt()) and torch.allclose(st[:, 1], -xd.mean(1) * st[:, 0])
