"""GPU: CORAL SparseRefiner (rows R1-R4) through the drop-in module against the reference's own eval forward (G9)."""
import math
import os
import sys

import pytest
import torch

from conftest import load_golden, maxdiff

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import refiner_init as RI  # noqa: E402
from oracle import refiner as OR  # noqa: E402
from oracle import vit as OV  # noqa: E402
from ucod_dpl_amd import native as N, ops  # noqa: E402
from ucod_dpl_amd.engine.config import CfgNode  # noqa: E402
from ucod_dpl_amd.models.UDLR import SparseRefiner  # noqa: E402


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("B,Nq,Nk,heads", [(1, 36, 36, 8), (2, 200, 77, 2), (1, 300, 3136, 1), (2, 64, 64, 8)])
def test_cross_attention_head_dim_96(B, Nq, Nk, heads):
    g = torch.Generator().manual_seed(Nq + Nk)
    D = heads * 96
    c = (96 ** -0.5) * math.log2(math.e)
    q = (torch.randn(B * Nq, D, generator=g) * 1.5 * c).to(torch.bfloat16)
    kv = (torch.randn(B * Nk, 2 * D, generator=g) * 1.5).to(torch.bfloat16)
    qh = q.float().view(B, Nq, heads, 96).transpose(1, 2)
    kh = kv.float()[:, :D].reshape(B, Nk, heads, 96).transpose(1, 2)
    vh = kv.float()[:, D:].reshape(B, Nk, heads, 96).transpose(1, 2)
    p = torch.softmax(torch.matmul(qh, kh.transpose(2, 3)) * math.log(2.0), dim=-1)
    ref = torch.matmul(p, vh).transpose(1, 2).reshape(B * Nq, D)
    qd, kvd = q.cuda(), kv.cuda()
    out = torch.empty(B * Nq, D, dtype=torch.bfloat16, device="cuda")
    N.check(N.load().ucod_cross_attention96_fwd(N.ptr(qd), D, N.ptr(kvd), kvd.data_ptr() + D * 2, 2 * D, N.ptr(out), B, Nq, Nk, heads, N.stream()), "xattn")
    assert maxdiff(out.float().cpu(), ref) < 3e-2 and rel_l2(out.float(), ref) < 1e-2


@pytest.mark.parametrize("tag", ["full", "partial"])
def test_sparse_refiner_matches_reference(tag):
    g = load_golden("g9_refiner")
    torch.manual_seed(RI.SEED)
    m = RI.perturb_(SparseRefiner.from_config(CfgNode(dict(window_size=3, threshold=0.0015)))).eval().cuda()
    l, h, preds = RI.make_inputs(tag == "partial")
    with torch.no_grad():
        out, ex, opt = m(l.cuda(), h.cuda(), preds.cuda())
    assert ex == 0
    assert torch.equal(opt["mask"].float().cpu(), g[tag + ".mask"])
    assert torch.equal(opt["coords_list"].cpu(), g[tag + ".coords_list"])
    assert maxdiff(opt["entropy"].cpu(), g[tag + ".entropy"]) < 1e-5
    wp, wref = opt["window_preds"].cpu(), g[tag + ".window_preds"]
    assert wp.shape == wref.shape
    # bf16 projections + bf16 attention probabilities vs the f32 reference
    assert rel_l2(wp, wref) < 2e-2, rel_l2(wp, wref)
    assert maxdiff(opt["h_preds"].cpu(), g[tag + ".h_preds"]) < 0.05 * wref.abs().max().item()
    assert maxdiff(opt["GE_w"].cpu(), g[tag + ".GE_w"]) < 1e-4
    assert rel_l2(out, g[tag + ".outputs"]) < 2e-2


def test_refiner_small_kernels_exact_f32():
    """The f32 pieces (gather/transpose, dwconv+mask head, window scatter, gated ensembling) against the oracle at 1e-5."""
    torch.manual_seed(RI.SEED)
    m = RI.perturb_(SparseRefiner.from_config(CfgNode(dict(window_size=3, threshold=0.0015)))).eval()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    lib = N.load()
    # gather + transpose
    src = torch.randn(5, 768, 30, generator=g)
    idx = torch.tensor([3, 0, 4], dtype=torch.int32)
    out = torch.empty(3 * 30, 768, device="cuda")
    src_d, idx_d = src.cuda(), idx.cuda()                   # keep the device buffers alive across the asynchronous launch
    N.check(lib.ucod_gather_tokens(N.ptr(src_d), N.ptr(idx_d), N.ptr(out), 3, 768, 30, N.stream()), "gather")
    assert torch.equal(out.cpu(), src[idx.long()].permute(0, 2, 1).reshape(90, 768))
    # depthwise conv + mask head on token-major activations
    x = torch.randn(2, 9, 9, 768, generator=g)
    ref = torch.nn.functional.conv2d(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), sd["HRE.CSF.depthwise_conv.weight"], sd["HRE.CSF.depthwise_conv.bias"],
                                                                padding=3, groups=768), sd["HRE.CSF.mask_dec.weight"], sd["HRE.CSF.mask_dec.bias"])
    mc = m.cuda()
    P = mc._prepare(torch.device("cuda", 0))
    win = torch.empty(2, 1, 9, 9, device="cuda")
    x_d = x.cuda().contiguous()
    N.check(lib.ucod_dwconv7_maskdec(N.ptr(x_d), N.ptr(P["dwT"]), N.ptr(P["dwb"]), N.ptr(P["mw"]), P["mb"], N.ptr(win), 2, 9, 9, 768, N.stream()), "dw")
    assert maxdiff(win.cpu(), ref) < 2e-4
    # gated ensembling
    l1 = torch.randn(2, 1, 7, 7, generator=g) * 2
    l2 = torch.randn(2, 1, 21, 21, generator=g)
    ref_out, ref_w = OR.gated_ensembler(l1, l2, sd)
    l1u = ops.bilinear_resize(l1.cuda(), 21, 21)
    o, w = torch.empty(2, 1, 21, 21, device="cuda"), torch.empty(2, 1, 21, 21, device="cuda")
    wsb = torch.empty(lib.ucod_gated_ensemble_workspace_bytes(2, 21, 21), dtype=torch.uint8, device="cuda")
    l2_d = l2.cuda()
    N.check(lib.ucod_gated_ensemble(N.ptr(l1u), N.ptr(l2_d), N.ptr(P["f0w"]), N.ptr(P["f0b"]), N.ptr(P["f2w"]), P["f2b"], N.ptr(o), N.ptr(w), N.ptr(wsb), 2, 21, 21, N.stream()), "ge")
    assert maxdiff(w.cpu(), ref_w) < 2e-5 and maxdiff(o.cpu(), ref_out) < 2e-5


def test_sparse_refiner_no_window_selected():
    """Edge case of ASR.py:41-51: confident predictions everywhere -> no window passes the entropy threshold.  The reference then
    feeds an empty batch through CSF; h_preds is all zeros and the output is the gated ensemble of preds with zeros.  HIP mirror vs
    the CPU oracle (pinned by G9 on the non-empty cases)."""
    from oracle import refiner as OR
    torch.manual_seed(RI.SEED)
    m = RI.perturb_(SparseRefiner.from_config(CfgNode(dict(window_size=3, threshold=0.0015)))).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    l, h, preds = RI.make_inputs(False)
    preds = torch.full_like(preds, 15.0)
    preds[1] = -15.0
    ref_out, ref_opt = OR.sparse_refiner_forward(l, h, preds, sd)
    assert int(ref_opt["mask"].sum()) == 0
    m = m.cuda()
    with torch.no_grad():
        out, ex, opt = m(l.cuda(), h.cuda(), preds.cuda())
    assert int(opt["mask"].sum()) == 0 and opt["coords_list"].shape[0] == 0
    assert float(opt["h_preds"].abs().max()) == 0.0
    assert maxdiff(out.cpu(), ref_out) < 1e-4 * max(1.0, ref_out.abs().max().item())
