"""GPU: CORAL SparseRefiner (rows R1-R4) through the drop-in module against the reference's own eval forward (G9)."""
import math
import os
import sys

import pytest
import torch

from conftest import load_golden, maxdiff, within

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
    within("refiner:window_preds", rel_l2(wp, wref), 3.5e-3)          # measured 1.5-1.6e-3 (bf16 projections)
    assert maxdiff(opt["h_preds"].cpu(), g[tag + ".h_preds"]) < 0.05 * wref.abs().max().item()
    assert maxdiff(opt["GE_w"].cpu(), g[tag + ".GE_w"]) < 1e-4
    within("refiner:outputs:" + tag, rel_l2(out, g[tag + ".outputs"]), 5e-4)          # measured 2.6e-5 / 2.0e-4


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


# ------------------------------------------------------------------------------------------------ CORAL validation loop (row N4)
def _coral_loop(req_m):
    import types
    from ucod_dpl_amd.engine.runner import LocalRefineValidationLoop
    from ucod_dpl_amd.models.uscod import baseline
    g = load_golden("g15_coral_loop")
    dev = torch.device("cuda", 0)
    model = baseline(CfgNode(dict(dim=768, feature_size=68, ema_weight=0.99, dis_use_features=False))).to(dev).eval()
    model.load_state_dict({k[4:]: v.to(dev) for k, v in g.items() if k.startswith("dec.")}, strict=True)
    torch.manual_seed(RI.SEED)
    refiner = RI.perturb_(SparseRefiner.from_config(CfgNode(dict(window_size=3, threshold=0.0015)))).eval().to(dev)
    logs = []
    runner = types.SimpleNamespace(device=dev, model=model, refiner=refiner, world_size=1, rank=0, val_dataloader=[],
                                   logger=types.SimpleNamespace(log_table=logs.append, log=logs.append))
    cfg = CfgNode(dict(train_cfg=dict(dist_train=False), model_cfg=dict(window_length=6), dataset_cfg=dict(valset_cfg=dict(require_m_patches=req_m, DATASET="X"))))
    return LocalRefineValidationLoop(cfg, runner), g, runner, logs


@pytest.mark.parametrize("req_m", [False, True])
def test_coral_validation_loop_matches_reference(req_m):
    """The reference's LocalRefineValidationLoop methods run on the real baseline / SparseRefiner (G15) vs the HIP-backed mirror."""
    loop, g, runner, _ = _coral_loop(req_m)
    t = f"m{int(req_m)}."
    l, m, h = RI.coral_inputs()
    fd = loop._prepare_validation_features(l, m, h)
    assert maxdiff(fd["l_features"].cpu(), g[t + "l_features"]) < 1e-5 and maxdiff(fd["h_features"].cpu(), g[t + "h_features"]) < 1e-5
    assert maxdiff(fd["preds"].cpu(), g[t + "preds"]) < 2e-4                       # exact-f32 decoder
    assert int(loop._should_crop_center(fd["preds"])) == int(g[t + "crop"])
    with torch.no_grad():
        out, _, _ = runner.refiner(fd["l_features"], fd["h_features"], fd["preds"])
    within("refiner:e2e_outputs", rel_l2(out, g[t + "outputs"]), 6e-4)                        # bf16 projections inside the refiner
    assert torch.equal(loop._center_pad(g[t + "outputs"].cuda()).cpu(), g[t + "padded"])
    for src, ref in ((t + "outputs", t + "up"), (t + "padded", t + "up_pad")):
        up = loop.process_preds(g[src].cuda(), (50, 70)).cpu()
        assert float((up != g[ref]).float().mean()) < 2e-3                         # thresholded bilinear: last-bit ties only
    assert torch.equal(loop.process_preds(g["probs_in"].cuda(), (20, 31)).cpu(), g["probs_up"])


def test_coral_loop_run_and_window_features():
    """run(): batches in the reference's dict layout -> MAE; WindowFeatures: one resize, nine windows by slicing, one batched pass."""
    import types
    import numpy as np
    from ucod_dpl_amd.engine.runner import WindowFeatures, loop_look_twice as LT
    from ucod_dpl_amd.data.utils.feature_extractor import backbone, random_state_dict, ARCHS
    loop, g, runner, logs = _coral_loop(False)
    l, m, h = RI.coral_inputs()
    label = (torch.rand(1, 1, 50, 70, generator=torch.Generator().manual_seed(2)) > 0.5).float()
    batch = dict(pseudo_label=None, label_tensor=label, features=l, img_path=["x"], m_inputs=m, h_inputs=h, index=[0])
    runner.val_dataloader = [batch, batch]
    res = loop.run()
    up = loop._process_validation_batch(batch, LT.MAEStatistics())
    assert abs(res["MAE"] - float((up.cpu() - label[0]).abs().mean())) < 1e-6 and logs
    # window features on a tiny backbone: windows are exact slices of ONE Pillow-bilinear resize of the whole image
    ARCHS["wf_vit"] = (128, 2, 2, 14, 56, True)
    dev = torch.device("cuda", 0)
    bb = backbone.from_state_dict(random_state_dict("wf_vit", seed=1, image_size=56), heads=2, device=dev)
    lt_runner = types.SimpleNamespace(device=dev, model=None, world_size=1, rank=0, val_dataloader=[], logger=None)
    lt_cfg = CfgNode(dict(train_cfg=dict(dist_train=False), model_cfg=dict(feature_size=68), val_cfg=dict(look_twice=True, look_twice_th=0.15, expand_type="dynamic"),
                          dataset_cfg=dict(valset_cfg=dict(image_size=(56, 56)))))
    lt = LT.ValLoop_Look_Twice(lt_cfg, lt_runner, feature_extractor=bb)
    wf = WindowFeatures(bb, lt, window_size=3, grid=(56, 56), extractor_size=(84, 84), image_size=(56, 56))
    rng = np.random.default_rng(0)
    img = rng.integers(0, 255, (90, 120, 3), dtype=np.uint8)
    lfe, hin, _ = wf.get_features(img, require_m_patches=False, crop_center=True)
    assert hin.shape == (1, 9, 128, 4, 4) and lfe.shape == (1, 128, 4, 4)
    from oracle import look_twice as OLT
    crop = wf.crop_center(img)
    big = OLT.crop_resize_normalize(crop, [0, 0, crop.shape[1], crop.shape[0]], (168, 168))
    from ucod_dpl_amd.vit_engine import SplitViTEngine
    assert isinstance(wf.fe.engine, SplitViTEngine)                                # (round 6: the reference computes these features in plain fp32, lr_dataset.py:97-157)
    _, ref_key = wf.fe(big[:, 56:112, 112:168].unsqueeze(0).cuda())               # window (row 1, col 2) = index 5
    assert torch.equal(hin[0, 5], ref_key[0]) or float((hin[0, 5] - ref_key[0]).norm() / ref_key[0].norm()) < 1e-6
    wf16 = WindowFeatures(bb, lt, window_size=3, grid=(56, 56), extractor_size=(84, 84), image_size=(56, 56), precision=None)
    _, h16, _ = wf16.get_features(img, require_m_patches=False, crop_center=True)
    assert torch.equal(h16[0, 5], bb(big[:, 56:112, 112:168].unsqueeze(0).cuda())[1][0])      # the extractor as given: the 16-bit engine, bit for bit
    assert float((h16 - hin).norm() / hin.norm()) < 5e-3


def test_local_refine_runner_from_the_coral_config(tmp_path):
    """configs/uscod/CORAL_dinov2.py -> create_runner -> LocalRefineRunner (engine/runner/runner.py:400-590, :631-650): the shipped
    first-stage checkpoint loads strictly into the frozen baseline, a refiner checkpoint written in the reference's layout loads
    strictly, and launch_val runs the validation loop to the nine COD measures; the result equals driving the loop by hand."""
    import os
    from safetensors.torch import save_file
    from conftest import ROOT
    from ucod_dpl_amd.engine.runner import create_runner, LocalRefineRunner, LocalRefineValidationLoop
    cfg = CfgNode(CfgNode.load_with_base(os.path.join(ROOT, "configs", "uscod", "CORAL_dinov2.py")))
    cfg.log_cfg.log_path = str(tmp_path)
    cfg.train_cfg.checkpoint = os.path.join(ROOT, "tests", "golden", "weights", "UCOD_DPL_dinov2.safetensors")
    cfg.model_cfg.window_length = 6                              # RI.coral_inputs() geometry (the config's 56 is the full-size value)
    torch.manual_seed(RI.SEED)
    ref_refiner = RI.perturb_(SparseRefiner.from_config(CfgNode(dict(window_size=3, threshold=0.0015)))).eval()
    rpath = tmp_path / "refiner.safetensors"
    save_file({k: v.detach().contiguous() for k, v in ref_refiner.state_dict().items()}, str(rpath))
    cfg.train_cfg.refiner_path = str(rpath)
    l, m, h = RI.coral_inputs()
    label = (torch.rand(1, 1, 50, 70, generator=torch.Generator().manual_seed(2)) > 0.5).float()
    batch = dict(pseudo_label=None, label_tensor=label, features=l, img_path=["x"], m_inputs=m, h_inputs=h, index=[0])
    runner = create_runner(cfg, val_dataloader=[batch, batch])
    assert isinstance(runner, LocalRefineRunner) and not any(p.requires_grad for p in runner.model.parameters())
    from safetensors.torch import load_file
    shipped = load_file(cfg.train_cfg.checkpoint)
    for k, v in runner.model.state_dict().items():
        assert torch.equal(v.cpu(), shipped[k]), k
    for k, v in runner.refiner.state_dict().items():
        assert torch.equal(v.cpu(), ref_refiner.state_dict()[k]), k
    res = runner.launch_val()
    assert sorted(res) == sorted(["Sm", "wFm", "meanFm", "adpFm", "maxFm", "meanEm", "adpEm", "maxEm", "MAE"]) or "MAE" in res
    by_hand = LocalRefineValidationLoop(cfg, runner).run()
    assert res == by_hand
    runner.save_checkpoint(3)
    saved = load_file(os.path.join(str(tmp_path), "refiner_ckp", "epoch3.pth", "model.safetensors"))
    assert sorted(saved) == sorted(ref_refiner.state_dict())
    with pytest.raises(NotImplementedError):
        runner.launch_train()


@pytest.mark.parametrize("tag", ["full", "partial"])
@pytest.mark.parametrize("kind", ["prob", "logit"])
def test_sparse_refiner_training_mode_matches_reference(tag, kind):
    """models/UDLR.py:52-86 with the module in .train() and h_targets given -- G9b = the REAL reference module run that way: the forward is
    the eval forward (every dropout is 0), cal_ex_loss is the IoU-weighted window loss.  The loss is checked twice: end to end against the
    reference's value (its window_preds are f32, ours come through bf16 projections -> 2e-2 relative), and -- the kernel alone -- on the
    reference's own window_preds against the reference's value (f32 arithmetic: 2e-6 relative)."""
    g, g9 = load_golden("g9b_refiner_train"), load_golden("g9_refiner")
    k = f"{tag}.{kind}."
    torch.manual_seed(RI.SEED)
    m = RI.perturb_(SparseRefiner.from_config(CfgNode(dict(window_size=3, threshold=0.0015)))).train().cuda()
    l, h, preds = RI.make_inputs(tag == "partial")
    ht = RI.make_h_targets(kind)
    with torch.no_grad():
        out, ex, opt = m(l.cuda(), h.cuda(), preds.cuda(), ht.cuda())
    ref = float(g[k + "ex_loss"])
    assert torch.equal(opt["window_targets"].cpu(), g[k + "window_targets"])
    within("refiner:train_ex_loss_rel", abs(float(ex) - ref) / ref, 5e-4)          # measured 1.2e-5 .. 1.4e-4
    within("refiner:train_outputs", rel_l2(out, g[k + "outputs"]), 5e-4)
    within("refiner:train_outputs_vs_eval", rel_l2(out, g9[tag + ".outputs"]), 5e-4)
    # the loss kernel on the reference's own window logits
    mask = g9[tag + ".mask"].bool()
    win_flat = torch.nonzero(mask.flatten()).flatten().to(torch.int32).cuda()
    wp = g[k + "window_preds"].cuda().contiguous()
    n, hh, ww = wp.shape[0], wp.shape[-2], wp.shape[-1]
    l_up = ops.bilinear_resize(preds.cuda(), hh * 3, ww * 3)
    htd = ht.cuda().contiguous()
    part, ious, loss = (torch.empty(n, device="cuda"), torch.empty(n, device="cuda"), torch.empty(1, device="cuda"))
    N.check(N.load().ucod_window_loss(N.ptr(wp), N.ptr(htd), N.ptr(win_flat), N.ptr(l_up), int(kind == "logit"), N.ptr(part), N.ptr(ious), N.ptr(loss),
                                      n, 2, hh, ww, 3, N.stream()), "ucod_window_loss")
    assert abs(float(loss[0]) - ref) < 2e-6 * max(1.0, abs(ref)), (float(loss[0]), ref)
    o_loss, o_t, o_iou = OR.cal_ex_loss(preds, g[k + "window_preds"], mask, ht, 3)
    assert maxdiff(ious.cpu() * 1.5, (o_iou).cpu().clamp(0, 1)) < 1e-6 or maxdiff((ious.cpu() * 1.5).clamp(0, 1), o_iou) < 1e-6


def test_sparse_refiner_training_mode_edge_cases():
    """No window selected -> ex_loss is the reference's python 0; eval mode ignores h_targets; training mode without h_targets raises."""
    torch.manual_seed(RI.SEED)
    m = RI.perturb_(SparseRefiner.from_config(CfgNode(dict(window_size=3, threshold=0.0015)))).train().cuda()
    l, h, preds = RI.make_inputs(False)
    confident = torch.full_like(preds, 14.0)                      # zero entropy everywhere: no window passes the threshold
    out, ex, opt = m(l.cuda(), h.cuda(), confident.cuda(), RI.make_h_targets("prob").cuda())
    assert ex == 0 and int(opt["mask"].sum()) == 0 and "window_targets" not in opt
    with pytest.raises(ValueError):
        m(l.cuda(), h.cuda(), preds.cuda())
    m.eval()
    out, ex, opt = m(l.cuda(), h.cuda(), preds.cuda(), RI.make_h_targets("prob").cuda())
    assert ex == 0
