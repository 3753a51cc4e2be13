"""COD measures (row N4): the HIP kernels behind the `statistics` mirror against the reference's own numbers (golden G16: the real
class on 18 seeded cases) and against the float64 oracle on larger and degenerate inputs."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conftest import load_golden  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"
KEYS = ("mae", "acc", "iou", "sm", "wfm", "adp_em", "adp_fm")
TOL = 1e-9                                                       # float64 on both sides; only the order of the sums differs


def record(pred, gt):
    from ucod_dpl_amd import ops
    r = ops.cod_metrics(torch.as_tensor(pred, dtype=torch.float32)[None].to(DEV), torch.as_tensor(gt, dtype=torch.float32)[None].to(DEV))[0].cpu().numpy()
    d = {k: r[i] for i, k in enumerate(KEYS)}
    d.update(em_curve=r[8:264], fm_curve=r[264:520], p_curve=r[520:776], r_curve=r[776:1032])
    return d


def test_every_measure_matches_the_reference_class():
    g = load_golden("g16_cod_metrics")
    for i in range(int(g["n"])):
        got = record(g[f"pred{i}"], g[f"gt{i}"])
        for k in KEYS + ("em_curve", "fm_curve", "p_curve", "r_curve"):
            d = float(np.max(np.abs(np.asarray(got[k]) - g[f"{k}{i}"].numpy())))
            assert d < TOL, (i, k, d)


def test_statistics_mirror_reproduces_get_result():
    from ucod_dpl_amd.engine.utils.metrics import statistics
    g = load_golden("g16_cod_metrics")
    st = statistics()
    for i in range(int(g["n"])):
        st.step(g[f"gt{i}"][None, None].to(DEV), g[f"pred{i}"][None, None].to(DEV))
    res = st.get_result()
    assert sorted(res) == sorted(k[6:] for k in g if k.startswith("final."))
    for k, v in res.items():
        assert abs(v - float(g["final." + k])) < TOL, (k, v, float(g["final." + k]))
    st.reset()
    assert st._records == []
    with pytest.raises(RuntimeError):
        st.step(g["gt0"][None, None], g["pred0"][None, None])       # host tensors: no CPU path


@pytest.mark.parametrize("h,w,kind", [(97, 131, "soft"), (240, 180, "binary"), (64, 64, "sparse"), (50, 75, "one_pixel"), (33, 40, "column"),
                                      (30, 30, "stripes")])
def test_against_the_float64_oracle(h, w, kind):
    """Sizes and shapes the golden set does not hold: a larger soft map, a binary mask, a few scattered foreground pixels (long
    nearest-pixel walks with many ties), a single foreground pixel (sample std of one element: the S-measure collapses to 0 as in the
    reference), foreground only in the last column (an empty S-measure quadrant), and regular stripes (every distance tied)."""
    from oracle import cod_metrics as OM
    g = torch.Generator().manual_seed(h * w)
    pred = torch.rand(h, w, generator=g)
    if kind == "soft":
        gt = (torch.rand(h, w, generator=g) > 0.6).float()
        gt[10:60, 20:90] = 1
    elif kind == "binary":
        gt = torch.zeros(h, w); gt[40:200, 30:150] = 1
        pred = (gt + (torch.rand(h, w, generator=g) > 0.9).float()).clamp(0, 1)
    elif kind == "sparse":
        gt = (torch.rand(h, w, generator=g) > 0.995).float()
        gt[5, 7] = 1
    elif kind == "one_pixel":
        gt = torch.zeros(h, w); gt[20, 30] = 1
    elif kind == "column":
        gt = torch.zeros(h, w); gt[5:25, w - 1] = 1
    else:
        gt = torch.zeros(h, w); gt[::5] = 1; gt[:, ::7] = 1
    ref = OM.image_measures(pred.numpy(), gt.numpy())
    got = record(pred, gt)
    for k in KEYS + ("em_curve", "fm_curve", "p_curve", "r_curve"):
        a, b = np.asarray(got[k], np.float64), np.asarray(ref[k], np.float64)
        assert np.array_equal(np.isnan(a), np.isnan(b)), (kind, k)
        d = float(np.nanmax(np.abs(a - b))) if a.size else 0.0
        assert d < TOL, (kind, k, d)


def test_batch_equals_one_by_one_and_is_repeatable():
    from ucod_dpl_amd import ops
    g = torch.Generator().manual_seed(5)
    pred = torch.rand(6, 48, 56, generator=g).to(DEV)
    gt = (torch.rand(6, 48, 56, generator=g) > 0.7).float().to(DEV)
    gt[3] = 0
    both = ops.cod_metrics(pred, gt)
    assert torch.equal(both, ops.cod_metrics(pred, gt))
    for i in range(6):
        assert torch.equal(both[i], ops.cod_metrics(pred[i:i + 1].contiguous(), gt[i:i + 1].contiguous())[0])
    assert float(both[3, 4]) == 0.0                                   # empty ground truth: weighted F is defined as 0
