"""CPU: host pieces of Look-Twice (C++ CCL, Pillow resampler restatement, integer box logic of the product loop)
against Pillow itself, the oracle and the reference's own integer tables (G7)."""
import types

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import load_golden
from oracle import look_twice as OLT
from oracle.resize import pil_resize_u8
from ucod_dpl_amd.engine.runner import loop_look_twice as LT


def fake_loop(th=0.15, expand="dynamic", img=(64, 64)):
    loop = LT.ValLoop_Look_Twice.__new__(LT.ValLoop_Look_Twice)
    loop.img_size = img
    loop.cfg = types.SimpleNamespace(val_cfg=types.SimpleNamespace(look_twice_th=th, expand_type=expand))
    return loop


def test_oracle_resampler_is_pillow():
    rng = np.random.default_rng(0)
    for (h, w, oh, ow) in ((37, 37, 200, 123), (300, 451, 518, 518), (1000, 700, 518, 518), (518, 518, 518, 518), (5, 9, 40, 3)):
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        for filt, pf in (("bilinear", Image.BILINEAR), ("bicubic", Image.BICUBIC)):
            ref = np.asarray(Image.fromarray(a, "L").resize((ow, oh), pf))
            assert np.array_equal(pil_resize_u8(a, ow, oh, filt), ref), (h, w, oh, ow, filt)
    rgb = rng.integers(0, 256, (211, 333, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(rgb, "RGB").resize((518, 518), Image.BILINEAR))
    assert np.array_equal(pil_resize_u8(rgb, 518, 518, "bilinear"), ref)


def test_native_host_resampler_is_pillow():
    rng = np.random.default_rng(1)
    for (h, w, oh, ow) in ((37, 37, 200, 123), (37, 37, 37, 90), (37, 37, 11, 37), (64, 80, 7, 5), (37, 37, 37, 37)):
        a = (rng.random((h, w)) > 0.5).astype(np.uint8) * 255
        ref = np.asarray(Image.fromarray(a, "L").resize((ow, oh)))           # Pillow default = BICUBIC for 'L'
        assert np.array_equal(LT.pil_resize_u8(a, ow, oh, bicubic=True), ref)
        ref = np.asarray(Image.fromarray(a, "L").resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(LT.pil_resize_u8(a, ow, oh, bicubic=False), ref)


def test_native_ccl_matches_oracle_labelling():
    rng = np.random.default_rng(2)
    for density in (0.05, 0.3, 0.5, 0.7):
        m = (rng.random((61, 47)) < density).astype(np.uint8)
        n1, l1 = LT.connected_components(m * 255)
        n2, l2 = OLT.connected_components(m)
        assert n1 == n2 and np.array_equal(l1, l2)
    n, lab = LT.connected_components(np.zeros((8, 8), np.uint8))
    assert n == 1 and lab.max() == 0
    diag = np.eye(9, dtype=np.uint8)
    assert LT.connected_components(diag)[0] == 2            # 8-connectivity joins the diagonal


def test_product_box_logic_reproduces_reference_tables():
    g = load_golden("g7_look_twice_int")
    masks = g["masks"].numpy()
    for et, col in (("dynamic", g["dynamic"]), ("const", g["const"])):
        loop = fake_loop(expand=et)
        for m, ref in zip(masks, col):
            try:
                bx = loop.boxes_from_mask(m * 255)
                got = "none" if bx is None else ";".join(",".join(str(v) for v in b) for b in bx)
            except ValueError:
                got = "ValueError"
            except ZeroDivisionError:
                got = "ZeroDivisionError"
            assert got == str(ref)
    for row in g["resize_bbox"].numpy():
        b, (ow, oh, nw, nh), exp = [int(v) for v in row[:4]], [int(v) for v in row[4:8]], [int(v) for v in row[8:]]
        assert LT.ValLoop_Look_Twice.resize_bbox(b, ow, oh, nw, nh) == exp


def test_mae_statistics_semantics():
    st = LT.MAEStatistics()
    gt = torch.zeros(1, 1, 4, 4); gt[..., :2, :] = 255
    pred = torch.zeros(1, 4, 4); pred[..., :1, :] = 1
    st.step(gt, pred > 0.5)
    assert abs(st.get_result()["MAE"] - 0.25) < 1e-12


def test_component_label_order_against_real_opencv_when_installed():
    """The goldens (G7 / G14 / G15) were generated with cv2.connectedComponents replaced by scipy.ndimage.label (raster order of each
    component's first pixel): OpenCV is not installed in the build container.  OpenCV's 8-connectivity labelling scans 2x2 blocks, so
    its label ORDER can differ for components that start in the same 2-row strip; label order feeds the stable-sort ties and the paste
    order of process_preds.  Where the real library is available this test pins the claim on it; elsewhere it is skipped and the
    limitation stands as documented (README.md)."""
    cv2 = pytest.importorskip("cv2")
    import numpy as np
    from oracle import look_twice as OLT
    rng = np.random.default_rng(3)
    for _ in range(50):
        m = (rng.random((48, 64)) > 0.72).astype(np.uint8) * 255
        n_ref, lab_ref = cv2.connectedComponents(m, connectivity=8)
        n, lab = OLT.connected_components(m)
        assert n == n_ref
        # same partition; the ORDER is what may differ
        pairs = {(int(a), int(b)) for a, b in zip(lab.ravel(), lab_ref.ravel())}
        assert len(pairs) == n
        if not np.array_equal(lab, lab_ref):
            pytest.xfail("OpenCV numbers components in a different order than raster-first-pixel on this mask: order-dependent steps "
                         "(tie order of equal-area boxes, paste order) follow the scipy convention of the goldens")
