"""CPU: the feature-cache on-disk format against a cache written by the reference's own
MultiCacheManager / MetaListPickleIO (tests/golden/cache_ref, produced by tests/golden/make_golden.py g13)."""
import json
import os
import pickle

import pytest
import torch

from conftest import load_golden, GOLDEN
from ucod_dpl_amd.data.datasets import MultiCacheManager, build_feature_cache

REF = os.path.join(GOLDEN, "cache_ref")


def test_reads_a_cache_written_by_the_reference():
    g = load_golden("g13_cache_items")
    m = MultiCacheManager(REF, "dinov2", "train", "COD10K")
    fc = m.get_features_cache()
    assert fc.mode == "r" and fc.length() == 3
    for i in range(3):
        assert torch.equal(fc.read_file(i), g[f"f{i}"])
    pc = m.get_pseudo_label_cache()
    assert pc.length() == 3 and torch.equal(pc.read_file(1), (g["f1"][:1] > 0).float())
    assert MultiCacheManager(REF, "dinov2", "val", "COD10K").get_pseudo_label_cache() is None     # cache_manager.py:88-92


def test_writes_what_the_reference_writes(tmp_path):
    g = load_golden("g13_cache_items")
    m = MultiCacheManager(str(tmp_path), "dinov2", "train", "COD10K")
    fc = m.get_features_cache()
    assert fc.mode == "w"
    fc.dump_list([g[f"f{i}"] for i in range(3)])
    assert fc.mode == "r"
    mine = tmp_path / "features_cache" / "dinov2" / "train" / "COD10K"
    ref = os.path.join(REF, "features_cache", "dinov2", "train", "COD10K")
    assert open(mine / "index.json").read() == open(os.path.join(ref, "index.json")).read()
    for i in range(3):
        a = pickle.load(open(mine / f"data_{i}.pkl", "rb"))
        b = pickle.load(open(os.path.join(ref, f"data_{i}.pkl"), "rb"))
        assert a.dtype == b.dtype and a.device == b.device and torch.equal(a, b)


def test_integrity_check_switches_to_write_mode(tmp_path):
    m = MultiCacheManager(str(tmp_path), "dinov2", "val", "NC4K")
    fc = m.get_features_cache()
    fc.dump_list([torch.zeros(2, 2)])
    os.remove(tmp_path / "features_cache" / "dinov2" / "val" / "NC4K" / "data_0.pkl")
    assert MultiCacheManager(str(tmp_path), "dinov2", "val", "NC4K").get_features_cache().mode == "w"   # pickleio.py:93-103


def test_build_feature_cache_batches_and_streams(tmp_path):
    """The batched pass: a stub extractor (the HIP backbone needs a GPU; tests/test_gpu_* cover it) with ragged image sizes."""
    calls = []

    def fe(x):
        calls.append(tuple(x.shape))
        return None, x[:, :, ::2, ::2] * 2.0

    imgs = [torch.full((3, 4, 4), float(i)) for i in range(5)] + [torch.full((3, 6, 6), 9.0)]
    fc = MultiCacheManager(str(tmp_path), "dinov2", "train", "X").get_features_cache()
    n = build_feature_cache(imgs, fe, fc, batch_size=2, device="cpu")
    assert n == 6 and fc.mode == "r" and fc.length() == 6
    assert calls == [(2, 3, 4, 4), (2, 3, 4, 4), (1, 3, 4, 4), (1, 3, 6, 6)]
    for i in range(5):
        t = fc.read_file(i)
        assert t.shape == (3, 2, 2) and float(t.mean()) == 2.0 * i
    assert fc.read_file(5).shape == (3, 3, 3)
    idx = json.load(open(tmp_path / "features_cache" / "dinov2" / "train" / "X" / "index.json"))
    assert idx == {str(i): f"data_{i}.pkl" for i in range(6)}
    with pytest.raises(RuntimeError):
        build_feature_cache(imgs, fe, fc, batch_size=2, device="cpu")


def test_pseudo_label_post_process_and_cache(tmp_path):
    """Row N3 host side (runs without a GPU: the connected components come from the library's host entry): the package's
    refine_post_process vs the reference's (G14), and the pseudo_label_cache directory it feeds."""
    from ucod_dpl_amd.generate_pseudo_label import refine_post_process
    g = load_golden("g14_pseudo_label")
    outs = []
    for key, a in (("pp_out", 4), ("pp_out_a9", 9)):
        for mk, ref in zip(g["pp_in"], g[key]):
            out = refine_post_process(mk.clone(), area_threshold=a)
            assert torch.equal(out, ref)
            outs.append(out)
    pc = MultiCacheManager(str(tmp_path), "dinov2", "train", "TR-CAMO+TR-COD10K").get_pseudo_label_cache()
    pc.dump_list(outs[:5])
    assert pc.mode == "r" and pc.length() == 5 and torch.equal(pc.read_file(3), outs[3])
    assert (tmp_path / "pseudo_label_cache" / "TR-CAMO+TR-COD10K" / "index.json").exists()
