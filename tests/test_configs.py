"""CPU: `CfgNode.load_with_base` on this repository's configs/ yields trees identical to the reference's (G17: the reference's own
loader run on its own seven config files, tests/golden/make_golden.py::g17), tuple-vs-list types included."""
import json
import os

import pytest

from conftest import ROOT, GOLDEN
from ucod_dpl_amd.engine.config import CfgNode

TREES = json.load(open(os.path.join(GOLDEN, "g17_config_trees.json")))


def plain(x):
    if isinstance(x, dict):
        return {k: plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return {"__tuple__" if isinstance(x, tuple) else "__list__": [plain(v) for v in x]}
    return x


@pytest.mark.parametrize("rel", sorted(TREES))
def test_config_tree_equals_the_reference(rel):
    mine = plain(dict(CfgNode(CfgNode.load_with_base(os.path.join(ROOT, "configs", rel)))))
    assert mine == TREES[rel]


def test_runner_type_detection_follows_the_reference():
    """engine/runner/runner.py:631-650: `window_size` in model_cfg or a `refiner_path` selects the local-refinement runner."""
    from ucod_dpl_amd.engine.runner.runner import RunnerFactory
    first = CfgNode(CfgNode.load_with_base(os.path.join(ROOT, "configs", "uscod", "UCOD-DPL_dinov2.py")))
    second = CfgNode(CfgNode.load_with_base(os.path.join(ROOT, "configs", "uscod", "CORAL_dinov2.py")))
    assert RunnerFactory._detect_runner_type(first) == "standard"
    assert RunnerFactory._detect_runner_type(second) == "local_refine"
    first.train_cfg.refiner_path = "x.safetensors"
    assert RunnerFactory._detect_runner_type(first) == "local_refine"
    assert RunnerFactory.get_available_runners() == ["standard", "local_refine", "lr"]
    with pytest.raises(ValueError):
        RunnerFactory.create_runner(first, "nope")
