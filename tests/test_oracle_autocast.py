"""The oracle's autocast emulation (oracle/vit.py::autocast_rounding: the roundings `accelerate launch --mixed_precision fp16` puts into the
reference's forward, scripts/launch_train_first_stage.sh:20) against torch's own autocast on the real HF Dinov2Model: the emulated forward must
sit much closer to the really-autocast forward than either sits to f32.  bench.py reports the emulated deviation at full size as the second
data point beside the f32 reference ("fp16-autocast reference vs bf16 build")."""
import pytest
import torch

from oracle import vit as OV

transformers = pytest.importorskip("transformers")


def rel_l2(a, b):
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_autocast_emulation_tracks_torch_autocast_on_hf_dinov2(dtype):
    cfg = transformers.Dinov2Config(hidden_size=64, num_hidden_layers=3, num_attention_heads=2, mlp_ratio=4, image_size=70, patch_size=14,
                                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layerscale_value=0.5,
                                    attn_implementation="eager")      # modeling_dinov2.py:153-179, the path SURVEY.md maps
    torch.manual_seed(0)
    model = transformers.Dinov2Model(cfg).eval()
    with torch.no_grad():
        for prm in model.parameters():                       # spread the weights so that the 16-bit roundings are visible above f32 noise
            prm.mul_(3.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.randn(2, 3, 70, 70)
    with torch.no_grad():
        ref32 = model(x).last_hidden_state
        try:
            with torch.autocast("cpu", dtype=dtype):
                real = model(x).last_hidden_state.float()
        except RuntimeError as e:                             # a CPU kernel this torch build lacks in that type
            pytest.skip(f"CPU autocast in {dtype} is not runnable here: {e}")
        emu32, _ = OV.dinov2_forward(x, sd, heads=2, patch=14, eps=cfg.layer_norm_eps)
        emu, _ = OV.dinov2_forward(x, sd, heads=2, patch=14, eps=cfg.layer_norm_eps, autocast=dtype)
    assert rel_l2(emu32, ref32) < 1e-5                        # the f32 restatement itself (G8 pins it too)
    d_real, d_emu, d_between = rel_l2(real, ref32), rel_l2(emu, ref32), rel_l2(emu, real)
    assert d_real > 1e-4 and d_emu > 1e-4                     # the roundings are visible
    assert 0.5 < d_emu / d_real < 2.0, (d_emu, d_real)        # same size of deviation from f32 ...
    assert d_between < 0.3 * d_real, (d_between, d_real)      # ... and mostly the SAME deviation (measured 0.13 bf16, 0.17 fp16)
