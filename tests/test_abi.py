"""CPU: the C-ABI library builds, loads, and exports exactly the symbols include/ucod_dpl.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from ucod_dpl_amd import native
    if not os.path.exists(native.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return native.load()


def header_functions():
    text = open(os.path.join(ROOT, "include", "ucod_dpl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ucod_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from ucod_dpl_amd import native
    names = header_functions()
    assert len(names) >= 20
    assert sorted(native.SIGNATURES.keys()) == names
    raw = ctypes.CDLL(native.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), n


def test_abi_version_and_argument_validation(lib):
    assert lib.ucod_abi_version() == 5
    # rejected before any device work: null pointers / bad sizes return UCOD_EINVAL (-1)
    assert lib.ucod_gemm_bf16(0, None, None, None, 1, 1, 64, None, None, None, None, 0, 0, None) == -1
    assert lib.ucod_layernorm(None, None, None, None, 1, 100, 1e-6, 0, None) == -1
    assert lib.ucod_bilinear_resize(None, None, 1, 1, 1, 1, 1, None) == -1
    assert lib.ucod_adamw_ema(None, None, None, None, None, 4, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, 0.0, None) == -1
    # the split-operand pass (ABI 5): two or three terms only, K a multiple of 8, null pointers refused
    assert lib.ucod_split_products(2) == 3 and lib.ucod_split_products(3) == 6 and lib.ucod_split_products(4) == 0
    assert lib.ucod_split_rows(None, 8, None, 1, 8, 2, 0, 0, 1.0, None) == -1
    assert lib.ucod_layernorm_split(None, None, None, None, 1, 128, 1e-6, 2, 0, None) == -1
    assert lib.ucod_attention_split_fwd(None, None, 1, 1, 1, 2, None) == -1
    assert lib.ucod_attention_split_operand_bytes(1, 33, 2, 3) == 2 * (2 * 64 * 3 * 64 * 2) + 3 * 2 * 64 * 64 * 2
    assert lib.ucod_attention_split_operand_bytes(1, 33, 2, 4) == 0
    assert lib.ucod_clock_probe(None, None) == -1


def test_workspace_size_helpers(lib):
    from ucod_dpl_amd import native
    d = native.VitDesc()
    d.B, d.C, d.H, d.W, d.P, d.D, d.heads, d.F, d.L, d.Kpad = 32, 3, 518, 518, 14, 768, 12, 3072, 12, 640
    d.eps = 1e-6
    need = lib.ucod_vit_workspace_bytes(ctypes.byref(d))
    M = 32 * 1370
    assert need >= M * 768 * 4 + M * 768 * 2 * 2 + M * 2304 * 2 + M * 3072 * 2
    d.heads = 11                                           # head_dim != 64 -> rejected
    assert lib.ucod_vit_workspace_bytes(ctypes.byref(d)) == 0
    d.heads = 12
    assert lib.ucod_vit_split_workspace_bytes(ctypes.byref(d), 2) >= M * 768 * 4 + 2 * M * 3 * 768 * 2 + M * 2304 * 4 + M * 3 * 3072 * 2
    assert lib.ucod_vit_split_workspace_bytes(ctypes.byref(d), 3) > lib.ucod_vit_split_workspace_bytes(ctypes.byref(d), 2)
    assert lib.ucod_vit_split_workspace_bytes(ctypes.byref(d), 4) == 0
    d.full_last_layer = 1                                  # the split pass is key-minimal only
    assert lib.ucod_vit_split_workspace_bytes(ctypes.byref(d), 2) == 0
    assert lib.ucod_disc_saved_bytes(32, 68) == (32 * (32 * 68 * 68 + 16 * 34 * 34 + 8 * 17 * 17) + 112) * 4 + 112 * 8
    assert lib.ucod_dba_bwd_workspace_bytes(2, 100) == 2 * 128 * 100 * 4


def test_product_path_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under ucod_dpl_amd/ may reference it."""
    pkg = os.path.join(ROOT, "ucod_dpl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)


def test_graft_entry_build_assertions_hold():
    """__graft_entry__.build() = make + these assertions; the driver runs it every round (a stale ABI number there fails the round's build check)."""
    import __graft_entry__ as G
    G.check_build()


def test_product_libraries_read_only_the_documented_environment_variables():
    """measurement knobs are compiled out of the product libraries (csrc/common.h: lab_env; `make knobs` builds them in for tools/).  The only
    UCOD_* names left in the binaries are the three GEMM summation-order switches include/ucod_dpl.h documents; the Python side reads no UCOD_LN_FOLD / UCOD_FOLD_* knob."""
    import subprocess
    from ucod_dpl_amd import native
    documented = {"UCOD_GEMM_NO_PATCH", "UCOD_GEMM_PATCH_ROUNDS", "UCOD_GEMM_NO_MIXED"}
    header = open(os.path.join(ROOT, "include", "ucod_dpl.h")).read()
    for path in (native.LIB_PATH, native.LIB_PATH_F16):
        names = {l.strip() for l in subprocess.run(["strings", path], capture_output=True, text=True, check=True).stdout.splitlines() if re.fullmatch(r"UCOD_[A-Z0-9_]+", l.strip())}
        assert names == documented, (path, names)
    assert all(n in header for n in documented)
    src = open(os.path.join(ROOT, "ucod_dpl_amd", "vit_engine.py")).read()
    assert "UCOD_LN_FOLD" not in src and "UCOD_FOLD_ABL" not in src
