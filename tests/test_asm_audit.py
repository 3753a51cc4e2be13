"""CPU: the inline-asm LDS reads of the product kernels keep their destination registers untouched until the covering s_waitcnt
(tools/asm_window_audit.py; ADVICE round 2: a hipcc upgrade or a register-pressure change could break this silently)."""
import os
import shutil
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import asm_window_audit as A  # noqa: E402

CSRC = os.path.join(ROOT, "ucod_dpl_amd", "csrc")


def test_auditor_sees_a_violation_and_accepts_the_correct_order():
    good = """
_Zkern:
	;;#ASMSTART
	ds_read_b64_tr_b16 v[64:65], v129
	;;#ASMEND
	;;#ASMSTART
	ds_read_b64_tr_b16 v[66:67], v129 offset:1024
	;;#ASMEND
	v_add_f32 v1, v2, v3
	;;#ASMSTART
	s_waitcnt lgkmcnt(1)
	;;#ASMEND
	v_mov_b32 v5, v64
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_mov_b32 v6, v67
"""
    v, k, r = A.audit(good)
    assert (len(v), k, r) == (0, 1, 2)
    bad = good.replace("v_add_f32 v1, v2, v3", "v_mov_b32 v9, v65")                  # a copy inside the window of the first read
    assert len(A.audit(bad)[0]) == 1
    late = good.replace("v_mov_b32 v5, v64", "v_mov_b32 v5, v66")                    # lgkmcnt(1) leaves the SECOND read in flight
    assert len(A.audit(late)[0]) == 1


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
@pytest.mark.parametrize("src,flags", [("attention_bwd.hip", []), ("gemm_split.hip", []), ("gemm_split.hip", ["-DUCOD_HALF_F16"])])
def test_product_kernels_keep_asm_read_windows_clean(src, flags):
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "a.s")
        subprocess.run([A.HIPCC] + A.BASE + flags + ["-I", CSRC, "-o", out, os.path.join(CSRC, src)], check=True, stderr=subprocess.DEVNULL)
        v, kernels, reads = A.audit(open(out).read())
    assert kernels >= 2 and reads >= 100, (kernels, reads)
    assert v == [], v[:5]
