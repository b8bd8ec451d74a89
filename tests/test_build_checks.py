"""Checks on the compiled code that need hipcc but no GPU."""
import os, shutil, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_inline_asm_dpp_instructions_keep_their_wait_states():
    # the DPP butterflies of the spectral kernels are inline assembly, invisible to the compiler's hazard recogniser: every one
    # of them must sit two wait states behind any VALU write of the register it reads (tools/check_dpp_hazard.py)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazard.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "DPP instructions, 0 without" in r.stdout


def test_hazard_checker_sees_a_violation():
    # the checker itself: a VALU write directly in front of a DPP read of the same register is reported, one behind an s_nop 1 is not
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import check_dpp_hazard as c
    finally:
        sys.path.pop(0)
    bad_asm = "\tv_mul_f32_e32 v3, v1, v2\n\tv_fmac_f32_dpp v3, v3, v4 row_ror:8 row_mask:0xf bank_mask:0xf\n"
    ok_asm = "\tv_mul_f32_e32 v3, v1, v2\n\ts_nop 1\n\tv_fmac_f32_dpp v3, v3, v4 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp v5, v5, v4 row_ror:8 row_mask:0xf bank_mask:0xf\n"
    one_apart = "\tv_mul_f32_e32 v3, v1, v2\n\tv_add_f32_e32 v9, v1, v2\n\tv_fmac_f32_dpp v3, v3, v4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
    assert c.check(bad_asm) == (1, [("v_mul_f32_e32 v3, v1, v2", "v_fmac_f32_dpp v3, v3, v4 row_ror:8 row_mask:0xf bank_mask:0xf")])
    assert c.check(ok_asm)[1] == []
    assert len(c.check(one_apart)[1]) == 1
