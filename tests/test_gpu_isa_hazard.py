"""The hardware behaviour that tools/check_isa.py guards against, shown in
isolation on the GPU: a 64-bit shift whose 32-bit amount sits in the last VGPR
of the wave's allocation (tools/ubench/shift_hazard.hip: kernels with nothing
but that instruction, eight work-groups per CU)."""
import os
import re
import subprocess
import warnings

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "ubench", "shift_hazard")


def test_shift_amount_in_the_last_allocated_vgpr():
    if not os.path.exists(EXE):
        pytest.fail("tools/ubench/shift_hazard is not built (python huffman_amd/build.py)")
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    wrong = {}
    for line in r.stdout.splitlines():
        m = re.match(r"(.*?)\s+\(numRegs \d+\) amount\s+(\d+): wrong lanes (\d+),", line)
        if m:
            wrong[(m.group(1).strip(), int(m.group(2)))] = int(m.group(3))
    assert len(wrong) >= 20, r.stdout
    # the controls: the same instruction with the amount one register lower, or with
    # one more register allocated, and instructions that are not 64-bit shifts
    for name in ("amount in v63, 72 VGPRs allocated", "amount in v62, 64 VGPRs allocated",
                 "v_mad_u64_u32, src0 in v63 of 64", "v_mad_u64_u32, src1 in v63 of 64",
                 "v_cvt_f64_u32, source in v63 of 64", "v_lshlrev_b32 (32-bit), amount in v63"):
        for amt in (5, 2):
            assert wrong[(name, amt)] == 0, (name, amt, r.stdout)
    assert wrong[("v_lshl_add_u64, amount in v63 of 64", 2)] == 0
    # the pattern itself: wrong in every work-group that is not the first on its CU
    hazard = [wrong[(name, amt)] for name in ("amount in v63, 64 VGPRs allocated",
                                              "amount in v71, 72 VGPRs allocated",
                                              "v_lshrrev_b64, amount in v63 of 64",
                                              "v_ashrrev_i64, amount in v63 of 64") for amt in (5, 2)]
    if not all(hazard):
        warnings.warn("the last-VGPR shift amount was read correctly here: the scan in "
                      "tools/check_isa.py may be stricter than this GPU needs\n" + r.stdout)
