"""The pair kernels (csrc/rollout_kernel.h: rollout_kernel2) carry hand-written DPP instructions (csrc/wave.h: WaveH::fma_pick /
fnma_pick / rcp_pick: v_fmac_f32_dpp, v_rcp_f32_dpp through inline asm) that the compiler's hazard recogniser cannot see.  This test
compiles both instantiations to ISA with the product flags (hipcc cross-compiles without a GPU) and checks the listing: no DPP
instruction may read a VGPR that one of the two preceding VALU issue slots wrote, and the kernels must not have grown scratch."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("wpb,occ,queue", [(1, 2, "false"), (4, 2, "true")])
def test_hand_written_dpp_instructions_respect_the_valu_write_hazard(wpb, occ, queue):
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not installed")
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "isa", "probe.sh"), "DimsGo2", str(wpb), str(occ), queue, "-DPROBE_PAIR"],
                         capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    listing = os.path.join(ROOT, "build", "isa", f"DimsGo2_{wpb}_{occ}_{queue}.s")
    chk = subprocess.run(["python", os.path.join(ROOT, "tools", "isa", "check_dpp_hazards.py"), listing, "rollout_kernel2"],
                         capture_output=True, text=True)
    assert chk.returncode == 0, chk.stdout
    m = re.search(r"(\d+) DPP instructions checked", chk.stdout)
    assert m and int(m.group(1)) > 300, chk.stdout          # the fused forms are there (product flags: -DDIAL_FUSED_DPP)
    text = open(listing).read()
    assert text.count("v_fmac_f32_dpp") > 300 and "v_permlane16_swap_b32" in text
