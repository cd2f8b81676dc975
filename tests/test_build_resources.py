"""Compile-time guards on the gfx950 kernels (hipcc cross-compiles here): the MFMA conv kernels must
not spill to scratch memory (a silent 20 % slowdown when an array was demoted to private memory)
and must leave room for two 4-wave blocks per CU (<= 256 VGPRs)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_conv_kernels_have_no_scratch(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "sbb_textline_detection_amd", "csrc", "kernels.hip")
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o",
                          str(tmp_path / "k.o"), "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    blocks = re.split(r"remark: [^\n]*Function Name: ", res.stderr)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if not any(k in name for k in ("conv_igemm_mfma", "stem_conv_pairs", "dec_tail_fused", "conv3x3_c64_direct", "bottleneck_fused")):
            continue
        seen += 1
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        vgprs = int(re.search(r" VGPRs: (\d+)", b).group(1))
        spill = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
        assert scratch == 0 and spill == 0, f"{name}: scratch {scratch} B/lane, {spill} VGPR spills"
        assert vgprs <= 256, f"{name}: {vgprs} VGPRs"
    assert seen >= 6
