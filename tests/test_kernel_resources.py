"""Build-time guard for the hot kernel (CPU-only: hipcc cross-compiles gfx950): no scratch spills and at most
128 VGPRs (4 waves/SIMD) for the association kernel's default instance.  A silent spill cost 16 MB of extra
HBM writes per launch once (profiles/README.md)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_association_kernel_has_no_spills(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "sr_livo_amd", "csrc", "srl_kernels.hip")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-fPIC",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", str(tmp_path / "k.o")]
    out = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    blocks = re.split(r"remark: Function Name: ", out)
    seen = 0
    armed = 0
    for b in blocks[1:]:
        name = b.split()[0]
        vg = int(re.search(r"VGPRs: (\d+)", b).group(1))
        sc = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        if "srl_assoc_kernel" in name or "srl_assoc_armed_kernel" in name:
            seen += 1
            armed += "srl_assoc_armed_kernel" in name
            assert sc == 0, (name, sc)
            assert vg <= 128, (name, vg)
    assert seen >= 3
    assert armed >= 14              # the armed launches (r = 1 and r = 2, seven workgroup sizes): the same budget, no scratch
