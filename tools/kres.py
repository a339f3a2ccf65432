"""Kernel resource usage of srl_kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage): name, SGPR, VGPR, scratch, occupancy, LDS."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "sr_livo_amd", "csrc", "srl_kernels.hip")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-fPIC",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(d, "k.o")]
    out = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
for b in re.split(r"remark: Function Name: ", out)[1:]:
    name = b.split()[0]
    if flt and flt not in name:
        continue
    def g(pat):
        return re.search(pat, b).group(1)
    vals = [g(r"TotalSGPRs: (\d+)"), g(r" VGPRs: (\d+)"), g(r"ScratchSize \[bytes/lane\]: (\d+)"), g(r"Occupancy \[waves/SIMD\]: (\d+)"),
            g(r"LDS Size \[bytes/block\]: (\d+)")]
    print("%-88s sgpr %4s vgpr %4s scratch %4s occ %s lds %s" % ((name[:88],) + tuple(vals)))
