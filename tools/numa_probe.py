"""experiment: does pinning the process to the GPU-local NUMA node change the per-iteration time?"""
import os, subprocess, sys, json
sys.path.insert(0, ".")
import torch
p = torch.cuda.get_device_properties(0)
print({k: getattr(p, k) for k in dir(p) if k.startswith("pci")})
dom, bus, dev = getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id
path = f"/sys/bus/pci/devices/{dom:04x}:{bus:02x}:{dev:02x}.0"
node = open(path + "/numa_node").read().strip(); cpus = open(path + "/local_cpulist").read().strip()
print(path, "numa", node, "cpus", cpus)
def run(prefix):
    out = subprocess.run(prefix + "python bench.py --steps 40 --no-cpu-baseline --no-configs --no-aux-legs 2>/dev/null | tail -1", shell=True, capture_output=True, text=True).stdout
    d = json.loads(out); return round(d["ms_per_esikf_iter"] * 1e3, 2), round(d["value"])
other = "64-127,192-255" if cpus.startswith("0") else "0-63,128-191"
for rep in range(2):
    print("unpinned", run(""), "local", run(f"taskset -c {cpus} "), "remote", run(f"taskset -c {other} "), flush=True)
