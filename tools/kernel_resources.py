#!/usr/bin/env python
"""Registers / scratch / LDS of the kernels of one HIP source (development): compiles it for gfx950 device-only to assembly and reads
the amdhsa.kernels metadata.   usage: python tools/kernel_resources.py [source=hip_backend.hip] [name filter regex]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else "hip_backend.hip"
flt = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
csrc = os.path.join(ROOT, "pcg-mpi-solver_amd", "csrc")
out = "/tmp/kernel_resources.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                       "-I" + csrc, "--cuda-device-only", "-S", os.path.join(csrc, src), "-o", out], stderr=subprocess.DEVNULL)
s = open(out).read()
md = s[s.index("amdhsa.kernels:"):]
for e in re.split(r"\n  - ", md):
    n = re.search(r"\.name:\s+(\S+)", e)
    if not n:
        continue
    dem = subprocess.run(["c++filt", n.group(1)], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"^void pcg::", "", dem).split("(")[0]
    if not flt.search(dem):
        continue
    g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", e) or [None, "?"])[1]
    print(f"{dem[:78]:80s} vgpr {g('vgpr_count'):>4} agpr {g('agpr_count'):>3} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}")
