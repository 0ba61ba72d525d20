"""bench.py's parts (round 6: the 1 140-line bench.py of round 5 split by job).

  launch   - `python bench.py --gpus N` outside a distributed launch spawns its N ranks itself
  line     - the ONE stdout line the driver parses: compact (< 4 KB, numbers only); everything else -> bench_extras.json + stderr
  cpu      - the CPU baseline leg (oracle = the reference's arithmetic, bit-identical; R processes x 1 thread)  [imports oracle/]
  points   - informational GPU points beside the headline (literal scalar-CSR SpMV, the box's identity)
  pmc      - HBM traffic of the run's own kernels: two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE)
  octree   - BASELINE configs[1]: the 1 M-dof graded octree mesh on the operators
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")
METRIC = "PCG iterations/sec + SpMV achieved HBM GB/s, 10M-DOF 3D elastostatic CSR"
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s HBM3E spec
F64_PEAK_TFLOPS = 78.6      # MI355X_MICROARCH.md: f64 vector = f64 matrix peak


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)
