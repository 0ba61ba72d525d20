"""`python bench.py --gpus N` outside any distributed launch spawns the N ranks itself."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

from . import ROOT, BENCH_PY, METRIC, HBM_PEAK_GBS, F64_PEAK_TFLOPS, log

import socket


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    def run(extra_env):
        env = dict(os.environ)
        env.update(extra_env)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH_PY] + sys.argv[1:]
        log("launching", args.gpus, "ranks:", " ".join(cmd[1:8]), "...")
        limit = float(os.environ.get("PCG_BENCH_RANKS_TIMEOUT_S", "900"))     # a hung collective must not eat the caller's whole budget
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, err = p.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            log(f"ranks did not finish within {limit:.0f} s - terminating the launch (process group {p.pid})")
            import signal
            os.killpg(p.pid, signal.SIGTERM)
            try:
                out, err = p.communicate(timeout=30)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                out, err = p.communicate()
            sys.stderr.write(err or "")
            return subprocess.CompletedProcess(cmd, 124, out, (err or "") + f"\n[bench] ranks did not finish within {limit:.0f} s")
        sys.stderr.write(err or "")
        return subprocess.CompletedProcess(cmd, p.returncode, out, err)

    def why(err):
        """The lines of the ranks' stderr that say what failed (RCCL / HIP / engine errors), for the JSON line."""
        keys = ("nccl", "rccl", "pcg_", "hip", "error", "Error", "did not finish")
        hit = [l.strip() for l in (err or "").splitlines() if any(k in l for k in keys) and "Traceback" not in l]
        return " | ".join(hit[-6:])[-900:]
    r = run({})
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    by_watchdog = bool(line) and '"extras": "the optional objects' in line[-1]       # rank 0 printed its headline and ended the job (main: extras_guard)
    if (r.returncode != 0 or not line) and not by_watchdog and args.comm == "native" and os.environ.get("PCG_BENCH_NO_RETRY") != "1":
        # keep the scaling point measurable if the native communicator cannot come up on this node: same kernels, same
        # RCCL, but the collectives are issued through torch.distributed callbacks; the line says which transport ran AND
        # carries what the native run reported (comm.native_error)
        reason = why(r.stderr) or f"exit code {r.returncode}, no diagnostic on stderr"
        log(f"native-communicator run failed (rc {r.returncode}): {reason}; retrying with --comm torch")
        r = run({"PCG_BENCH_COMM": "torch", "PCG_BENCH_NATIVE_FAILED": "1", "PCG_BENCH_NATIVE_ERROR": reason})
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:                                      # nothing ran: the driver still gets a line that says why
        line = [json.dumps({"metric": METRIC, "value": None, "n_gpus": args.gpus,
                            "error": why(r.stderr) or f"exit code {r.returncode}", "unit": "iterations/s"})]
    if line:
        print(line[-1], flush=True)
    return 0 if by_watchdog else (r.returncode if r.returncode != 0 else (0 if line else 1))
