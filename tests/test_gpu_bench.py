"""GPU (-m gpu): bench.py in the two forms the driver launches it -- directly (N = 1) and under `python -m
torch.distributed.run` (what it does for N > 1: one rank per device, RCCL for the barrier and the max-over-ranks of the
timing) -- at a size that takes seconds.  The launcher form runs here with ONE rank (a test box has one device): torch and
its RCCL imported before the library, the process group on `nccl`, the barriers around the timed region, rank 0's line.
Both lines must carry the contract's keys, `roofline` and the parity sample."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--tiles", "1024", "--no-extras", "--no-cpu-baseline"]


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def _check(d):
    assert d["metric"] and d["unit"] and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


def test_bench_launched_directly(built):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    _check(_line(res.stdout))


def test_bench_under_the_launcher(built):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py")] + SMALL,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _line(res.stdout)
    _check(d)
