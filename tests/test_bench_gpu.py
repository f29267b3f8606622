"""bench.py's multi-rank branch as a dry run on ONE GPU: two ranks over the `gloo` backend sharing the device
(MICROMIX_BENCH_BACKEND=gloo).  Not a measurement -- it checks that the script the driver launches for N > 1 runs to its JSON line:
rendezvous, K-shard layer, settle phase with the ranks in step, timed region, GEMM / all-reduce split, row-parallel figure, MLP
pairing.  (With >= 2 GPUs tests/test_tp_rccl_gpu.py covers the RCCL path itself.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_dry_run():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MICROMIX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["parallelism"].startswith("tp2")
    assert {"gemm_us_max_over_ranks", "allreduce_us_max_over_ranks", "allreduce_payload_bytes", "rank0_roofline"} <= set(d["tp"])
    assert d["row_parallel_no_exchange"]["global_rows"] == 8192 and d["tp_mlp"]["mlp_us"] > 0
