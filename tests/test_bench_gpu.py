"""bench.py's multi-rank branch as a dry run on ONE GPU: two ranks over the `gloo` backend sharing the device
(MICROMIX_BENCH_BACKEND=gloo).  Not a measurement -- it checks that the script the driver launches for N > 1 runs to its JSON line:
rendezvous, K-shard layer, settle phase with the ranks in step, timed region, GEMM / all-reduce split, row-parallel figure, MLP
pairing.  (With >= 2 GPUs tests/test_tp_rccl_gpu.py covers the RCCL path itself.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_dry_run():
    """the driver's own form of the command: `python bench.py --gpus 2 ...` with no rendezvous in the environment -- bench.py starts its
    ranks itself (torch.distributed.run as a child process) and relays rank 0's line"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MICROMIX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    details = os.path.join(ROOT, "gpurun_out", "bench_details_tp2.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--details", details]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6144, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["parallelism"].startswith("tp2")
    assert d["tp"]["gemm_us"] > 0 and d["tp"]["allreduce_us"] > 0 and d["tp"]["allreduce_payload_bytes"] == 4096 * 4096 * 2
    assert d["row_parallel_no_exchange"]["global_rows"] == 8192 and d["tp_mlp"]["mlp_us"] > 0
    with open(details) as f:
        full = json.load(f)
    assert {"gemm_us_max_over_ranks", "allreduce_us_max_over_ranks", "allreduce_payload_bytes", "rank0_roofline"} <= set(full["tp"])


def test_bench_line_is_short_and_complete():
    """one rank, the driver's command with fewer steps: ONE line under 6 KB with the numbers the judge reads"""
    details = os.path.join(ROOT, "gpurun_out", "bench_details_test.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--details", details],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6144
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["roofline"]["frac"] > 0.2 and d["cpu_baseline"]["value"] > 0
    assert set(d["mixed"]) >= {"q_o_all_fp4", "q_o_3072_896_128", "down_12288_1024_1024"} and d["roofline"]["zero_operands"]["kernel_us"] > 0
    assert "power" in d and "llama_layer" in d and "quantizers" in d
