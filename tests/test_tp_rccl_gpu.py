"""The tensor-parallel layouts over the real RCCL backend, one process per GPU.  Needs at least two GPUs on
the node: the single-GPU boxes of the pool skip it, the first multi-GPU lease runs it.  The ranks are started as CHILD processes
(torch.distributed.run) before any of them touches a GPU; this process only counts the devices."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus():
    import torch
    return torch.cuda.device_count()        # does not initialise the GPU on this image


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, script, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", script)]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_tp_layouts_over_rccl(world):
    if _gpus() < world:
        pytest.skip(f"needs {world} GPUs on the node, found {_gpus()}")
    p = _run(world, "tp_rccl_worker.py")
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    for r in range(world):
        assert f"RCCL-OK rank {r}" in p.stdout


def test_rccl_worker_runs_with_one_rank():
    """the same worker with a world of one: RCCL initialises, every layout runs through its collective calls and matches the
    unsharded product exactly -- keeps the script green on the single-GPU boxes, where the multi-rank cases above skip"""
    if _gpus() < 1:
        pytest.skip("needs a GPU")
    p = _run(1, "tp_rccl_worker.py")
    assert p.returncode == 0 and "RCCL-OK rank 0" in p.stdout, p.stdout[-3000:] + "\n" + p.stderr[-3000:]


def test_two_devices_driven_by_one_process():
    """the launcher state (dynamic-LDS attributes, CU counts) is cached per device id: one process, two GPUs, results equal
    (twin of tests/test_matmul_gpu.py::test_two_devices_in_one_process, as a child process so that it can run anywhere in the
    session)"""
    if _gpus() < 2:
        pytest.skip("needs 2 GPUs on the node")
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from micromix_amd import mixedgemm\n"
        "g = torch.Generator().manual_seed(0)\n"
        "x = torch.randn((300, 1024), generator=g).to(torch.bfloat16); w = (torch.randn((512, 1024), generator=g) * 0.02).to(torch.bfloat16)\n"
        "idx = torch.randperm(1024, generator=g).to(torch.int16); split = (512, 128, 384); outs = []\n"
        "for d in (0, 1, 0, 1):\n"
        "    dev = torch.device('cuda', d)\n"
        "    a = mixedgemm.reorder_quantize_x(x.to(dev), idx.to(dev), *split); b = mixedgemm.reorder_quantize_w4(w.to(dev), idx.to(dev), *split)\n"
        "    outs.append(mixedgemm.matmul(a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3], a[4], b[4], a[5], b[5]).cpu())\n"
        "assert all(torch.equal(o, outs[0]) for o in outs); print('TWO-DEVICES-OK')\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0 and "TWO-DEVICES-OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
