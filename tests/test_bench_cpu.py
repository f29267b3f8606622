"""CPU tests of the measurement harness: the per-precision roofline formula of bench.py, its synthetic
inputs, and the kernel-dispatch description the bench line quotes (mm_matmul_describe, no device work)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from micromix_amd import _lib  # noqa: E402


def test_roofline_time_matches_survey_reference_points():
    # 4096^3 = 137.4 GFLOP -> t* = 27.3 us at the fp8 rate, 13.7 us at the fp4 / fp6 rate
    assert bench.roofline_time_s(4096, 4096, (0, 0, 4096)) * 1e6 == pytest.approx(27.3, abs=0.05)
    assert bench.roofline_time_s(4096, 4096, (4096, 0, 0)) * 1e6 == pytest.approx(13.65, abs=0.05)
    assert bench.roofline_time_s(4096, 4096, (0, 4096, 0)) * 1e6 == pytest.approx(13.65, abs=0.05)
    # mixed: segments add; fp6 activations run at the fp4 rate on CDNA4
    t = bench.roofline_time_s(4096, 4096, (2048, 128, 1920)) * 1e6
    assert t == pytest.approx((2048 + 128) / 4096 * 13.65 + 1920 / 4096 * 27.3, abs=0.05)
    assert bench.roofline_time_s(4096, 4096, (12288, 1024, 1024)) * 1e6 == pytest.approx(51.2, abs=0.1)


def test_synthetic_inputs_are_reproducible_and_shaped():
    import torch
    x, w, idx = bench.synth_inputs(seed=0, m=64, n=32, k=256)
    x2, w2, idx2 = bench.synth_inputs(seed=0, m=64, n=32, k=256)
    assert x.shape == (64, 256) and w.shape == (32, 256) and idx.dtype == torch.int16
    assert torch.equal(x, x2) and torch.equal(w, w2) and torch.equal(idx, idx2)
    assert sorted(idx.tolist()) == list(range(256))          # a permutation: ascending mean |x| (reorder_indices.py:64-69)
    mean = x.float().abs().mean(0)
    assert set(idx[-2:].tolist()) == set(torch.topk(mean, 2).indices.tolist())   # the x20 outlier channels land in the fp8 segment


def test_matmul_describe_names_the_dispatch():
    lib = _lib.load()
    d = lambda *a: lib.mm_matmul_describe(*a).decode()
    assert "g256" in d(4096, 4096, 0, 0, 4096, 1, 0, 0) and "256 workgroups" in d(4096, 4096, 0, 0, 4096, 1, 0, 0)
    assert "last 8 tile columns" in d(4096, 14336, 0, 0, 4096, 1, 0, 0)        # 896 tiles = 3.5 rounds: tail balancing
    assert "mx_gemm_stream_kernel" in d(16, 4096, 0, 0, 4096, 1, 0, 0) and "mx_gemm_stream_kernel" in d(1, 14336, 2048, 128, 1920, 1, 0, 0)
    assert "skinny" in d(1, 4096, 0, 0, 4096, 1, 0, 0) and "mx_gemm_stream_kernel" in d(64, 4096, 0, 0, 4096, 1, 0, 0)   # all start-up / 32 < M <= 64, few features
    assert "g32n" in d(64, 14336, 0, 0, 4096, 1, 0, 0)                                                                    # ... many features: tiles
    assert "<false,false>" in d(4096, 4096, 0, 0, 4096, 0, 0, 0)               # matching-precision weights
    assert d(0, 4096, 0, 0, 4096, 1, 0, 0) == "none" and d(4096, 4096, 100, 0, 0, 1, 0, 0) == "none"
    assert "split-K" in d(192, 256, 12288, 1024, 1024, 1, _lib.MM_SPLIT_K_ALWAYS, 1 << 30)
    # few tiles: the 4-wave 64-row tiles while they fit one round of workgroups, then 128 x 128, 128 x 256
    assert "g16" in d(128, 4096, 0, 0, 4096, 1, 0, 0) and "256 workgroups (32x64 tiles)" in d(128, 4096, 0, 0, 4096, 1, 0, 0)
    assert "g32n" in d(256, 4096, 0, 0, 4096, 1, 0, 0) and "256 workgroups (64x64 tiles)" in d(256, 4096, 0, 0, 4096, 1, 0, 0)
    assert "g32::" in d(512, 4096, 0, 0, 4096, 1, 0, 0) and "(64x128 tiles)" in d(512, 4096, 0, 0, 4096, 1, 0, 0)
    assert "g64" in d(1024, 4096, 0, 0, 4096, 1, 0, 0) and "g128" in d(2048, 4096, 0, 0, 4096, 1, 0, 0)


def test_compact_line_fits_the_driver_record():
    """the ONE printed line stays under 6 KB and still carries `mixed`, `power`, `zero_operands` as numbers; the
    canned full result is round 4's own bench output (profiles/r04_summary.txt)."""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "bench_result_r04.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 12000                      # what the driver's tail could not hold
    line = bench.compact_line(full, "gpurun_out/bench_details.json")
    assert len(line) < bench.LINE_LIMIT == 6144 and "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == full["value"] and d["config"]["workload"].startswith("BASELINE.json configs[1]")
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    assert d["roofline"]["zero_operands"]["kernel_us"] == full["roofline"]["zero_operands"]["kernel_us"]
    assert d["power"]["package_w"] == full["power"]["package_w"]
    assert d["mixed"]["q_o_all_fp4"] == [full["mixed"]["q_o_all_fp4"]["kernel_us"], full["mixed"]["q_o_all_fp4"]["frac"]]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])
    assert d["llama_layer"]["1"]["us"] == full["llama_layer"]["by_rows"]["1"]["us_per_layer_graph"]
    # an oversized result sheds extras, never contract keys
    fat = dict(full, small_m={f"case{i}": dict(full["small_m"]["down_M16"]) for i in range(400)})
    line = bench.compact_line(fat, None)
    assert len(line) < bench.LINE_LIMIT and {"metric", "value", "roofline", "cpu_baseline", "mixed"} <= set(json.loads(line))


def test_gpus_n_without_rendezvous_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no WORLD_SIZE) starts torch.distributed.run as a CHILD and returns its exit
    code.  On this CPU-only box the ranks stop at "needs an MI355X": what is checked is the hop -- two ranks were started with a
    rendezvous on 127.0.0.1 and their failure is the parent's exit code (not a SystemExit about WORLD_SIZE)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MICROMIX_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    import torch
    if torch.cuda.is_available():
        assert p.returncode == 0, p.stderr[-1500:]
        return
    assert p.returncode != 0
    assert "launching the ranks" in p.stderr and "--nproc-per-node=2" in p.stderr and "--standalone --local-addr 127.0.0.1" in p.stderr
    assert "needs an MI355X" in p.stderr and "launch with torch.distributed.run" not in p.stderr


def test_describe_names_the_product_kernels_and_ignores_retired_switches():
    """round 6 pruned the round-5 experiments (one wave per SIMD, persistent tiles) from the product library: their environment switches
    no longer select anything -- the describe strings name the shipped kernels whatever MICROMIX_GEMM_W1 / MICROMIX_GEMM_PERSIST say"""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r)\nfrom micromix_amd import _lib\nlib = _lib.load()\n"
            "print(lib.mm_matmul_describe(4096, 4096, 2048, 128, 1920, 1, 0, 0).decode())\n"
            "print(lib.mm_matmul_describe(4096, 14336, 2048, 128, 1920, 1, 0, 0).decode())\n"
            "print(lib.mm_gate_up_activate_describe(4096, 14336).decode())\n") % ROOT
    def run(**env):
        e = {k: v for k, v in os.environ.items() if not k.startswith("MICROMIX_GEMM")}
        e.update(env)
        p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-800:]
        return p.stdout.splitlines()
    d = run()
    assert "mm::g256::mx_gemm256_kernel<true,false> x 256" in d[0] and "g256::" in d[1] and "mx_gemm256_act_kernel x 1792" in d[2]
    assert run(MICROMIX_GEMM_W1="1", MICROMIX_GEMM_PERSIST="1") == d
