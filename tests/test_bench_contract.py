"""bench.py's output contract (the driver parses this line)."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_cpu_baseline_leg():
    """The reported CPU baseline is the reference's comparator (torch SDPA on the host cores) on a bounded sample;
    the oracle port is timed next to it."""
    m = _bench_module()
    small = dict(m.CFG, seqlen=256)
    r = m.cpu_baseline(small, budget_s=0.6)
    assert r["kind"] == "reference" and r["unit"] == "TFLOP/s" and r["value"] > 0 and r["cores"] >= 1
    assert "scaled_dot_product_attention" in r["sample"]
    assert r["oracle_port_tflops"] > 0 and "oracle" in r["oracle_port_sample"]


def test_flop_convention_matches_baseline_md():
    m = _bench_module()
    # BASELINE.md / SURVEY 8(d): config 2 forward = 549.76 GFLOP, fwd+bwd = 3.5x
    assert abs(m.fwd_flops(m.CFG) / 1e9 - 549.76) < 0.01


@pytest.mark.gpu
def test_bench_json_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5",
                          "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["dtype"] == "bf16" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or (r["traffic"] > 1e8 and "NOT measured in this run" in r["traffic_source"])
    # the socket's own matrix ceiling, measured in the run (bare MFMA loop, random operands), next to the nominal peak
    pc = r["practical_ceiling"]
    assert pc is not None and 1000.0 < pc["tflops"] <= 2600.0 and pc["seconds"] > 0.3, pc
    assert abs(r["frac_of_ceiling"] - r["achieved"] / pc["tflops"]) < 1e-3
    assert r["frac_of_ceiling"] < 1.0 and r["step_frac_of_ceiling"] < 1.0
    assert r["sustained_step"]["steps"] >= 20 and r["sustained_step"]["tflops"] > 0
    assert abs(r["frac_median"] - r["achieved_median"] / r["peak"]) < 1e-3
    oc = d["other_configs"]
    assert {"config3", "config4_fp8_kv", "config4_fp16_kv", "config5_shard_1_of_8"} <= set(oc)
    assert oc["config4_fp8_kv"]["achieved_gbs"] > 0 and oc["config3"]["fwd_tflops"] > 0
    sc = d["strong_scaling_config5"]
    assert sc["scaling"] == "strong" and sc["n_gpus"] == 1 and sc["heads_per_gpu"] == 32
    # value is consistent with ms_per_step and the algorithmic FLOPs of config 2
    assert abs(d["value"] - 1924.16e9 / (d["ms_per_step"] * 1e-3) / 1e12) / d["value"] < 1e-2
    # the per-kernel medians (individually evented launches) compose the step they were taken next to
    t = d["kernels"]["timing"]
    assert abs(t["sum_of_kernels_ms"] - (d["kernels"]["fwd"]["ms"] + d["kernels"]["bwd_all"]["ms"])) < 1e-3
    assert 0.97 <= t["sum_over_step_evented"] <= 1.03, t        # same statistic on both sides: the kernels compose the step
    assert 0.97 <= t["sum_over_step"] <= 1.03, t                # vs the wall-clock mean of the timed steps: the region runs on a socket
    #                                                              in its sustained state (0.974 when it timed the clock ramp)
    # the same W + K region timed from an idle socket rides along; the sustained figure is not slower than the ramp
    cs = d["cold_start"]
    assert cs["ms_per_step"] > 0 and abs(cs["value"] - 1924.16e9 / (cs["ms_per_step"] * 1e-3) / 1e12) / cs["value"] < 1e-2
    assert d["ms_per_step"] <= 1.02 * cs["ms_per_step"], (d["ms_per_step"], cs)
    for name in ("fwd", "bwd_dq", "bwd_all"):
        assert d["kernels"][name]["ms_min"] <= d["kernels"][name]["ms"]


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_dry_run_of_the_multi_gpu_path():
    """N > 1 without a GPU: the driver's launch line (torch.distributed.run, one rank per GPU) with --dry-run - rendezvous on
    127.0.0.1, barrier on both sides of the K steps, MAX over ranks (rank 1 sleeps twice as long: the reported time must be
    its time), ONE JSON line from rank 0 with the contract's keys and the backend that carried the barrier."""
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "2", "--dry-run"],
                         capture_output=True, text=True, cwd=ROOT, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in r, key
    assert r["n_gpus"] == 2 and r["steps"] == 20 and r["dist_backend"] == "gloo"
    assert r["ms_per_step"] >= 2.0                  # rank 1's 2 ms per "step", not rank 0's 1 ms: the MAX over ranks
    assert "workload" in r["config"]


def test_nccl_failure_falls_back_to_gloo(monkeypatch):
    """bench.init_dist: an RCCL init problem must not cost the scaling run - barrier and MAX move to gloo (here: no GPU at all,
    so the nccl branch fails in-process; world size 1 keeps the test single-process)."""
    m = _bench_module()
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.delenv("FA_BENCH_DIST_BACKEND", raising=False)
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU (the nccl init succeeds here)")
    d, backend = m.init_dist(0)
    try:
        assert backend == "gloo"
        assert m.max_over_ranks(d, backend, 1.25, None) == 1.25
    finally:
        d.destroy_process_group()
