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
    assert 0.95 <= t["sum_over_step"] <= 1.03, t                # vs the wall-clock mean of the timed steps (outliers, host tail)
    for name in ("fwd", "bwd_dq", "bwd_all"):
        assert d["kernels"][name]["ms_min"] <= d["kernels"][name]["ms"]
