#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X fused attention path.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
  dense forward + backward, bf16, causal, batch 8, 16 heads, seqlen 4096, head_dim 128,
  synthetic N(0,1) Q/K/V/dO (never zero-filled), resident in HBM before the timed region.
A "step" = one forward + one backward of that batch through the public Python API
(flash_attn.flash_attn_func -> C ABI -> HIP kernels).  FLOPs use the FlashAttention
convention: fwd = 4*B*H*S*S*D/2 (causal), bwd = 2.5 x fwd.

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver through torch.distributed.run (one rank per GPU); the
path shards batch x heads with no collective, so every rank runs the full per-GPU workload
(weak scaling) and the only distributed calls are the timing barrier and a MAX over ranks.

Besides the contract fields the JSON line carries
  roofline      - the dominant kernel of the step (by measured launch duration), its
                  ALGORITHMIC FLOPs per launch / that duration vs the 2.5 PFLOP/s dense
                  bf16 MFMA peak; durations measured here with HIP events on the stream the
                  kernels run on (torch's current stream);
  kernels       - the same for every kernel of the step (fwd, bwd dK/dV, bwd dQ, preprocess);
  cpu_baseline  - the oracle (numpy fp64 restatement of the reference algorithm) on the host
                  cores (rank 0, N = 1 only), bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "flash-attention-v100_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md: ~2.5 PF dense)
CFG = dict(batch=8, nheads=16, nheads_k=16, seqlen=4096, head_dim=128, causal=True)


def fwd_flops(c):
    f = 4.0 * c["batch"] * c["nheads"] * c["seqlen"] * c["seqlen"] * c["head_dim"]
    return f * (0.5 if c["causal"] else 1.0)


def event_time_ms(fn, iters):
    """Average duration of `fn` (kernel launches only) over `iters` back-to-back calls."""
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def cpu_baseline(c, budget_s=15.0):
    """The oracle (oracle/attention.py: the numpy fp64 restatement of the reference algorithm,
    fwd + bwd) timed on the host cores on a bounded sample of the workload: 1 batch x 1 head
    of S4096 D128 causal per repetition, for ~budget_s seconds.  numpy's BLAS supplies the
    threading.  The torch-SDPA CPU path (bf16) is timed next to it for orientation."""
    import numpy as np
    from oracle import attention as oa
    B, H, S, D = 1, 1, c["seqlen"], c["head_dim"]
    rng = np.random.default_rng(421)
    q, k, v, do = (rng.standard_normal((B, H, S, D)) for _ in range(4))
    scale = D ** -0.5
    try:
        from threadpoolctl import threadpool_info
        cores = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1

    def ostep():
        o, lse, _ = oa.attn_fwd(q, k, v, scale, causal=True)
        oa.attn_bwd(do, q, k, v, o, lse, scale, causal=True)

    ostep()
    t0 = time.perf_counter()
    n = 0
    while True:
        ostep()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    flops = 3.5 * 4.0 * B * H * S * S * D * 0.5
    res = {"value": round(flops * n / el / 1e12, 5), "unit": "TFLOP/s", "cores": cores, "kind": "port",
           "sample": f"oracle (numpy fp64) fwd+bwd causal B{B} H{H} S{S} D{D}, {n} reps in {el:.1f}s"}

    # orientation only: PyTorch's own CPU attention in bf16 on the same shape family
    Bt, Ht = 1, 2
    g = torch.Generator().manual_seed(421)
    tq, tk, tv, tdo = (torch.randn(Bt, Ht, S, D, generator=g).to(torch.bfloat16) for _ in range(4))
    tq.requires_grad_(True); tk.requires_grad_(True); tv.requires_grad_(True)

    def tstep():
        o = torch.nn.functional.scaled_dot_product_attention(tq, tk, tv, is_causal=True)
        o.backward(tdo)
        tq.grad = tk.grad = tv.grad = None

    tstep()
    t0 = time.perf_counter()
    n = 0
    while True:
        tstep()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s / 3 or n >= 100:
            break
    res["torch_sdpa_cpu_tflops"] = round(3.5 * 4.0 * Bt * Ht * S * S * D * 0.5 * n / el / 1e12, 4)
    res["torch_sdpa_cpu_threads"] = torch.get_num_threads()
    return res


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/*_traffic.json, written by tools/collect_profiles.sh on the same workload)."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        for name, v in d.get("kernels", {}).items():
            if name.startswith(kernel + "<") and "bf16" in name:
                best = (v["total_bytes"], os.path.relpath(path, ROOT))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    import flash_attn
    from flash_attn_mi355 import _lib

    c = CFG
    B, H, Hk, S, D = c["batch"], c["nheads"], c["nheads_k"], c["seqlen"], c["head_dim"]
    g = torch.Generator(device="cpu").manual_seed(421 + rank)
    mk = lambda h: torch.randn(B, S, h, D, generator=g).to(torch.bfloat16).to(dev)
    q, k, v, do = mk(H), mk(Hk), mk(Hk), mk(H)
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)

    def step():
        o = flash_attn.flash_attn_func(q, k, v, causal=c["causal"])
        o.backward(do)
        q.grad = k.grad = v.grad = None

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ff = fwd_flops(c)
    step_flops = 3.5 * ff
    ms_per_step = elapsed / args.steps * 1e3
    value = world * step_flops / (elapsed / args.steps) / 1e12

    out = None
    if rank == 0:
        # ---- per-kernel durations (HIP events on the launch stream) --------------------
        it = 20
        with torch.no_grad():
            t_fwd = event_time_ms(lambda: flash_attn.flash_attn_func(q, k, v, causal=c["causal"]), it)
        o = flash_attn.flash_attn_func(q, k, v, causal=c["causal"])

        def bwd_only():
            torch.autograd.grad(o, (q, k, v), do, retain_graph=True)

        kern = {}
        for name, mask in (("bwd_preprocess", 1), ("bwd_dkdv", 2), ("bwd_dq", 4), ("bwd_all", 7)):
            _lib.lib.fa_debug_set_bwd_phases(mask)
            kern[name] = event_time_ms(bwd_only, it)
        _lib.lib.fa_debug_set_bwd_phases(7)
        pairs_flops = ff / 2.0            # one GEMM over the visible pairs = 2*D*pairs
        # algorithmic FLOPs: fwd 2 GEMMs; bwd 5 GEMMs split as dK/dV kernel 4 (S, dP, dV, dK)
        # and dQ kernel 1 (its S/dP recomputation is overhead, not algorithmic work).
        alg = {"fwd": 2 * pairs_flops, "bwd_dkdv": 4 * pairs_flops, "bwd_dq": 1 * pairs_flops}
        dur = {"fwd": t_fwd, "bwd_dkdv": kern["bwd_dkdv"], "bwd_dq": kern["bwd_dq"]}
        kernels = {}
        for name in ("fwd", "bwd_dkdv", "bwd_dq"):
            ach = alg[name] / (dur[name] * 1e-3) / 1e12
            kernels[name] = {"ms": round(dur[name], 4), "algorithmic_tflop": round(alg[name] / 1e12, 5),
                             "achieved": round(ach, 1), "frac": round(ach / PEAK_BF16_TFLOPS, 4)}
        kernels["bwd_preprocess"] = {"ms": round(kern["bwd_preprocess"], 4)}
        kernels["bwd_all"] = {"ms": round(kern["bwd_all"], 4),
                              "achieved": round(2.5 * ff / (kern["bwd_all"] * 1e-3) / 1e12, 1)}
        dom = max(("fwd", "bwd_dkdv", "bwd_dq"), key=lambda n: dur[n])
        roofline = {"bound": "mfma", "kernel": {"fwd": "fa_fwd_kernel", "bwd_dkdv": "fa_bwd_dkdv2_kernel",
                                                  "bwd_dq": "fa_bwd_dq_kernel"}[dom],
                    "achieved": kernels[dom]["achieved"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": kernels[dom]["frac"], "traffic": None}
        tr = measured_traffic(roofline["kernel"])
        if tr:
            roofline["traffic"] = tr[0]
            roofline["traffic_unit"] = "bytes of HBM per launch (FETCH_SIZE x2 + WRITE_SIZE)"
            roofline["traffic_source"] = tr[1]
        out = {
            "metric": "attention TFLOPS fwd+bwd (seqlen 4096, hd128, causal)",
            "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "dense fwd+bwd bf16 causal B8 H16 S4096 D128 (BASELINE configs[1])",
                       "batch_per_gpu": B, "nheads": H, "nheads_k": Hk, "seqlen": S, "head_dim": D,
                       "causal": True, "sharding": "batch x heads per GPU, no collective"},
            "frac_of_mfma_peak": round(value / world / PEAK_BF16_TFLOPS, 4),
            "fwd_tflops": kernels["fwd"]["achieved"], "fwd_frac_of_mfma_peak": kernels["fwd"]["frac"],
            "roofline": roofline, "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(c)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
