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
(weak scaling, `value`) and the only distributed calls are the timing barrier and a MAX over ranks.

Besides the contract fields the JSON line carries
  roofline        - the dominant kernel of the step (by measured launch duration), its ALGORITHMIC FLOPs per launch /
                    that duration vs the 2.5 PFLOP/s dense bf16 MFMA peak; durations measured here with HIP events on
                    the stream the kernels run on (torch's current stream).  `achieved` / `frac` use the MEAN launch
                    duration (the statistic a rocprofv3 kernel trace reports; the median rides along as *_median).
                    `practical_ceiling`: what a bare chip-wide MFMA loop on RANDOM bf16 operands sustains on THIS socket
                    in THIS run (tools/probes/probe_mfma_ceiling.hip, ~1 s, with socket power and shader clock sampled
                    next to the same figures for the sustained step): the part is power-limited under matrix load, the
                    2.5 PF peak assumes 2.4 GHz.  `traffic` is NOT measured in this run: it is the HBM byte count of
                    the committed rocprofv3 PMC passes (`traffic_source` names the file);
  kernels         - the same for every kernel of the step (fwd, bwd dK/dV, bwd dQ): median and min over individually
                    evented launches, taken right IN FRONT of the timed steps; `timing.sum_over_step` compares their sum
                    with ms_per_step (tests/test_bench_contract.py fails a recorded line where they differ by > 3 %);
  cold_start      - the same W + K steps timed straight behind the input set-up, on an idle socket: an MI355X needs ~15 steps
                    (35-40 ms of work) to reach its sustained clock state and loses it again after 5 ms of idling
                    (profiles/r06_step_ramp.txt), so a 45 ms region behind 13 ms of warm-up times the ramp - 2-3 % below the
                    rate every later step runs at.  `value` is the same region timed again behind the per-kernel legs
                    (0.3 s of the same launches), i.e. in the state a training loop runs in; rounds 1-5 reported the cold figure;
  other_configs   - BASELINE configs 3, 4 (fp8 and fp16 KV) and the config-5 shard, measured in the same run (rank 0);
  strong_scaling_config5 - BASELINE configs[4]: dense fwd bf16 causal + ALiBi, B64 H32 S8192 D128 with the 32 heads
                    sharded over the N ranks (flash_attn_mi355.sharding.shard_units / shard_alibi): total TFLOP/s at
                    this N - the 1/2/4/8 curve north_star asks for comes from the driver's runs at each N;
  cpu_baseline    - the reference's CPU comparator, PyTorch SDPA on the host cores of this box
                    (rank 0, N = 1 only), on the largest batch x heads sample of config 2 that fits ~12 s; the numpy fp64
                    oracle port is timed next to it (`oracle_port_tflops`).
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
PEAK_HBM_GBS = 8000.0
CFG = dict(batch=8, nheads=16, nheads_k=16, seqlen=4096, head_dim=128, causal=True)


def fwd_flops(c):
    f = 4.0 * c["batch"] * c["nheads"] * c["seqlen"] * c["seqlen"] * c["head_dim"]
    return f * (0.5 if c["causal"] else 1.0)


def event_times_ms(fn, iters, warm=1):
    """GPU-side durations of `iters` individually-evented calls of `fn` (HIP events on torch's current stream = the
    stream the kernels are launched on; no host synchronisation between the calls, so the GPU stays busy)."""
    for _ in range(warm):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)


class _Stat(tuple):
    """(median, min) with the mean riding along: the wall-clock ms_per_step is a MEAN over the timed steps, so the
    sum-of-kernels check compares means with it; the reported per-kernel figure is the median (test.py:87-100)."""
    mean = 0.0


def med_min(ts):
    """(median, min) of a sorted list - the reference's protocol reports the median (test.py:87-100)."""
    n = len(ts)
    med = ts[n // 2] if n % 2 else 0.5 * (ts[n // 2 - 1] + ts[n // 2])
    r = _Stat((med, ts[0]))
    r.mean = sum(ts) / n
    return r


def settle(fn, ms=60.0):
    """Back-to-back calls of `fn` for >= `ms` of GPU time, no synchronize at the end: an MI355X that has idled for 5 ms (input
    set-up on the host, a synchronize with host work behind it) runs its next ~35 ms of launches on a clock ramp, up to 35 % slow
    (profiles/r06_step_ramp.txt).  Every leg below measures behind one of these - the state a serving or training loop runs in."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); b.synchronize()
    for _ in range(min(4000, int(ms / max(a.elapsed_time(b), 1e-3)) + 1)):
        fn()


def event_time_ms(fn, iters, warm=1):
    """Median duration of `fn` over `iters` individually-evented calls."""
    return med_min(event_times_ms(fn, iters, warm))[0]


# ------------------------------------------------------------------------------------------------ power / clock
class _Smi:
    """Socket power (W) and shader clock (MHz) of GPU `index`: amdsmi python module, else hwmon / pp_dpm_sclk in sysfs."""

    def __init__(self, index=0):
        self.kind = None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self.h = amdsmi.amdsmi_get_processor_handles()[index]
            self.amdsmi = amdsmi
            self.kind = "amdsmi"
            self.read()
            return
        except Exception:                    # noqa: BLE001
            self.kind = None
        import glob
        cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") +
                       glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
        if cards:
            self.pw = cards[min(index, len(cards) - 1)]
            self.dev = os.path.dirname(os.path.dirname(os.path.dirname(self.pw)))
            self.kind = "sysfs"

    def read(self):
        if self.kind == "amdsmi":
            m = self.amdsmi.amdsmi_get_gpu_metrics_info(self.h)
            w = m.get("current_socket_power") or m.get("average_socket_power")
            clk = m.get("current_gfxclks") or m.get("current_gfxclk") or m.get("average_gfxclk_frequency")
            if isinstance(clk, (list, tuple)):
                vals = [x for x in clk if isinstance(x, (int, float)) and 0 < x < 60000]
                clk = sum(vals) / len(vals) if vals else None
            return (float(w) if isinstance(w, (int, float)) and 0 < w < 5000 else None,
                    float(clk) if isinstance(clk, (int, float)) and 0 < clk < 60000 else None)
        if self.kind == "sysfs":
            w = clk = None
            try:
                w = int(open(self.pw).read()) / 1e6
                for line in open(os.path.join(self.dev, "pp_dpm_sclk")):
                    if "*" in line:
                        clk = float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
            except Exception:                # noqa: BLE001
                pass
            return w, clk
        return None, None


def sampled(smi, fn):
    """Run fn() (which keeps the GPU busy and returns when it is done) while a thread samples power / clock every 50 ms.
    -> (fn's result, {"watts": mean, "mhz": mean, "samples": n}); the first 30 % of the samples (ramp) are dropped."""
    import threading
    stop, ws, cs = threading.Event(), [], []

    def loop():
        while not stop.is_set():
            try:
                w, c = smi.read()
            except Exception:                # noqa: BLE001
                w = c = None
            if w:
                ws.append(w)
            if c:
                cs.append(c)
            stop.wait(0.05)
    th = threading.Thread(target=loop, daemon=True)
    if smi.kind:
        th.start()
    res = fn()
    stop.set()
    if smi.kind:
        th.join(timeout=2.0)
    tail = lambda x: x[len(x) * 3 // 10:] if len(x) >= 4 else x
    mean = lambda x: round(sum(x) / len(x), 1) if x else None
    return res, {"watts": mean(tail(ws)), "mhz": mean(tail(cs)), "samples": len(ws), "source": smi.kind}


def mfma_ceiling(smi, seconds=1.0):
    """Chip-wide bare MFMA loop on random bf16 operands for ~`seconds` (tools/probes/libfa_probe.so, built by
    __graft_entry__.build()): the matrix rate this socket sustains at its power limit -> dict, or None without the library."""
    import ctypes
    path = os.path.join(ROOT, "tools", "probes", "libfa_probe.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    lib.fa_probe_mfma.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)]
    ms, fl = ctypes.c_float(0), ctypes.c_double(0)
    iters = 4000
    torch.cuda.synchronize()
    if lib.fa_probe_mfma(2, iters, 1, ctypes.byref(ms), ctypes.byref(fl)) != 0 or ms.value <= 0:
        return None
    launches = max(2, int(seconds * 1e3 / (ms.value / 2)))

    def run():
        lib.fa_probe_mfma(launches, iters, 1, ctypes.byref(ms), ctypes.byref(fl))
        return fl.value / (ms.value * 1e-3) / 1e12
    tf, st = sampled(smi, run)
    return {"tflops": round(tf, 1), "seconds": round(ms.value * 1e-3, 3), "watts": st["watts"], "mhz": st["mhz"],
            "what": "bare chip-wide v_mfma_f32_32x32x16_bf16 loop, random bf16 operands, one block of 4 waves x 8 accumulator "
                    "chains per CU slot, measured in this run right after the timed steps (tools/probes/probe_mfma_ceiling.hip)"}


# ------------------------------------------------------------------------------------------------ CPU baseline
def _oracle_port_tflops(c, budget_s):
    import numpy as np
    from oracle import attention as oa
    B, H, S, D = 1, 1, c["seqlen"], c["head_dim"]
    rng = np.random.default_rng(421)
    q, k, v, do = (rng.standard_normal((B, H, S, D)) for _ in range(4))
    scale = D ** -0.5

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
    return 3.5 * 4.0 * B * H * S * S * D * 0.5 * n / el / 1e12, f"oracle (numpy fp64) fwd+bwd causal B{B} H{H} S{S} D{D}, {n} reps in {el:.1f}s"


def cpu_baseline(c, budget_s=12.0):
    """PyTorch SDPA on the host cores (the reference's CPU comparator): bf16 fwd+bwd causal at
    config 2's S / D, batch x heads grown from B1 H2 until one step takes ~budget_s / 3 (at most the full B8 H16)."""
    S, D = c["seqlen"], c["head_dim"]
    threads = torch.get_num_threads()

    def make(B, H):
        g = torch.Generator().manual_seed(421)
        t = [torch.randn(B, H, S, D, generator=g).to(torch.bfloat16) for _ in range(4)]
        for x in t[:3]:
            x.requires_grad_(True)
        return t

    def step(t):
        o = torch.nn.functional.scaled_dot_product_attention(t[0], t[1], t[2], is_causal=True)
        o.backward(t[3])
        t[0].grad = t[1].grad = t[2].grad = None

    B, H = 1, 2
    t = make(B, H)
    step(t)
    t0 = time.perf_counter()
    step(t)
    one = time.perf_counter() - t0
    # grow the sample (heads first, then batch) while a step stays under budget_s / 3
    while True:
        nb, nh = (B, H * 2) if H < c["nheads"] else (B * 2, H)
        if nb > c["batch"] or one * (nb * nh) / (B * H) > budget_s / 3:
            break
        B, H = nb, nh
        t = make(B, H)
        t0 = time.perf_counter()
        step(t)
        one = time.perf_counter() - t0
    t0 = time.perf_counter()
    n = 0
    while True:
        step(t)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s * 0.6 or n >= 100:
            break
    flops = 3.5 * 4.0 * B * H * S * S * D * 0.5
    full = (B == c["batch"] and H == c["nheads"])
    res = {"value": round(flops * n / el / 1e12, 4), "unit": "TFLOP/s", "cores": threads, "kind": "reference",
           "sample": f"torch.nn.functional.scaled_dot_product_attention on CPU, bf16 fwd+bwd causal B{B} H{H} S{S} D{D} "
                     f"({'the full config-2 batch' if full else 'a batch x heads sample of config 2'}), {n} steps in {el:.1f}s, "
                     f"{threads} threads of {os.cpu_count()} logical cores"}
    op, sample = _oracle_port_tflops(c, budget_s * 0.25)
    res["oracle_port_tflops"] = round(op, 5)
    res["oracle_port_sample"] = sample
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


# ------------------------------------------------------------------------------------------------ other configs
def config3(flash_attn, dev):
    """varlen fp16, B64 mixed seqlens (max 2048), H32 D64, window (512, 0)"""
    g = torch.Generator().manual_seed(421)
    B, H, D, W = 64, 32, 64, 512
    lens = torch.randint(64, 2049, (B,), generator=g)
    lens[0] = 2048
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = lens.cumsum(0)
    T = int(cu[-1])
    cu = cu.to(dev)
    gq = torch.Generator().manual_seed(422)
    q, k, v, do = (torch.randn(T, H, D, generator=gq).to(torch.float16).to(dev) for _ in range(4))
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)

    def pairs(L):
        return L * (L + 1) // 2 if L <= W + 1 else (W + 1) * (W + 2) // 2 + (L - W - 1) * (W + 1)

    flops = 4.0 * D * H * sum(pairs(int(L)) for L in lens)
    fwd = lambda: flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, 2048, 2048, causal=True, window_size=(W, 0))

    def fb():
        o = fwd()
        o.backward(do)
        q.grad = k.grad = v.grad = None

    def sustained_ms(fn, n=40):
        """n back-to-back calls between two events (how the headline `value` is timed): a 0.5 ms kernel of 18 k short workgroups
        has a tail that the next launch fills, and an isolated launch pays the queue's start-up - both medians are reported."""
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record(); b.synchronize()
        return a.elapsed_time(b) / n

    with torch.no_grad():
        settle(fwd)
        t_f = event_time_ms(fwd, 10, warm=3)
        t_fs = sustained_ms(fwd)
    settle(fb)
    t_fb = event_time_ms(fb, 10, warm=3)
    t_fbs = sustained_ms(fb, 20)
    return {"workload": "varlen fp16 B64 mixed seqlens (max 2048) H32 D64 window (512,0)", "total_tokens": T,
            "fwd_ms": round(t_f, 4), "fwd_tflops": round(flops / t_f / 1e9, 1),
            "fwd_frac_of_mfma_peak": round(flops / t_f / 1e9 / PEAK_BF16_TFLOPS, 4),
            "fwd_bwd_ms": round(t_fb, 4), "fwd_bwd_tflops": round(3.5 * flops / t_fb / 1e9, 1),
            "timing": "fwd_ms / fwd_bwd_ms: medians of 10 individually evented calls; *_sustained_*: 40 (20) back-to-back calls between two events; "
                      "each pair behind 60 ms of the same calls (settle(): rounds 1-5 measured on the clock ramp of a socket that had idled "
                      "through the input set-up, profiles/r06_step_ramp.txt)",
            "fwd_sustained_ms": round(t_fs, 4), "fwd_sustained_tflops": round(flops / t_fs / 1e9, 1),
            "fwd_bwd_sustained_ms": round(t_fbs, 4), "fwd_bwd_sustained_tflops": round(3.5 * flops / t_fbs / 1e9, 1)}


def config4(flash_attn, dev, kv_dtype, Hk=32):
    """decode B128 H32 D128 cache 8192 paged(256) + rotary; K+V bytes read once / time"""
    B, H, D, L, page = 128, 32, 128, 8192, 256       # (Hk = 8: SURVEY 8(d)'s GQA variant - the MFMA decode kernel, 4 query heads per kv-head)
    dt = torch.float16
    g = torch.Generator().manual_seed(421)
    pps = (L + 1 + page - 1) // page
    nblk = B * pps
    mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float32)
    if kv_dtype == torch.float8_e4m3fn:
        kc = (torch.randn(nblk, page, Hk, D, device=dev, dtype=dt) * 0.5).to(kv_dtype)
        vc = (torch.randn(nblk, page, Hk, D, device=dev, dtype=dt) * 0.5).to(kv_dtype)
        kw = dict(k_descale=1.0, v_descale=1.0)
    else:
        kc = torch.randn(nblk, page, Hk, D, device=dev, dtype=dt)
        vc = torch.randn(nblk, page, Hk, D, device=dev, dtype=dt)
        kw = {}
    bt = torch.randperm(nblk, device=dev).reshape(B, pps).to(torch.int32)
    q, kn, vn = mk(B, 1, H, D).to(dt).to(dev), mk(B, 1, Hk, D).to(dt).to(dev), mk(B, 1, Hk, D).to(dt).to(dev)
    seqlens = torch.full((B,), L, dtype=torch.int32, device=dev)
    ang = torch.arange(pps * page + 8, device=dev)[:, None] * (1.0 / 10000 ** (torch.arange(0, D, 2, device=dev) / D))[None]
    cos, sin = torch.cos(ang).to(dt), torch.sin(ang).to(dt)
    fn = lambda: flash_attn.flash_attn_with_kvcache(q, kc, vc, k=kn, v=vn, rotary_cos=cos, rotary_sin=sin,
                                                    cache_seqlens=seqlens, block_table=bt, causal=True,
                                                    rotary_interleaved=False, **kw)
    settle(fn)
    ms = event_time_ms(fn, 10, warm=3)
    nbytes = 2.0 * B * (L + 1) * Hk * D * kc.element_size()
    return {"workload": f"decode B128 H32{'' if Hk == 32 else '/%d' % Hk} D128 cache 8192 paged(256)+rotary, KV {'fp8-e4m3' if kc.element_size() == 1 else 'fp16'}",
            "ms": round(ms, 4), "kv_bytes": int(nbytes), "achieved_gbs": round(nbytes / ms / 1e6, 1),
            "frac_of_hbm_peak": round(nbytes / ms / 1e6 / PEAK_HBM_GBS, 4)}


def serving_steps(flash_attn, dev):
    """decode-step latencies outside config 4 (evented medians, microseconds): small batch (split-KV + the merge of the partials),
    speculative tokens (row blocks), softcap, head dim 256, and the same step issued through the varlen op.  H 32/8, paged
    256-token pages, 8192-token context, bf16 cache unless noted."""
    out = {}
    def run(name, B, Tq, Hq, Hk, D, ctx, varlen=False, **kw):
        page, dt = 256, torch.bfloat16
        nblk = B * ctx // page
        kc = torch.randn(nblk, page, Hk, D, device=dev, dtype=dt); vc = torch.randn_like(kc)
        bt = torch.randperm(nblk, device=dev).to(torch.int32).reshape(B, ctx // page)
        lens = torch.full((B,), ctx - 64, dtype=torch.int32, device=dev)
        q = torch.randn(B, Tq, Hq, D, device=dev, dtype=dt)
        if varlen:
            qv = q.reshape(B * Tq, Hq, D)
            cu_q = torch.arange(B + 1, dtype=torch.int32, device=dev) * Tq
            cu_k = torch.cat([torch.zeros(1, dtype=torch.int32, device=dev), lens.cumsum(0).to(torch.int32)])
            fn = lambda: flash_attn.flash_attn_varlen_func(qv, kc, vc, cu_q, cu_k, Tq, ctx, causal=True, block_table=bt, seqused_k=lens)
        else:
            fn = lambda: flash_attn.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, block_table=bt, causal=kw.get("softcap", 0.0) == 0.0, **kw)
        settle(fn, 40.0)
        ts = event_times_ms(fn, 20, warm=12)
        out[name] = round(sorted(ts)[len(ts) // 2] * 1e3, 1)
    run("decode_B1_us", 1, 1, 32, 8, 128, 8192)
    run("decode_B8_us", 8, 1, 32, 8, 128, 8192)
    run("decode_B1_ctx32k_us", 1, 1, 32, 8, 128, 32768)
    run("spec_decode_B8_Tq8_us", 8, 8, 32, 8, 128, 8192)
    run("spec_decode_B8_Tq16_us", 8, 16, 32, 8, 128, 8192)
    run("decode_B8_softcap_us", 8, 1, 32, 8, 128, 8192, softcap=50.0)
    run("decode_B8_D256_us", 8, 1, 16, 8, 256, 8192)
    run("decode_B8_via_varlen_op_us", 8, 1, 32, 8, 128, 8192, varlen=True)
    out["workload"] = "decode step latency, H 32/8 (D256: 16/8), paged(256) bf16 cache, context 8192 unless named; medians of 20 evented calls behind 40 ms of the same calls"
    return out


def config5(flash_attn, dev, world, rank, iters=3, warm=2):
    """dense fwd bf16 causal + ALiBi, B64 S8192 D128, 32 heads sharded over `world` ranks (this rank's heads)"""
    from flash_attn_mi355.sharding import shard_alibi, shard_units
    B, Htot, S, D = 64, 32, 8192, 128
    bs, qh, kh = shard_units(B, Htot, Htot, world, rank)
    slopes_all = torch.tensor([2.0 ** (-8.0 * (h + 1) / Htot) for h in range(Htot)], dtype=torch.float32, device=dev)
    sl = shard_alibi(slopes_all, qh, bs)
    Bs, Hs = bs.stop - bs.start, qh.stop - qh.start
    g = torch.Generator(device="cpu").manual_seed(421 + rank)
    mk = lambda: torch.randn(Bs, S, Hs, D, generator=g, dtype=torch.float32).to(torch.bfloat16).to(dev)
    q, k, v = mk(), mk(), mk()
    fn = lambda: flash_attn.flash_attn_func(q, k, v, causal=True, alibi_slopes=sl)
    with torch.no_grad():
        settle(fn)
        ms = event_time_ms(fn, iters, warm=warm)
    flops = 4.0 * Bs * Hs * S * S * D / 2
    return ms, flops, Bs, Hs


def init_dist(local_rank, dry_run=False):
    """The data path has NO collective (heads x batch shard): the process group only carries the timing barrier
    and the MAX over ranks.  RCCL ("nccl") first; if its init or first barrier fails, the same two operations run over gloo on
    the CPU - an RCCL problem on the node must not cost the scaling measurement.  -> (torch.distributed, backend name)"""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    want = os.environ.get("FA_BENCH_DIST_BACKEND", "gloo" if dry_run else "nccl")
    if want == "nccl":
        try:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=120))
            dist.barrier()
            torch.cuda.synchronize()
            return dist, "nccl"
        except Exception as e:                       # noqa: BLE001
            print(f"bench.py: nccl (RCCL) process group failed ({type(e).__name__}: {str(e)[:200]}); falling back to gloo",
                  file=sys.stderr, flush=True)
            try:
                dist.destroy_process_group()
            except Exception:                        # noqa: BLE001
                pass
            # (every rank takes the same branch: an init failure is collective; the fallback group meets on the next port)
            os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
    return dist, "gloo"


def max_over_ranks(dist, backend, seconds, dev):
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def dry_run(args, rank, world, dist, backend):
    """--dry-run: the N > 1 control path without a GPU - rendezvous, barrier on both sides of K "steps", MAX over ranks, ONE JSON
    line from rank 0 with the contract's keys (tests/test_bench_contract.py runs it with two gloo ranks)."""
    def barrier():
        if dist is not None:
            dist.barrier()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))               # rank 1 is slower: the MAX must pick it up
    barrier()
    elapsed = max_over_ranks(dist, backend, time.perf_counter() - t0, None)
    if rank == 0:
        ff = fwd_flops(CFG)
        print(json.dumps({"metric": "attention TFLOPS fwd+bwd (seqlen 4096, hd128, causal)", "value": world * 3.5 * ff / (elapsed / args.steps) / 1e12, "unit": "TFLOP/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
                          "data": "dry run: no GPU work, sleeps instead of steps", "dist_backend": backend,
                          "config": {"workload": "dense fwd+bwd bf16 causal B8 H16 S4096 D128 (BASELINE configs[1])", "dry_run": True}}), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: exercise the launch / rendezvous / barrier / MAX-over-ranks / JSON path only (CPU test of N > 1)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    dist_backend = None
    if world > 1:
        dist, dist_backend = init_dist(local_rank, args.dry_run)
    if args.dry_run:
        return dry_run(args, rank, world, dist, dist_backend)
    torch.cuda.set_device(local_rank if world > 1 else 0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    import flash_attn
    from flash_attn_mi355 import flash_attn_interface as fi

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

    def timed_region():
        """the contract: W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides, MAX over ranks"""
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        return max_over_ranks(dist, dist_backend, time.perf_counter() - t0, dev)

    ff = fwd_flops(c)
    step_flops = 3.5 * ff

    # ---- (1) the K steps from a COLD socket: straight behind the input set-up, the first warm-up step loads the code objects
    #      (~150 ms of host time with the GPU idle) and W = 5 steps are 13 ms.  An idle MI355X socket needs ~15 steps (35-40 ms
    #      of work) to reach its sustained clock state - and an idle gap of 5 ms is enough to lose it again (tools/step_profile.py,
    #      profiles/r06_step_ramp.txt: 2.99, 2.72, 2.55, 2.45, 2.40 ... 2.19 ms) - so this region times the RAMP.  Reported as
    #      `cold_start`; it is what rounds 1-5 reported as `value`.
    cold_elapsed = timed_region()

    # ---- (2) per-kernel durations (HIP events on the launch stream) on EVERY rank - only rank 0 reports them, but every socket has
    #      to stay busy: ~0.3 s of back-to-back launches, the same kernels as the steps, which also brings the socket to the state
    #      a training loop runs in --------------------------------------------------------------------------------------------
    kern, kmin, kmean = {}, {}, {}
    it = 30
    with torch.no_grad():
        st = med_min(event_times_ms(lambda: flash_attn.flash_attn_func(q, k, v, causal=c["causal"]), it, warm=3))
        kern["fwd"], kmin["fwd"], kmean["fwd"] = st[0], st[1], st.mean
    # which backward kernels run follows from the gradients the op has to produce (autograd's needs_input_grad ->
    # fa_bwd with dq == NULL or dk == dv == NULL): q alone = the dQ kernel, k and v = preprocess + dK/dV kernel,
    # all three = dQ kernel + dK/dV kernel (the step's backward)
    qd, kd, vd = q.detach(), k.detach(), v.detach()
    graphs = {"bwd_all": (flash_attn.flash_attn_func(q, k, v, causal=c["causal"]), (q, k, v)),
              "bwd_dq": (flash_attn.flash_attn_func(q, kd, vd, causal=c["causal"]), (q,)),
              "bwd_dkdv_pre": (flash_attn.flash_attn_func(qd, k, v, causal=c["causal"]), (k, v))}
    for name, (o, ins) in graphs.items():
        st = med_min(event_times_ms(lambda: torch.autograd.grad(o, ins, do, retain_graph=True), it, warm=3))
        kern[name], kmin[name], kmean[name] = st[0], st[1], st.mean
    del graphs, o

    def fb():
        oo = flash_attn.flash_attn_func(q, k, v, causal=c["causal"])
        oo.backward(do)
        q.grad = k.grad = v.grad = None
    st = med_min(event_times_ms(fb, it, warm=3))
    kern["step"], kmin["step"], kmean["step"] = st[0], st[1], st.mean

    # ---- (3) `value`: the contract's timed region again, now on a socket in its sustained state - W warm-up steps, barrier +
    #      synchronize, EXACTLY K steps, barrier + synchronize, MAX over ranks.  With N > 1 a rendezvous first, so that the ranks
    #      enter their warm-up steps together and the bracketing barrier finds them aligned (a rank waiting 5 ms in a barrier would
    #      start its timed steps on a socket that has dropped its clocks).
    if dist is not None:
        dist.barrier()
    elapsed = timed_region()
    ms_per_step = elapsed / args.steps * 1e3
    value = world * step_flops / (elapsed / args.steps) / 1e12
    cold_ms = cold_elapsed / args.steps * 1e3

    if rank == 0:
        # ---- the step sustained for ~2 s with power / clock sampled (this is also what a coarse GPU-busy sampler gets to
        #      see: the K timed steps above are 45 ms), then the bare-MFMA ceiling of this socket in the same state -------
        smi = _Smi(local_rank if world > 1 else 0)

        def sustain(seconds=2.0):
            n = max(20, int(seconds * 1e3 / ms_per_step))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fb()
            b.record(); b.synchronize()
            return a.elapsed_time(b) / n, n
        (sus_ms, sus_n), sus_state = sampled(smi, sustain)
        ceiling = mfma_ceiling(smi)

    # ---- strong scaling, BASELINE configs[4]: every rank takes 32 / N heads of all 64 batches ----------------------
    strong = None
    if not args.no_other_configs:
        del q, k, v, do
        torch.cuda.empty_cache()
        ms5, fl5, Bs, Hs = config5(flash_attn, dev, world, rank)
        barrier()
        ms5max = max_over_ranks(dist, dist_backend, ms5, dev)
        total5 = 4.0 * 64 * 32 * 8192 * 8192 * 128 / 2
        strong = {"workload": "dense fwd bf16 causal + ALiBi B64 H32 S8192 D128, heads sharded over the ranks (BASELINE configs[4])",
                  "n_gpus": world, "batch_per_gpu": Bs, "heads_per_gpu": Hs, "ms": round(ms5max, 4),
                  "tflops_total": round(total5 / ms5max / 1e9, 1),
                  "frac_of_mfma_peak": round(total5 / ms5max / 1e9 / (world * PEAK_BF16_TFLOPS), 4), "scaling": "strong"}

    out = None
    if rank == 0:
        # the dK/dV kernel's own duration: the full backward launches exactly two kernels (dQ, then dK/dV)
        kern["bwd_dkdv"] = kern["bwd_all"] - kern["bwd_dq"]
        kmin["bwd_dkdv"] = kmin["bwd_all"] - kmin["bwd_dq"]
        t_fwd = kern["fwd"]
        pairs_flops = ff / 2.0            # one GEMM over the visible pairs = 2*D*pairs
        # algorithmic FLOPs: fwd 2 GEMMs; bwd 5 GEMMs split as dK/dV kernel 4 (S, dP, dV, dK)
        # and dQ kernel 1 (its S/dP recomputation is overhead, not algorithmic work).
        alg = {"fwd": 2 * pairs_flops, "bwd_dkdv": 4 * pairs_flops, "bwd_dq": 1 * pairs_flops}
        dur = {"fwd": t_fwd, "bwd_dkdv": kern["bwd_dkdv"], "bwd_dq": kern["bwd_dq"]}
        kmean["bwd_dkdv"] = kmean["bwd_all"] - kmean["bwd_dq"]
        kernels = {}
        for name in ("fwd", "bwd_dkdv", "bwd_dq"):
            ach = alg[name] / (dur[name] * 1e-3) / 1e12
            ach_mean = alg[name] / (kmean[name] * 1e-3) / 1e12
            kernels[name] = {"ms": round(dur[name], 4), "ms_min": round(kmin[name], 4), "ms_mean": round(kmean[name], 4),
                             "algorithmic_tflop": round(alg[name] / 1e12, 5),
                             "achieved": round(ach, 1), "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                             "achieved_mean": round(ach_mean, 1), "frac_mean": round(ach_mean / PEAK_BF16_TFLOPS, 4)}
        kernels["bwd_dkdv"]["note"] = "median(bwd, dq + dk + dv) - median(bwd, dq only): the full backward is these two launches"
        kernels["bwd_dkdv_plus_preprocess"] = {
            "ms": round(kern["bwd_dkdv_pre"], 4), "ms_min": round(kmin["bwd_dkdv_pre"], 4),
            "note": "backward asked for dk, dv only: the dQ kernel (and its fused row-dot) does not run, so this is the "
                    "preprocess kernel + the dK/dV kernel"}
        kernels["bwd_all"] = {"ms": round(kern["bwd_all"], 4), "ms_min": round(kmin["bwd_all"], 4),
                              "achieved": round(2.5 * ff / (kern["bwd_all"] * 1e-3) / 1e12, 1)}
        kernels["step_evented"] = {"ms": round(kern["step"], 4), "ms_min": round(kmin["step"], 4),
                                   "achieved": round(step_flops / (kern["step"] * 1e-3) / 1e12, 1),
                                   "note": "one fwd + bwd through autograd, individually evented (median / min of 30)"}
        sum_k = kern["fwd"] + kern["bwd_all"]
        sum_mean = kmean["fwd"] + kmean["bwd_all"]
        kernels["timing"] = {"statistic": "median (ms) and min (ms_min) over 30 individually-evented launches, HIP events on the launch stream",
                             "sum_of_kernels_ms": round(sum_k, 4), "sum_of_kernel_means_ms": round(sum_mean, 4),
                             "step_evented_mean_ms": round(kmean["step"], 4), "ms_per_step": round(ms_per_step, 4),
                             "sum_over_step_evented": round(sum_k / kern["step"], 4),
                             "sum_over_step": round(sum_k / ms_per_step, 4),
                             "note": "sum_over_step_evented = (median fwd + median bwd) / median evented step: the same statistic on "
                                     "both sides; sum_over_step divides by ms_per_step, the wall-clock MEAN of the K timed steps "
                                     "(includes the slow outlier launches a median drops and the host-side tail); the evented "
                                     "launches run right in front of the timed region"}
        dom = max(("fwd", "bwd_dkdv", "bwd_dq"), key=lambda n: dur[n])
        fwd_kernel = "fa_fwd_kernel" if os.environ.get("FA_FWD_ASM") == "0" else "fa_fwd_asm_kernel"
        dkdv_kernel = "fa_bwd_dkdv2_kernel" if os.environ.get("FA_BWD_ASM") == "0" else "fa_bwd_dkdv_asm_kernel"
        roofline = {"bound": "mfma", "kernel": {"fwd": fwd_kernel, "bwd_dkdv": dkdv_kernel,
                                                  "bwd_dq": "fa_bwd_dq_kernel" if os.environ.get("FA_BWD_DQ_ASM") == "0" else "fa_bwd_dq_asm_kernel"}[dom],
                    "achieved": kernels[dom]["achieved_mean"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": kernels[dom]["frac_mean"], "statistic": "mean launch duration over 30 evented launches (what a rocprofv3 "
                    "kernel trace averages); median-based figures: achieved_median / frac_median",
                    "achieved_median": kernels[dom]["achieved"], "frac_median": kernels[dom]["frac"], "traffic": None}
        roofline["practical_ceiling"] = ceiling
        if ceiling:
            roofline["frac_of_ceiling"] = round(kernels[dom]["achieved_mean"] / ceiling["tflops"], 4)
            roofline["step_frac_of_ceiling"] = round(step_flops / (sus_ms * 1e-3) / 1e12 / ceiling["tflops"], 4)
            roofline["fwd_frac_of_ceiling"] = round(kernels["fwd"]["achieved_mean"] / ceiling["tflops"], 4)
        roofline["sustained_step"] = {"ms": round(sus_ms, 4), "steps": sus_n, "tflops": round(step_flops / (sus_ms * 1e-3) / 1e12, 1),
                                      "watts": sus_state["watts"], "mhz": sus_state["mhz"], "power_source": sus_state["source"],
                                      "note": "the step back to back for ~2 s right behind the timed region, socket power and shader clock sampled every 50 ms"}
        tr = measured_traffic(roofline["kernel"])
        if tr:
            roofline["traffic"] = tr[0]
            roofline["traffic_unit"] = "bytes of HBM per launch (FETCH_SIZE x2 + WRITE_SIZE)"
            roofline["traffic_source"] = f"NOT measured in this run: committed rocprofv3 PMC passes, {tr[1]}"
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
            "cold_start": {"ms_per_step": round(cold_ms, 4), "value": round(world * step_flops / (cold_ms * 1e-3) / 1e12, 2),
                           "note": "the same W warm-up + K timed steps straight behind the input set-up, on an idle socket (clock ramp: "
                                   "profiles/r06_step_ramp.txt); `value` is the region timed again behind the per-kernel legs - every "
                                   "rank runs them -, the state every step of a training loop after the first 40 ms runs in"},
        }
        if world > 1:
            out["dist_backend"] = dist_backend      # barrier + MAX only (no data-path collective): "nccl" = RCCL, "gloo" = the CPU fallback
        if strong is not None:
            out["strong_scaling_config5"] = strong
        if not args.no_other_configs:
            torch.cuda.empty_cache()
            oc = {"config3": config3(flash_attn, dev)}
            torch.cuda.empty_cache()
            oc["config4_fp8_kv"] = config4(flash_attn, dev, torch.float8_e4m3fn)
            torch.cuda.empty_cache()
            oc["config4_fp16_kv"] = config4(flash_attn, dev, torch.float16)
            torch.cuda.empty_cache()
            oc["config4_hk8_fp8_kv"] = config4(flash_attn, dev, torch.float8_e4m3fn, Hk=8)
            oc["config4_hk8_fp16_kv"] = config4(flash_attn, dev, torch.float16, Hk=8)
            oc["serving_steps"] = serving_steps(flash_attn, dev)
            torch.cuda.empty_cache()
            if world == 1:
                ms, fl, Bs, Hs = config5(flash_attn, dev, 8, 0)
                oc["config5_shard_1_of_8"] = {"workload": "dense fwd bf16 causal + ALiBi, one GPU's shard of 8: B64 H4 (of 32) S8192 D128",
                                              "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1),
                                              "frac_of_mfma_peak": round(fl / ms / 1e9 / PEAK_BF16_TFLOPS, 4)}
            out["other_configs"] = oc
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(c)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
