"""Evidence runs for BASELINE.json configs 3, 4 and 5 (bench.py carries config 2, the headline).
One JSON line per config; every rate uses ALGORITHMIC work:
  cfg3  varlen fp16, B 64 mixed seqlens (max 2048), H 32, D 64, window (512, 0): fwd and fwd+bwd
        FLOPs = 4 D H sum_b pairs(L_b), pairs(L) = sum_i min(i+1, 513);
  cfg4  decode, B 128, H 32, D 128, cache_seqlen 8192, paged (page 256) + rotary, fp8-e4m3 and
        fp16 KV, Hk 32 and 8: bytes = K+V cache read once -> GB/s vs 8 TB/s;
  cfg5  dense fwd bf16 causal + ALiBi, B 64, S 8192, D 128, 32 heads sharded over 8 GPUs:
        one GPU's shard (4 heads of every batch), FLOPs = 4 B H S^2 D / 2.
Usage: python tools/bench_configs.py [3] [4] [5]
"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
import flash_attn  # noqa: E402

PEAK_TF, PEAK_GBS = 2500.0, 8000.0


def timeit(fn, iters=10, warm=10, settle_ms=60.0):
    """mean of `iters` back-to-back calls between two events, behind `warm` calls AND >= settle_ms of the same calls: an idle MI355X
    runs its next ~35 ms of launches on a clock ramp (profiles/r06_step_ramp.txt) - 10 warm-up calls of a 0.5 ms kernel end inside it"""
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); e.synchronize()
    for _ in range(max(warm, min(4000, int(settle_ms / max(s.elapsed_time(e), 1e-3)) + 1))):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def cfg3():
    g = torch.Generator().manual_seed(421)
    B, H, D, W = 64, 32, 64, 512
    lens = torch.randint(64, 2049, (B,), generator=g)
    lens[0] = 2048
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = lens.cumsum(0)
    T = int(cu[-1])
    cu = cu.cuda()
    torch.manual_seed(421)
    q, k, v, do = (torch.randn(T, H, D, device="cuda", dtype=torch.float16) for _ in range(4))
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)

    def pairs(L):
        return L * (L + 1) // 2 if L <= W + 1 else (W + 1) * (W + 2) // 2 + (L - W - 1) * (W + 1)

    flops = 4.0 * D * H * sum(pairs(int(L)) for L in lens)
    fwd = lambda: flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, 2048, 2048, causal=True, window_size=(W, 0))

    def fb():
        o = fwd()
        o.backward(do)
        q.grad = k.grad = v.grad = None

    with torch.no_grad():
        t_f = timeit(fwd)
    t_fb = timeit(fb)
    return {"config": "cfg3 varlen fp16 B64 mixed seqlens (max 2048) H32 D64 window (512,0)", "total_tokens": T,
            "fwd_ms": round(t_f, 4), "fwd_tflops": round(flops / t_f / 1e9, 1),
            "fwd_bwd_ms": round(t_fb, 4), "fwd_bwd_tflops": round(3.5 * flops / t_fb / 1e9, 1),
            "fwd_frac_of_mfma_peak": round(flops / t_f / 1e9 / PEAK_TF, 4), "algorithmic_fwd_gflop": round(flops / 1e9, 2)}


def cfg4():
    import bench_decode
    rows = []
    for hk in (32, 8):
        for kvd in (torch.float8_e4m3fn, torch.float16):
            ms = bench_decode.run(Hk=hk, kv_dtype=kvd)
            bpe = 1 if kvd == torch.float8_e4m3fn else 2
            nbytes = 128 * 8192 * hk * 128 * 2 * bpe
            rows.append({"config": f"cfg4 decode B128 H32 Hk{hk} D128 cache 8192 paged(256)+rotary KV {'fp8-e4m3' if bpe == 1 else 'fp16'}",
                         "ms": round(ms, 4), "kv_bytes": nbytes, "achieved_gbs": round(nbytes / ms / 1e6, 1),
                         "frac_of_hbm_peak": round(nbytes / ms / 1e6 / PEAK_GBS, 4)})
    return rows


def cfg5():
    B, Hs, Htot, S, D = 64, 4, 32, 8192, 128          # one of 8 GPUs: heads [0, 4)
    torch.manual_seed(421)
    slopes_all = torch.tensor([2.0 ** (-8.0 * (h + 1) / Htot) for h in range(Htot)], dtype=torch.float32)
    out = []
    for name, sl in (("alibi", slopes_all[:Hs].cuda()), ("no-bias (same shape, for comparison)", None)):
        q, k, v = (torch.randn(B, S, Hs, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
        with torch.no_grad():
            t = timeit(lambda: flash_attn.flash_attn_func(q, k, v, causal=True, alibi_slopes=sl), iters=5, warm=2)
        flops = 4.0 * B * Hs * S * S * D / 2
        out.append({"config": f"cfg5 shard (1 of 8 GPUs): dense fwd bf16 causal {name} B64 H4(of 32) S8192 D128",
                    "ms": round(t, 4), "tflops": round(flops / t / 1e9, 1), "frac_of_mfma_peak": round(flops / t / 1e9 / PEAK_TF, 4)})
        del q, k, v
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["3", "4", "5"]
    res = []
    if "3" in which:
        res.append(cfg3())
    if "4" in which:
        res += cfg4()
    if "5" in which:
        res += cfg5()
    for r in res:
        print(json.dumps(r), flush=True)
