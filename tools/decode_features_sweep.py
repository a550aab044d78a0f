"""Decode step (T_q = 1) with the score modifiers: plain / sliding window / softcap / ALiBi - which kernel serves it and at what cost.
  python tools/decode_features_sweep.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn as fa
def t_us(f, n=20):
    for _ in range(4): f()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)[n // 2] * 1e3
print("us per step".ljust(30) + "".join(f"{x:>12s}" for x in ("plain", "window4k", "softcap", "alibi", "Tq4 softcap", "Tq4 alibi")))
for kv in ("fp16", "fp8"):
    for (B, Hq, Hk, ctx) in ((1, 32, 8, 16384), (8, 32, 8, 16384), (64, 32, 8, 8192), (16, 16, 16, 8192)):
        D, page = 128, 256
        nblk = B * ctx // page
        kc = torch.randn(nblk, page, Hk, D, device="cuda", dtype=torch.float16); vc = torch.randn_like(kc)
        kw = {}
        if kv == "fp8":
            kc, vc = kc.to(torch.float8_e4m3fn), vc.to(torch.float8_e4m3fn); kw = dict(k_descale=1.0, v_descale=1.0)
        bt = torch.randperm(nblk, device="cuda").to(torch.int32).reshape(B, ctx // page)
        lens = torch.full((B,), ctx - 64, dtype=torch.int32, device="cuda")
        slopes = torch.rand(Hq, device="cuda", dtype=torch.float32) * 0.1
        row = f"B{B:3d} Hq{Hq} Hk{Hk:2d} ctx{ctx} {kv:4s}:".ljust(30)
        for T, extra in ((1, {}), (1, dict(window_size=(4096, 0))), (1, dict(softcap=50.0, causal=False)), (1, dict(alibi_slopes=slopes)),
                         (4, dict(softcap=50.0, causal=False)), (4, dict(alibi_slopes=slopes))):
            q = torch.randn(B, T, Hq, D, device="cuda", dtype=torch.float16)
            args = dict(cache_seqlens=lens, block_table=bt, causal=True); args.update(extra); args.update(kw)
            try:
                row += f"{t_us(lambda: fa.flash_attn_with_kvcache(q, kc, vc, **args)):12.1f}"
            except Exception as ex:
                row += f"{'err':>12s}"
        print(row, flush=True)
        del kc, vc
