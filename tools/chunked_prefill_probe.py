"""Chunked prefill over a paged cache at small batch: T_q new tokens attend to L cached ones + themselves (causal).
  python tools/chunked_prefill_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn as fa
def t_ms(f, n=8):
    for _ in range(2): f()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)[n // 2]
for kv in ("bf16", "fp8"):
    for (B, Tq, L, Hq, Hk) in ((1, 512, 8192, 32, 8), (1, 2048, 8192, 32, 8), (1, 2048, 32768, 32, 8), (1, 8192, 0, 32, 8), (1, 2048, 131072, 32, 8),
                               (4, 512, 8192, 32, 8), (1, 512, 32768, 64, 8), (1, 256, 16384, 32, 8), (2, 1024, 4096, 32, 32)):
        D, page = 128, 256
        cap = ((L + Tq + page - 1) // page) * page
        nblk = B * cap // page
        kc = torch.randn(nblk, page, Hk, D, device="cuda", dtype=torch.bfloat16); vc = torch.randn_like(kc)
        kw = {}
        if kv == "fp8":
            kc, vc = kc.to(torch.float8_e4m3fn), vc.to(torch.float8_e4m3fn); kw = dict(k_descale=1.0, v_descale=1.0)
        bt = torch.randperm(nblk, device="cuda").to(torch.int32).reshape(B, cap // page)
        lens = torch.full((B,), L, dtype=torch.int32, device="cuda")
        q = torch.randn(B, Tq, Hq, D, device="cuda", dtype=torch.bfloat16)
        kn = torch.randn(B, Tq, Hk, D, device="cuda", dtype=torch.bfloat16); vn = torch.randn_like(kn)
        ms = t_ms(lambda: fa.flash_attn_with_kvcache(q, kc, vc, k=kn, v=vn, cache_seqlens=lens, block_table=bt, causal=True, **kw))
        fl = 4.0 * B * Hq * D * (Tq * L + Tq * (Tq + 1) / 2)
        print(f"{kv:4s} B{B} Tq{Tq:5d} L{L:6d} H{Hq}/{Hk}: {ms:8.3f} ms  {fl / ms / 1e9:6.0f} TFLOP/s", flush=True)
        del kc, vc
