"""Decode step and chunked prefill against the page size of the paged cache (16 .. 256 tokens): pages of 64 tokens and more take
the aligned fast paths, smaller ones the per-row block-table lookups.   python tools/page_size_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn as fa
def t_us(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)[n // 2] * 1e3
Hq, Hk, D, ctx = 32, 8, 128, 8192
print("us per call".ljust(44) + "".join(f"{'page ' + str(pg):>10s}" for pg in (16, 32, 64, 128, 256)))
for name, B, Tq, kv in (("decode B64 bf16", 64, 1, "bf16"), ("decode B64 fp8", 64, 1, "fp8"), ("decode B8 bf16", 8, 1, "bf16"), ("decode B64 Hq64 (G 8) bf16", 64, 1, "g8"),
                        ("spec decode B8 Tq8 bf16", 8, 8, "bf16"), ("prefill B1 Tq2048 bf16", 1, 2048, "bf16"), ("prefill B8 Tq512 bf16", 8, 512, "bf16")):
    row = name.ljust(44)
    hq = 64 if kv == "g8" else Hq
    for page in (16, 32, 64, 128, 256):
        nblk = B * ctx // page
        kc = torch.randn(nblk, page, Hk, D, device="cuda", dtype=torch.bfloat16); vc = torch.randn_like(kc)
        kw = {}
        if kv == "fp8":
            kc, vc = kc.to(torch.float8_e4m3fn), vc.to(torch.float8_e4m3fn); kw = dict(k_descale=1.0, v_descale=1.0)
        bt = torch.randperm(nblk, device="cuda").to(torch.int32).reshape(B, ctx // page)
        lens = torch.full((B,), ctx - 64 - Tq, dtype=torch.int32, device="cuda")
        q = torch.randn(B, Tq, hq, D, device="cuda", dtype=torch.bfloat16)
        row += f"{t_us(lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens + Tq, block_table=bt, causal=True, **kw)):10.1f}"
        del kc, vc
    print(row, flush=True)
