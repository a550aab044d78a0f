"""Small-batch decode: evented step time against num_splits (0 = the heuristic), paged 16-bit / fp8 cache.
  python tools/decode_splits_sweep.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn as fa
def t_us(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)[n // 2] * 1e3
SPL = [int(x) for x in os.environ.get("SPLITS", "0,1,4,16,32,64,128,256").split(",")]
print("                                   " + "".join(f"{('s=' + str(s)) if s else 'auto':>8s}" for s in SPL))
CFG = os.environ.get("CFG")
for (B, Hq, Hk, ctx, kv) in [tuple(int(y) if y.isdigit() else y for y in x.split(":")) for x in CFG.split(",")] if CFG else ((1, 32, 8, 4096, "fp16"), (1, 32, 8, 32768, "fp16"), (1, 32, 32, 4096, "fp16"), (1, 64, 8, 8192, "bf16"),
                             (4, 32, 8, 4096, "fp16"), (4, 32, 8, 32768, "fp16"), (16, 32, 8, 4096, "fp16"), (16, 32, 8, 32768, "fp16"),
                             (1, 32, 8, 32768, "fp8"), (8, 32, 32, 8192, "fp8"), (32, 32, 8, 8192, "fp16"), (128, 32, 32, 8192, "fp8")):
    D, page = 128, 256
    dt = torch.bfloat16 if kv == "bf16" else torch.float16
    nblk = B * ctx // page
    kc = torch.randn(nblk, page, Hk, D, device="cuda", dtype=dt); vc = torch.randn_like(kc)
    kw = {}
    if kv == "fp8":
        kc, vc = kc.to(torch.float8_e4m3fn), vc.to(torch.float8_e4m3fn); kw = dict(k_descale=1.0, v_descale=1.0)
    bt = torch.randperm(nblk, device="cuda").to(torch.int32).reshape(B, ctx // page)
    lens = torch.full((B,), ctx - 64, dtype=torch.int32, device="cuda")
    q = torch.randn(B, 1, Hq, D, device="cuda", dtype=dt)
    row = f"B{B:3d} Hq{Hq} Hk{Hk:2d} ctx{ctx:6d} {kv:4s}:"
    for s in SPL:
        try:
            row += f"{t_us(lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, block_table=bt, causal=True, num_splits=s, **kw)):8.1f}"
        except Exception as ex:
            row += "     err"
    print(row, flush=True)
    del kc, vc
