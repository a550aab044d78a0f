"""Decode at small batch: per-kernel device time (run under rocprofv3 --kernel-trace --stats) and evented step time.
  python tools/decode_small_batch.py [B ...]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn as fa
for B in [int(x) for x in sys.argv[1:]] or [1, 4, 16]:
    for (Hq, Hk, ctx, dt) in ((32, 8, 4096, torch.float16), (32, 8, 32768, torch.float16), (32, 32, 4096, torch.float16), (64, 8, 8192, torch.bfloat16)):
        D, page = 128, 256
        nblk = B * ctx // page
        kc = torch.randn(nblk, page, Hk, D, device="cuda", dtype=dt); vc = torch.randn_like(kc)
        bt = torch.randperm(nblk, device="cuda").to(torch.int32).reshape(B, ctx // page)
        lens = torch.full((B,), ctx - 64, dtype=torch.int32, device="cuda")
        q = torch.randn(B, 1, Hq, D, device="cuda", dtype=dt)
        f = lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, block_table=bt, causal=True)
        for _ in range(5): f()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for s, e in evs:
            s.record(); f(); e.record()
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) for s, e in evs)
        gb = 2.0 * (ctx - 64) * Hk * D * 2 * B / 1e9
        print(f"B{B:3d} Hq{Hq} Hk{Hk} ctx{ctx:6d}: {ts[15]*1e3:7.1f} us (min {ts[0]*1e3:6.1f})  {gb/ts[15]*1e3/1e3:5.2f} TB/s", flush=True)
