"""Decode benchmark (BASELINE config 4): flash_attn_with_kvcache, batch 128, 32 heads, D 128,
cache_seqlen 8192, paged KV (page 256) + rotary; reports achieved HBM GB/s (K+V bytes / time)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch
import flash_attn


def run(B=128, H=32, Hk=32, D=128, L=8192, page=256, dtype=torch.float16, kv_dtype=None, rotary=True, iters=10):
    dev = "cuda"
    torch.manual_seed(421)
    pages_per_seq = (L + 1 + page - 1) // page
    nblk = B * pages_per_seq
    kv_dtype = kv_dtype or dtype
    if kv_dtype == torch.float8_e4m3fn:
        kc = (torch.randn(nblk, page, Hk, D, device=dev, dtype=torch.float16) * 0.5).to(kv_dtype)
        vc = (torch.randn(nblk, page, Hk, D, device=dev, dtype=torch.float16) * 0.5).to(kv_dtype)
    else:
        kc = torch.randn(nblk, page, Hk, D, device=dev, dtype=dtype)
        vc = torch.randn(nblk, page, Hk, D, device=dev, dtype=dtype)
    bt = torch.randperm(nblk, device=dev).reshape(B, pages_per_seq).to(torch.int32)
    q = torch.randn(B, 1, H, D, device=dev, dtype=dtype)
    kn = torch.randn(B, 1, Hk, D, device=dev, dtype=dtype)
    vn = torch.randn(B, 1, Hk, D, device=dev, dtype=dtype)
    seqlens = torch.full((B,), L, dtype=torch.int32, device=dev)
    cos = sin = None
    if rotary:
        ang = torch.arange(pages_per_seq * page + 8, device=dev)[:, None] * (1.0 / 10000 ** (torch.arange(0, D, 2, device=dev) / D))[None]
        cos, sin = torch.cos(ang).to(dtype), torch.sin(ang).to(dtype)
    kw = {}
    if kv_dtype == torch.float8_e4m3fn:
        kw = dict(k_descale=1.0, v_descale=1.0)
    fn = lambda: flash_attn.flash_attn_with_kvcache(q, kc, vc, k=kn, v=vn, rotary_cos=cos, rotary_sin=sin,
                                                    cache_seqlens=seqlens, block_table=bt, causal=True,
                                                    rotary_interleaved=False, **kw)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    nbytes = 2.0 * B * (L + 1) * Hk * D * kc.element_size()
    print(f"B{B} H{H}/{Hk} D{D} L{L} page{page} kv={kv_dtype}: {ms:.3f} ms  {nbytes/ms/1e6:.0f} GB/s "
          f"({nbytes/ms/1e6/8000*100:.1f}% of 8 TB/s)", flush=True)
    return ms


if __name__ == "__main__":
    run()
    run(Hk=8)
    run(B=8, L=8192)
    if "--fp8" in sys.argv:
        run(kv_dtype=torch.float8_e4m3fn)
        run(Hk=8, kv_dtype=torch.float8_e4m3fn)
