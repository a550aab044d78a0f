"""Config 4 at H_k = 8: does the rate depend on WHERE the pages lie?  Same step with (a) the bench's random page permutation,
(b) pages in order (block table = identity), (c) 1024- and 2048-token pages (random), (d) no paging at all (contiguous cache,
cache_batch_idx-free).  fp8 and fp16 caches; medians of 15 evented calls."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn
dev = torch.device("cuda", 0)

def step(kv_dtype, page, order, B=128, H=32, Hk=8, L=8192, D=128, n=15):
    dt = torch.float16
    kw = dict(k_descale=1.0, v_descale=1.0) if kv_dtype == torch.float8_e4m3fn else {}
    q = torch.randn(B, 1, H, D, device=dev, dtype=dt); kn = torch.randn(B, 1, Hk, D, device=dev, dtype=dt); vn = torch.randn_like(kn)
    seqlens = torch.full((B,), L, dtype=torch.int32, device=dev)
    if page:
        pps = (L + 1 + page - 1) // page
        nblk = B * pps
        kc = (torch.randn(nblk, page, Hk, D, device=dev, dtype=dt) * 0.5).to(kv_dtype); vc = (torch.randn(nblk, page, Hk, D, device=dev, dtype=dt) * 0.5).to(kv_dtype)
        bt = (torch.randperm(nblk, device=dev) if order == "random" else torch.arange(nblk, device=dev)).reshape(B, pps).to(torch.int32)
        rows = pps * page
    else:
        rows = L + 8
        kc = (torch.randn(B, rows, Hk, D, device=dev, dtype=dt) * 0.5).to(kv_dtype); vc = (torch.randn(B, rows, Hk, D, device=dev, dtype=dt) * 0.5).to(kv_dtype)
        bt = None
    ang = torch.arange(rows + 8, device=dev)[:, None] * (1.0 / 10000 ** (torch.arange(0, D, 2, device=dev) / D))[None]
    cos, sin = torch.cos(ang).to(dt), torch.sin(ang).to(dt)
    fn = lambda: flash_attn.flash_attn_with_kvcache(q, kc, vc, k=kn, v=vn, rotary_cos=cos, rotary_sin=sin, cache_seqlens=seqlens,
                                                    block_table=bt, causal=True, rotary_interleaved=False, **kw)
    for _ in range(4):
        fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = sorted(ts)[len(ts) // 2]
    nbytes = 2.0 * B * (L + 1) * Hk * D * kc.element_size()
    print(f"H {H}/{Hk} {'fp8 ' if kc.element_size() == 1 else 'fp16'} page {page or 'none':>5} {order:8s}: {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:6.2f} TB/s", flush=True)
    del kc, vc
    torch.cuda.empty_cache()

for dt in (torch.float8_e4m3fn, torch.float16):
    step(dt, 256, "random"); step(dt, 256, "in order"); step(dt, 1024, "random"); step(dt, 2048, "random"); step(dt, 0, "-")
