"""Is config 4 at H_k = 8 slower per byte because of its kernel or because of its SIZE?  The same step (paged 256-token pages,
append + rotary, 8192-token caches, fp8 / fp16) at equal cache bytes through the token-major kernel (H 32/32, smaller batch) and the
MFMA kernels (H 32/8), plus larger batches of the GQA shape.  Medians of 15 evented calls."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn
dev = torch.device("cuda", 0)

def step(B, H, Hk, kv_dtype, L=8192, page=256, D=128, n=15):
    dt = torch.float16
    pps = (L + 1 + page - 1) // page
    nblk = B * pps
    kc = (torch.randn(nblk, page, Hk, D, device=dev, dtype=dt) * 0.5).to(kv_dtype)
    vc = (torch.randn(nblk, page, Hk, D, device=dev, dtype=dt) * 0.5).to(kv_dtype)
    kw = dict(k_descale=1.0, v_descale=1.0) if kv_dtype == torch.float8_e4m3fn else {}
    bt = torch.randperm(nblk, device=dev).reshape(B, pps).to(torch.int32)
    q = torch.randn(B, 1, H, D, device=dev, dtype=dt); kn = torch.randn(B, 1, Hk, D, device=dev, dtype=dt); vn = torch.randn_like(kn)
    seqlens = torch.full((B,), L, dtype=torch.int32, device=dev)
    ang = torch.arange(pps * page + 8, device=dev)[:, None] * (1.0 / 10000 ** (torch.arange(0, D, 2, device=dev) / D))[None]
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
    print(f"B {B:4d} H {H}/{Hk:<2d} {'fp8 ' if kc.element_size() == 1 else 'fp16'} L {L}: {nbytes / 1e9:6.2f} GB  {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:6.2f} TB/s", flush=True)
    del kc, vc
    torch.cuda.empty_cache()

f8, f16 = torch.float8_e4m3fn, torch.float16
for dt in (f8, f16):
    step(128, 32, 32, dt)      # config 4
    step(32, 32, 32, dt)       # the token-major kernel at config-4-H_k-8 bytes
    step(64, 32, 32, dt)
    step(128, 32, 8, dt)       # config 4, H_k 8
    step(256, 32, 8, dt)       # ... at twice / four times the batch
    step(512, 32, 8, dt)
