"""Config-4 decode (fp8 KV) against an explicit num_splits (0 = the library's choice)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn
B, H, D, L, page = 128, 32, 128, 8192, 256
dev = "cuda"; torch.manual_seed(421)
pps = (L + 1 + page - 1) // page; nblk = B * pps
KV16 = os.environ.get("KV") == "fp16"           # KV=fp16: 16-bit cache
kc = torch.randn(nblk, page, H, D, device=dev, dtype=torch.float16) * 0.5
vc = torch.randn(nblk, page, H, D, device=dev, dtype=torch.float16) * 0.5
if not KV16:
    kc, vc = kc.to(torch.float8_e4m3fn), vc.to(torch.float8_e4m3fn)
bt = torch.randperm(nblk, device=dev).reshape(B, pps).to(torch.int32)
q = torch.randn(B, 1, H, D, device=dev, dtype=torch.float16)
kn = torch.randn(B, 1, H, D, device=dev, dtype=torch.float16); vn = torch.randn(B, 1, H, D, device=dev, dtype=torch.float16)
sl = torch.full((B,), L, dtype=torch.int32, device=dev)
ang = torch.arange(pps * page + 8, device=dev)[:, None] * (1.0 / 10000 ** (torch.arange(0, D, 2, device=dev) / D))[None]
cos, sin = torch.cos(ang).half(), torch.sin(ang).half()
for ns in [int(x) for x in sys.argv[1:]] or [0, 4, 6, 8, 12, 16, 24, 32]:
    fn = lambda: flash_attn.flash_attn_with_kvcache(q, kc, vc, k=kn, v=vn, rotary_cos=cos, rotary_sin=sin, cache_seqlens=sl,
                                                    block_table=bt, causal=True, rotary_interleaved=False, num_splits=ns,
                                                    **({} if KV16 else dict(k_descale=1.0, v_descale=1.0)))
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"num_splits {ns:2d}: {ms:.3f} ms  {2.0 * B * (L + 1) * H * D * kc.element_size() / ms / 1e6:.0f} GB/s", flush=True)
