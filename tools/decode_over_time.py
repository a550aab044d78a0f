"""BASELINE config 4 decode over time: does the achieved bandwidth of the token-major kernel depend on how long the socket has been streaming?
  python tools/decode_over_time.py      (chunks of 20 calls between events, ~1.2 s per cache type; then sizes)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn
f8 = torch.float8_e4m3fn


def make(B, kvd, H=32, Hk=32, D=128, L=8192, page=256):
    dt = torch.float16
    pps = (L + 1 + page - 1) // page
    nblk = B * pps
    if kvd == f8:
        kc = (torch.randn(nblk, page, Hk, D, device="cuda", dtype=dt) * 0.5).to(f8); vc = (torch.randn(nblk, page, Hk, D, device="cuda", dtype=dt) * 0.5).to(f8)
        kw = dict(k_descale=1.0, v_descale=1.0)
    else:
        kc = torch.randn(nblk, page, Hk, D, device="cuda", dtype=dt); vc = torch.randn(nblk, page, Hk, D, device="cuda", dtype=dt); kw = {}
    bt = torch.randperm(nblk, device="cuda").reshape(B, pps).to(torch.int32)
    q, kn, vn = (torch.randn(B, 1, h, D, device="cuda", dtype=dt) for h in (H, Hk, Hk))
    lens = torch.full((B,), L, dtype=torch.int32, device="cuda")
    ang = torch.arange(pps * page + 8, device="cuda")[:, None] * (1.0 / 10000 ** (torch.arange(0, D, 2, device="cuda") / D))[None]
    cos, sin = torch.cos(ang).to(dt), torch.sin(ang).to(dt)
    fn = lambda: flash_attn.flash_attn_with_kvcache(q, kc, vc, k=kn, v=vn, rotary_cos=cos, rotary_sin=sin, cache_seqlens=lens,
                                                    block_table=bt, causal=True, rotary_interleaved=False, **kw)
    return fn, 2.0 * B * (L + 1) * Hk * D * kc.element_size()


for name, kvd in (("fp16", None), ("fp8", f8)):
    fn, nbytes = make(128, kvd)
    fn(); torch.cuda.synchronize()
    import time; time.sleep(0.5)                          # start from an idle socket
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(25)]
    ev[0].record()
    for c in range(24):
        for _ in range(20):
            fn()
        ev[c + 1].record()
    torch.cuda.synchronize()
    print(f"{name} KV, B 128: TB/s per chunk of 20 calls from an idle socket: " +
          " ".join(f"{nbytes / (ev[c].elapsed_time(ev[c + 1]) / 20) / 1e9:.2f}" for c in range(24)), flush=True)
    del fn
    torch.cuda.empty_cache()
