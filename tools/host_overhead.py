"""Host-side cost per call of the Python operator layer (tiny GPU work, many calls)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn
torch.manual_seed(0)
dev = "cuda"
q = torch.randn(1, 1, 32, 128, device=dev, dtype=torch.float16)
kc = torch.randn(1, 256, 8, 128, device=dev, dtype=torch.float16); vc = torch.randn_like(kc)
kn = torch.randn(1, 1, 8, 128, device=dev, dtype=torch.float16); vn = torch.randn_like(kn)
sl = torch.tensor([100], dtype=torch.int32, device=dev)
qd = torch.randn(1, 128, 4, 128, device=dev, dtype=torch.float16)
def bench(name, fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); el = time.perf_counter() - t
    print(f"{name:40s} {el / n * 1e6:8.1f} us/call", flush=True)
bench("kvcache decode (append, no rotary)", lambda: flash_attn.flash_attn_with_kvcache(q, kc, vc, k=kn, v=vn, cache_seqlens=sl, causal=True))
bench("kvcache decode (no append)", lambda: flash_attn.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=sl, causal=True))
with torch.no_grad():
    bench("dense fwd tiny (no grad)", lambda: flash_attn.flash_attn_func(qd, qd, qd, causal=True))
bench("torch.empty_like baseline", lambda: torch.empty_like(q))
bench("torch sdpa tiny", lambda: torch.nn.functional.scaled_dot_product_attention(qd.transpose(1, 2), qd.transpose(1, 2), qd.transpose(1, 2), is_causal=True))
