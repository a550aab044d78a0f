"""Paged vs contiguous K/V in the varlen forward (prefill against a paged cache)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, flash_attn
from bench_configs import timeit
torch.manual_seed(0)
B, S, H, Hk, D, page = 16, 4096, 16, 16, 128, 256
q = torch.randn(B * S, H, D, device="cuda", dtype=torch.bfloat16)
k = torch.randn(B * S, Hk, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k)
cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device="cuda")
pps = S // page
kp = k.view(B * pps, page, Hk, D); vp = v.view(B * pps, page, Hk, D)
perm = torch.randperm(B * pps, device="cuda")
kp2 = torch.empty_like(kp); vp2 = torch.empty_like(vp); kp2[perm] = kp; vp2[perm] = vp
bt = perm.view(B, pps).to(torch.int32)
fl = 4.0 * B * H * S * S * D / 2
with torch.no_grad():
    t0 = timeit(lambda: flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, S, S, causal=True))
    t1 = timeit(lambda: flash_attn.flash_attn_varlen_func(q, kp2, vp2, cu, cu, S, S, causal=True, block_table=bt))
    a = flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, S, S, causal=True)
    b = flash_attn.flash_attn_varlen_func(q, kp2, vp2, cu, cu, S, S, causal=True, block_table=bt)
print(f"contiguous {t0:.3f} ms ({fl/t0/1e9:.0f} TF) | paged {t1:.3f} ms ({fl/t1/1e9:.0f} TF) | max diff {(a.float()-b.float()).abs().max().item():.2e}")
