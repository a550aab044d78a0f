"""Packed (varlen) op against the dense op on the same equal-length problem, at batch x kv-heads that are small or odd: does the flat work
list leave the chip idle where the dense grid did before round 5's unit placement?   python tools/varlen_small_probe.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn as fa

def b2b(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

for (B, S, H, Hk, D) in ((1, 8192, 8, 1, 128), (2, 4096, 8, 1, 128), (1, 4096, 32, 8, 128), (1, 4096, 28, 4, 128), (4, 4096, 32, 1, 128), (3, 4096, 16, 4, 128),
                         (2, 4096, 14, 2, 64), (8, 4096, 16, 16, 128), (1, 4096, 8, 8, 128)):
    dt = torch.bfloat16
    q = torch.randn(B, S, H, D, device="cuda", dtype=dt, requires_grad=True)
    k = torch.randn(B, S, Hk, D, device="cuda", dtype=dt, requires_grad=True)
    v = torch.randn(B, S, Hk, D, device="cuda", dtype=dt, requires_grad=True)
    do = torch.randn_like(q)
    cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device="cuda")
    qv, kv, vv = (t.detach().reshape(B * S, -1, D).requires_grad_(True) for t in (q, k, v))
    dov = do.reshape(B * S, H, D)
    fl = 4.0 * B * H * D * S * (S + 1) / 2
    fd = lambda: fa.flash_attn_func(q, k, v, causal=True)
    fv = lambda: fa.flash_attn_varlen_func(qv, kv, vv, cu, cu, S, S, causal=True)
    with torch.no_grad():
        td, tv = b2b(fd), b2b(fv)
    tdb = b2b(lambda: torch.autograd.grad(fd(), (q, k, v), do), n=10)
    tvb = b2b(lambda: torch.autograd.grad(fv(), (qv, kv, vv), dov), n=10)
    print(f"B{B} S{S} H{H}/{Hk} D{D}: dense fwd {fl / td / 1e9:6.0f} TF fwd+bwd {3.5 * fl / tdb / 1e9:6.0f} TF | varlen fwd {fl / tv / 1e9:6.0f} TF fwd+bwd {3.5 * fl / tvb / 1e9:6.0f} TF", flush=True)
