"""Host and device cost of the split dK/dV form on a tiny shape: python tools/split_host_probe.py"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, flash_attn as fa
from flash_attn_mi355 import flash_attn_interface as fi
from _bwdsel import bwd_call

def host_us(f, n=300):
    for _ in range(20): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    t = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return t * 1e6

def dev_us(f, n=50):
    for _ in range(10): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for mb in (1, 8, 20, 64, 256):
    print(f"torch.empty({mb} MiB) + free: {host_us(lambda: torch.empty(mb << 20, dtype=torch.uint8, device='cuda')):.1f} us", flush=True)
for (B, Sq, Sk, H, Hk, D, causal, dt) in ((8, 4096, 77, 8, 8, 40, False, torch.float16), (4, 512, 128, 12, 12, 64, False, torch.float16),
                                           (1, 2048, 2048, 32, 8, 128, True, torch.bfloat16)):
    q = torch.randn(B, Sq, H, D, device="cuda", dtype=dt, requires_grad=True)
    k = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt, requires_grad=True)
    v = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt, requires_grad=True)
    do = torch.randn_like(q)
    for on in (True, False):
        f = lambda a, b, c, on=on: fa.flash_attn_func(a, b, c, causal=causal, deterministic=not on)     # deterministic=True: FA_FLAG_NO_DKV_SPLIT
        for nm in ("dkdv", "dq", "all"):
            c = bwd_call(f, q, k, v, do, nm)
            print(f"B{B} Sq{Sq} Sk{Sk} H{H}/{Hk} D{D} split={on} {nm:5s}: issue loop {host_us(c):7.1f} us/call, back-to-back {dev_us(c):7.1f} us/call", flush=True)
