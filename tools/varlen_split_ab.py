"""Packed backward, split forms on / off (FA_FLAG_NO_DKV_SPLIT), on mixed-length batches around the size where the split engages.
  python tools/varlen_split_ab.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, flash_attn as fa
from flash_attn_mi355 import flash_attn_interface as fi
from _bwdsel import bwd_call

def b2b(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

CASES = [
    ("4 seqs 2048+1024+512+512, H32/8 D128", [2048, 1024, 512, 512], 32, 8, 128, (-1, -1)),
    ("3 seqs 4096+3000+1096, H32/8 D128", [4096, 3000, 1096], 32, 8, 128, (-1, -1)),
    ("2 seqs 3000+1000, H32/8 D128", [3000, 1000], 32, 8, 128, (-1, -1)),
    ("8 seqs 256..2048, H16/16 D64 window 512", [256, 2048, 700, 1300, 512, 1800, 900, 1024], 16, 16, 64, (512, 0)),
    ("8 seqs 256..2048, H32/4 D128", [256, 2048, 700, 1300, 512, 1800, 900, 1024], 32, 4, 128, (-1, -1)),
    ("16 seqs of 512, H16/2 D64", [512] * 16, 16, 2, 64, (-1, -1)),
    ("6 seqs, H8/1 D128", [4096, 100, 2000, 1500, 300, 196], 8, 1, 128, (-1, -1)),
    ("32 seqs 128..1024, H32/8 D128", [128 + 28 * i for i in range(32)], 32, 8, 128, (-1, -1)),
]
for name, lens, H, Hk, D, win in CASES:
    T = sum(lens)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    q = torch.randn(T, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(T, Hk, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(T, Hk, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    do = torch.randn_like(q)
    r = {}
    for on in (False, True, False, True):
        f = lambda a, b, c, on=on: fa.flash_attn_varlen_func(a, b, c, cu, cu, max(lens), max(lens), causal=True, window_size=win, deterministic=not on)
        r.setdefault(on, []).append((b2b(bwd_call(f, q, k, v, do, "dkdv")), b2b(bwd_call(f, q, k, v, do, "all"))))
    fmt = lambda xs: " / ".join(f"{a:6.1f}" for a in xs)
    print(f"{name:42s} key blocks x kv-heads {(T // 128 + len(lens)) * Hk:5d} | dK/dV(+pre) off {fmt([x[0] for x in r[False]])}  on {fmt([x[0] for x in r[True]])} us"
          f" | backward off {fmt([x[1] for x in r[False]])}  on {fmt([x[1] for x in r[True]])} us", flush=True)
