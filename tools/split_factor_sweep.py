"""dK/dV(+preprocess) and full backward for one library variant (FA_MI355_LIB=...libfa_mi355_s<N>.so: FA_DKV_SPLIT_FORCE=N) over dense
shapes around the sizes where dkv_split_factor has to decide.   python tools/split_factor_sweep.py [nosplit]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, flash_attn as fa
from flash_attn_mi355 import flash_attn_interface as fi
from _bwdsel import bwd_call
NOSPLIT = len(sys.argv) > 1 and sys.argv[1] == "nosplit"     # deterministic=True: FA_FLAG_NO_DKV_SPLIT

def b2b(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

tag = os.path.basename(os.environ.get("FA_MI355_LIB", "product")) + (" nosplit" if NOSPLIT else "")
SH = [(1, 4096, 32, 8, 128, True), (3, 4096, 8, 8, 128, True), (3, 4096, 32, 8, 128, True), (5, 4096, 32, 8, 128, True), (1, 2048, 32, 8, 128, True),
      (6, 2048, 8, 8, 128, True), (2, 8192, 16, 16, 128, False), (12, 1024, 16, 16, 128, False), (1, 4096, 16, 8, 256, True), (3, 4096, 16, 4, 64, True),
      (12, 2048, 16, 16, 64, True), (4, 1024, 16, 16, 64, False)]
out = []
for (B, S, H, Hk, D, causal) in SH:
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    do = torch.randn_like(q)
    f = lambda a, b, c: fa.flash_attn_func(a, b, c, causal=causal, deterministic=NOSPLIT)
    out.append(f"{b2b(bwd_call(f, q, k, v, do, 'all')):7.1f}")
print(f"{tag:28s} " + " ".join(out), flush=True)
