"""Where the host time of a call goes: cProfile over a small forward, a small forward + backward and a decode step (the kernels are
short, the loop is host-bound).  python tools/host_profile.py [n]"""
import os, sys, cProfile, pstats, io, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn as fa
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
q, k, v = (torch.randn(1, 512, 16, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
do = torch.randn_like(q)
B, Hq, Hk, D, ctx, page = 8, 32, 8, 128, 4096, 256
nblk = B * ctx // page
kc = torch.randn(nblk, page, Hk, D, device="cuda", dtype=torch.float16); vc = torch.randn_like(kc)
bt = torch.randperm(nblk, device="cuda").to(torch.int32).reshape(B, ctx // page)
lens = torch.full((B,), ctx - 64, dtype=torch.int32, device="cuda")
qd = torch.randn(B, 1, Hq, D, device="cuda", dtype=torch.float16)

def fwd():
    with torch.no_grad():
        fa.flash_attn_func(q, k, v, causal=True)
def fwdbwd():
    torch.autograd.grad(fa.flash_attn_func(q, k, v, causal=True), (q, k, v), do)
def dec():
    fa.flash_attn_with_kvcache(qd, kc, vc, cache_seqlens=lens, block_table=bt, causal=True)
for name, f in (("forward", fwd), ("forward + backward", fwdbwd), ("decode (no append)", dec)):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        f()
    t = (time.perf_counter() - t0) / N
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(N):
        f()
    pr.disable(); torch.cuda.synchronize()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
    print(f"===== {name}: {t * 1e6:.1f} us per call (unprofiled issue loop)")
    for ln in s.getvalue().splitlines()[6:34]:
        print(ln[:170])
