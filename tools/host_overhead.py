"""Host cost of one call: eager wall time per call (launch-bound loop, one synchronisation at the end) against the same
step replayed from a HIP graph and against the device time of the kernels (events), for a decode step and a small
training step.  python tools/host_overhead.py   (on the GPU box)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn as fa


def measure(name, step, n=300):
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t_issue = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / n
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / n
    print(f"{name:46s} host issue {t_issue*1e6:7.1f} us/call | eager {t_eager*1e6:7.1f} us | graph replay {t_graph*1e6:7.1f} us", flush=True)


for B in (8, 1, 1, 64):
    Hq, Hk, D, ctx, page = 32, 8, 128, 4096, 256
    nblk = B * ctx // page
    kc = torch.randn(nblk, page, Hk, D, device="cuda", dtype=torch.float16)
    vc = torch.randn_like(kc)
    bt = torch.randperm(nblk, device="cuda").to(torch.int32).reshape(B, ctx // page)
    lens = torch.full((B,), ctx - 64, dtype=torch.int32, device="cuda")
    q = torch.randn(B, 1, Hq, D, device="cuda", dtype=torch.float16)
    kn = torch.randn(B, 1, Hk, D, device="cuda", dtype=torch.float16); vn = torch.randn_like(kn)
    pos = torch.arange(ctx + 8, dtype=torch.float32)[:, None] / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))[None, :]
    cos, sin = torch.cos(pos).half().cuda(), torch.sin(pos).half().cuda()
    measure(f"decode B{B} Hq32 Hk8 D128 ctx4k paged + append + rope",
            lambda: fa.flash_attn_with_kvcache(q, kc, vc, k=kn, v=vn, rotary_cos=cos, rotary_sin=sin, cache_seqlens=lens,
                                               block_table=bt, causal=True, rotary_interleaved=False))
    measure(f"decode B{B} (no append)",
            lambda: fa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, block_table=bt, causal=True))
q = torch.randn(1, 512, 16, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
k = torch.randn(1, 512, 16, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
v = torch.randn(1, 512, 16, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
do = torch.randn_like(q)
with torch.no_grad():
    measure("forward B1 S512 H16 D128", lambda: fa.flash_attn_func(q, k, v, causal=True))
measure("forward + backward B1 S512 H16 D128", lambda: torch.autograd.grad(fa.flash_attn_func(q, k, v, causal=True), (q, k, v), do))
