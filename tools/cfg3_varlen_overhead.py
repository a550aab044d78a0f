"""What does the PACKED form cost config 3's kernels?  The same D 64, H 32, window (512, 0) problem with 64 equal sequences of
1054 tokens (config 3's mean) through the varlen op (flat work list: every workgroup finds its sequence with two dependent loads
of cu_seqlens before it can fetch anything) and through the dense op (geometry from the launch arguments), forward and backward."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn
def b2b(fn, n=40):
    for _ in range(8): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n
H, D, W = 32, 64, 512
for L in (1054, 1024, 2048, 512):
    B = 67456 // L
    T = B * L
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device="cuda")
    q, k, v, do = (torch.randn(T, H, D, device="cuda", dtype=torch.float16) for _ in range(4))
    qd, kd, vd, dod = (x.view(B, L, H, D) for x in (q, k, v, do))
    with torch.no_grad():
        tv = b2b(lambda: flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, L, L, window_size=(W, 0)))
        td = b2b(lambda: flash_attn.flash_attn_func(qd, kd, vd, window_size=(W, 0)))
    qv, kv, vv = (x.clone().requires_grad_(True) for x in (q, k, v))
    ov = flash_attn.flash_attn_varlen_func(qv, kv, vv, cu, cu, L, L, window_size=(W, 0))
    tvb = b2b(lambda: torch.autograd.grad(ov, (qv, kv, vv), do, retain_graph=True), 20)
    q2, k2, v2 = (x.clone().requires_grad_(True) for x in (qd, kd, vd))
    od = flash_attn.flash_attn_func(q2, k2, v2, window_size=(W, 0))
    tdb = b2b(lambda: torch.autograd.grad(od, (q2, k2, v2), dod, retain_graph=True), 20)
    print(f"{B} x {L} tokens: forward packed {tv:.4f} ms dense {td:.4f} ms ({tv / td:.3f} x) | backward packed {tvb:.4f} ms dense {tdb:.4f} ms ({tvb / tdb:.3f} x)", flush=True)
