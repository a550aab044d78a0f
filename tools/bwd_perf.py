"""Per-kernel backward timing at config 2 (bf16 causal B8 H16 S4096 D128) (which kernels run follows from the gradients asked for)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _bwdsel import bwd_call
torch.manual_seed(421)
for (B, S, H, Hk, causal) in ((8, 4096, 16, 16, True), (8, 4096, 16, 16, False), (4, 4096, 32, 8, True)):
    q = torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(B, S, Hk, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(B, S, Hk, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    do = torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16)
    o = flash_attn.flash_attn_func(q, k, v, causal=causal)
    res = {}
    for name in ("dkdv", "dq", "all"):
        fn = bwd_call(lambda a, b, c: flash_attn.flash_attn_func(a, b, c, causal=causal), q, k, v, do, name)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record(); torch.cuda.synchronize()
        res[name] = s.elapsed_time(e) / 10
    res["dkdv_kernel"] = res["all"] - res["dq"]
    fl = 4.0 * B * H * S * S * 128 * (0.5 if causal else 1.0)
    print(f"B{B} S{S} H{H}/{Hk} causal={causal}: " + "  ".join(f"{k_} {v_:.3f} ms" for k_, v_ in res.items()) +
          f"  | dkdv {2 * fl / res['dkdv'] / 1e9:.0f} TF  bwd {2.5 * fl / res['all'] / 1e9:.0f} TF", flush=True)
