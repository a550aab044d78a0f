"""dK/dV kernel time against sequence length at constant B*S (constant number of key-block passes): the intercept of
T(S) is the per-pass overhead, the slope the per-stage cost."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn
torch.manual_seed(421)
H = 16
for causal in (True, False):
    for (B, S) in ((32, 1024), (16, 2048), (8, 4096), (4, 8192), (2, 16384)):
        q, k, v = (torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
        do = torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16)
        o = flash_attn.flash_attn_func(q.detach(), k, v, causal=causal)      # dk, dv only: preprocess + dK/dV kernel
        if True:                                # dK/dV alone (+ the preprocess kernel): only dk, dv are asked for
            for _ in range(3):
                torch.autograd.grad(o, (k, v), do, retain_graph=True)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                torch.autograd.grad(o, (k, v), do, retain_graph=True)
            e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        passes = B * H * S // 128
        stages = B * H * (S // 128) * (S // 32) * (0.5 if causal else 1.0) + (B * H * S // 128 * 2 if causal else 0)
        print(f"causal={causal} B{B} S{S}: {ms:.3f} ms  passes/CU {passes / 256:.0f}  stages/CU {stages / 256:.0f}  "
              f"-> {ms * 1e3 / (stages / 256):.3f} us/stage", flush=True)
