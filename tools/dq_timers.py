"""dQ measurement build (generator cfg timers=1): per-pass s_memtime stamps read back from softmax_d rows r0..r0+3 of wave 0.
  build here:   VARIANT_KERNEL=dq python tools/asm_variants.py build tm:'--cfg={"timers":1}'
  run on GPU:   FA_MI355_LIB=tools/variants/libfa_tm.so python tools/dq_timers.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import ctypes, torch, flash_attn
from flash_attn_mi355 import flash_attn_interface as fi
torch.manual_seed(421)
B, S, H = 8, 4096, 16
for causal in (True, False):
    q = torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16)
    k, v, do = (torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    o, lse, _ = flash_attn.flash_attn_func(q, k, v, causal=causal, return_attn_probs=True)
    dq = torch.empty_like(q)
    for _ in range(2):
        sd = fi._dense_backward(do, q, k, v, o, lse, None, 0.0, 128 ** -0.5, causal, (-1, -1), 0.0, None, dq, None, None)
    torch.cuda.synchronize()
    t = sd.view(torch.int32).view(B, H, S // 256, 256)[..., 0:4].to(torch.int64) & 0xffffffff      # wave 0's stamps of every 256-row block
    pro = (t[..., 1] - t[..., 0]) & 0xffffffff
    loop = (t[..., 2] - t[..., 1]) & 0xffffffff
    epi = (t[..., 3] - t[..., 2]) & 0xffffffff
    nqb = S // 256
    print(f"causal={causal}: ticks (s_memtime) per pass, mean over batch x heads")
    for qb in (0, 1, 4, 8, 12, 15):
        stages = 8 * (qb + 1) if causal else S // 32
        lm = loop[..., qb].float().mean().item()
        print(f"  q block {qb:2d} ({stages:3d} stages): prologue {pro[..., qb].float().mean().item():8.0f}  loop {lm:9.0f} "
              f"({lm / (stages + 2):6.0f} / iteration)  epilogue {epi[..., qb].float().mean().item():7.0f}")
    tot = (t[..., 3] - t[..., 0]) & 0xffffffff
    print(f"  sums / 256 CUs: prologue {pro.sum().item() / 256:.0f}  loop {loop.sum().item() / 256:.0f}  "
          f"epilogue {epi.sum().item() / 256:.0f}  total {tot.sum().item() / 256:.0f} ticks", flush=True)
