"""dK/dV measurement build (generator cfg timers=1): per-pass s_memtime stamps read back from the dK rows.
  build here:   VARIANT_KERNEL=bwd python tools/asm_variants.py build tm:'--cfg={"timers":1}'
  run on GPU:   FA_MI355_LIB=tools/variants/libfa_tm.so python tools/bwd_timers.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn
torch.manual_seed(421)
B, S, H = 8, 4096, 16
for causal in (True, False):
    q, k, v = (torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
    do = torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16)
    o = flash_attn.flash_attn_func(q, k, v, causal=causal)
    for _ in range(2):
        dq, dk, dv = torch.autograd.grad(o, (q, k, v), do, retain_graph=True)
    torch.cuda.synchronize()
    t = dk.view(torch.int32).view(B, S, H, 64)[:, :, :, 0:16:4].to(torch.int64) & 0xffffffff      # [B, S, H, 4] stamps
    tb = t[:, 0::128]                                          # first key of every key block (wave 0, lane 0): [B, nkb, H, 4]
    pro = (tb[..., 1] - tb[..., 0]) & 0xffffffff
    loop = (tb[..., 2] - tb[..., 1]) & 0xffffffff
    epi = (tb[..., 3] - tb[..., 2]) & 0xffffffff
    nkb = S // 128
    print(f"causal={causal}: ticks (s_memtime) per pass, mean over batch x heads")
    for kb in (0, 1, 8, 15, 16, 24, 30, 31):
        stages = (S // 32 - 4 * kb) if causal else S // 32
        lm = loop[:, kb].float().mean().item()
        print(f"  key block {kb:2d} ({stages:3d} stages): prologue {pro[:, kb].float().mean().item():8.0f}  loop {lm:9.0f} "
              f"({lm / (stages + 2):6.0f} / iteration)  epilogue {epi[:, kb].float().mean().item():7.0f}")
    if causal:   # the second pass of a workgroup starts after the first one's end (blocks kb and nkb-1-kb)
        gap = (tb[:, nkb - 1, :, 0] - tb[:, 0, :, 3]) & 0xffffffff
        print(f"  gap between the two passes of a workgroup (block 0 -> block {nkb - 1}): {gap.float().mean().item():.0f}")
    tot = (tb[..., 3] - tb[..., 0]) & 0xffffffff
    print(f"  sums / 256 CUs: prologue {pro.sum().item() / 256:.0f}  loop {loop.sum().item() / 256:.0f}  "
          f"epilogue {epi.sum().item() / 256:.0f}  total {tot.sum().item() / 256:.0f} ticks", flush=True)
