"""Forward asm measurement build (generator cfg timers=1): per-pass s_memtime stamps read back from the LSE rows.
  build here:   VARIANT_KERNEL=asm python tools/asm_variants.py build ftm:'--cfg={"timers":1}'
  run on GPU:   FA_MI355_LIB=tools/variants/libfa_ftm.so python tools/fwd_timers.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn
torch.manual_seed(421)
SHAPES = [tuple(int(x) for x in a.split(",")) + (True,) for a in sys.argv[1:]] or [(8, 4096, 16, True), (8, 4096, 16, False), (4, 8192, 16, True)]
for (B, S, H, causal) in SHAPES:
    q, k, v = (torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    for _ in range(2):
        o, lse, _ = flash_attn.flash_attn_func(q, k, v, causal=causal, return_attn_probs=True)
    torch.cuda.synchronize()
    t = lse.view(torch.int32).view(B, H, S // 256, 256)[..., 0:6].to(torch.int64) & 0xffffffff     # wave 0 of every 256-row pass
    pro = (t[..., 1] - t[..., 0]) & 0xffffffff
    loop = (t[..., 2] - t[..., 1]) & 0xffffffff
    epi = (t[..., 3] - t[..., 2]) & 0xffffffff
    gen, fast = t[..., 4], t[..., 5]
    print(f"B{B} S{S} causal={causal}: ticks per 256-row pass (mean over batch x heads)")
    nqb = S // 256
    for qb in sorted(set((0, 1, nqb // 4, nqb // 2, nqb - 2, nqb - 1))):
        lm, g_, f_ = loop[:, :, qb].float().mean().item(), gen[:, :, qb].float().mean().item(), fast[:, :, qb].float().mean().item()
        print(f"  q block {qb:2d}: prologue {pro[:, :, qb].float().mean().item():7.0f}  loop {lm:9.0f} = {g_:.0f} generic + {f_:.0f} fast iterations "
              f"({lm / max(g_ + f_, 1):6.0f} / iteration)  epilogue {epi[:, :, qb].float().mean().item():7.0f}")
    if causal and nqb >= 2:      # mirrored pairs: gap between the end of the first pass (block i) and the start of the second (nqb-1-i)
        gap = (t[:, :, nqb - 1, 0] - t[:, :, 0, 3]) & 0xffffffff
        print(f"  gap between the two passes of a workgroup (end of block 0 -> start of block {nqb - 1}): {gap.float().mean().item():.0f} ticks")
    tot = (t[..., 3] - t[..., 0]) & 0xffffffff
    span = (t[..., 3].max() - t[..., 0].min()).item()
    print(f"  first stamp to last stamp (whole kernel, one clock domain assumed): {span} ticks; passes per CU {B * H * nqb / 256:.1f}")
    print(f"  sums / 256 CUs: prologue {pro.sum().item() / 256:.0f}  loop {loop.sum().item() / 256:.0f}  epilogue {epi.sum().item() / 256:.0f}  "
          f"total {tot.sum().item() / 256:.0f} ticks; generic iterations {gen.sum().item() / 256:.0f}, fast {fast.sum().item() / 256:.0f} per CU", flush=True)
