"""Fresh-prompt prefill through the varlen op: paged K / V (pages of 16 and 256 tokens: compiler-scheduled forward) against packed K / V
(hand-scheduled forward).   python tools/paged_varlen_prefill.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn as fa
def t_ms(f, n=8):
    for _ in range(2): f()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)[n // 2]
Hq, Hk, D = 32, 8, 128
for lens in ([2048] * 8, [4096] * 4, [1000, 3000, 500, 2500, 4000, 700, 1500, 2684], [8192] * 2):
    for page in (16, 256):
        B = len(lens); T = sum(lens)
        pps = [(l + page - 1) // page for l in lens]
        nblk = sum(pps)
        kc = torch.randn(nblk, page, Hk, D, device="cuda", dtype=torch.bfloat16); vc = torch.randn_like(kc)
        bt = torch.zeros(B, max(pps), dtype=torch.int32)
        perm = iter(torch.randperm(nblk).tolist())
        for b in range(B):
            for j in range(pps[b]): bt[b, j] = next(perm)
        bt = bt.cuda()
        q = torch.randn(T, Hq, D, device="cuda", dtype=torch.bfloat16)
        cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
        su = torch.tensor(lens, dtype=torch.int32, device="cuda")
        fl = 4.0 * Hq * D * sum(l * (l + 1) / 2 for l in lens)
        ms = t_ms(lambda: fa.flash_attn_varlen_func(q, kc, vc, cu, cu, max(lens), max(lens), causal=True, block_table=bt, seqused_k=su))
        kp = torch.randn(T, Hk, D, device="cuda", dtype=torch.bfloat16); vp = torch.randn_like(kp)
        ms2 = t_ms(lambda: fa.flash_attn_varlen_func(q, kp, vp, cu, cu, max(lens), max(lens), causal=True))
        print(f"prefill {B} prompts ({T} tokens) page {page:3d}: paged {ms:.3f} ms {fl/ms/1e9:.0f} TF | packed (unpaged) {ms2:.3f} ms {fl/ms2/1e9:.0f} TF", flush=True)
