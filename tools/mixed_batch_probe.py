"""A mixed serving batch through ONE varlen call (N decode sequences with one query token + one prefill chunk, paged K / V), against
the two uniform calls it could be split into.   python tools/mixed_batch_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import torch, flash_attn as fa
def t_us(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in evs:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in evs)[n // 2] * 1e3
Hq, Hk, D, page, ctx = 32, 8, 128, 256, 8192
for (ndec, chunks) in ((32, [512]), (64, [2048]), (64, [512] * 4), (8, [512]), (128, []), (0, [2048])):
    B = ndec + len(chunks)
    qlens = [1] * ndec + chunks
    nblk = B * ctx // page
    kc = torch.randn(nblk, page, Hk, D, device="cuda", dtype=torch.bfloat16); vc = torch.randn_like(kc)
    bt = torch.randperm(nblk, device="cuda").to(torch.int32).reshape(B, ctx // page)
    lens = torch.full((B,), ctx - 64, dtype=torch.int32, device="cuda")
    q = torch.randn(sum(qlens), Hq, D, device="cuda", dtype=torch.bfloat16)
    cu_q = torch.tensor([0] + list(torch.tensor(qlens).cumsum(0)), dtype=torch.int32, device="cuda")
    cu_k = torch.cat([torch.zeros(1, dtype=torch.int32, device="cuda"), lens.cumsum(0).to(torch.int32)])
    mixed = t_us(lambda: fa.flash_attn_varlen_func(q, kc, vc, cu_q, cu_k, max(qlens), ctx, causal=True, block_table=bt, seqused_k=lens))
    parts = 0.0
    if ndec:
        qd = q[:ndec].reshape(ndec, 1, Hq, D)
        parts += t_us(lambda: fa.flash_attn_with_kvcache(qd, kc, vc, cache_seqlens=lens[:ndec], block_table=bt[:ndec], causal=True))
    if chunks:
        qp = q[ndec:].reshape(len(chunks), chunks[0], Hq, D)
        parts += t_us(lambda: fa.flash_attn_with_kvcache(qp, kc, vc, cache_seqlens=lens[ndec:], block_table=bt[ndec:], causal=True))
    print(f"{ndec:3d} decode seqs + {len(chunks)} prefill chunk(s) of {chunks[0] if chunks else 0:4d} (ctx {ctx}, H {Hq}/{Hk}): one varlen call {mixed:8.1f} us | decode call + prefill call {parts:8.1f} us", flush=True)
