import os, sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo/flash-attention-v100_amd")
import torch, flash_attn
f8 = torch.float8_e4m3fn
def run(B, H, Hk, L, ns):
    D, page = 128, 256
    pps = (L + page) // page; nblk = B * pps
    kc = (torch.randn(nblk, page, Hk, D, device="cuda", dtype=torch.float16) * 0.5).to(f8); vc = (torch.randn(nblk, page, Hk, D, device="cuda", dtype=torch.float16) * 0.5).to(f8)
    bt = torch.randperm(nblk, device="cuda").reshape(B, pps).to(torch.int32)
    q = torch.randn(B, 1, H, D, device="cuda", dtype=torch.float16)
    lens = torch.full((B,), L, dtype=torch.int32, device="cuda")
    fn = lambda: flash_attn.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=lens, block_table=bt, causal=True, num_splits=ns, k_descale=1.0, v_descale=1.0)
    for _ in range(3): fn()
    ts = []
    for _ in range(15):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); ms = ts[len(ts) // 2]
    nb = 2.0 * B * L * Hk * D
    print(f"B{B} H{H}/{Hk} L{L} splits {ns}: {ms*1e3:.1f} us {nb/ms/1e6:.0f} GB/s", flush=True)
for B, H, Hk, L in ((128, 64, 8, 8192), (32, 64, 8, 8192), (8, 64, 8, 8192), (128, 32, 16, 8192), (1, 64, 8, 32768)):
    for ns in (0, 1, 2, 4, 8, 16, 32):
        run(B, H, Hk, L, ns)
