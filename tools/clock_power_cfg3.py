"""Socket power and shader clock under BASELINE config 3's forward / backward and under the fp8 GQA decode step
(the question behind profiles/r04_config3_forward.txt: is a kernel with VALU 53 % / matrix pipe 28 % busy at the power limit too?).
  python tools/clock_power_cfg3.py [seconds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from clock_power import Smi, loop, flash_attn
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
smi = Smi()
print("SMI source:", smi.kind, getattr(smi, "err", ""))
print("idle reading:", smi.read() if smi.kind else None)
B, H, D, W = 64, 32, 64, 512
g = torch.Generator().manual_seed(421)
lens = torch.randint(64, 2049, (B,), generator=g); lens[0] = 2048
cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = lens.cumsum(0); T = int(cu[-1]); cu = cu.cuda()
q, k, v, do = (torch.randn(T, H, D, device="cuda", dtype=torch.float16) for _ in range(4))
def pairs(L): return L * (L + 1) // 2 if L <= W + 1 else (W + 1) * (W + 2) // 2 + (L - W - 1) * (W + 1)
fl = 4.0 * D * H * sum(pairs(int(L)) for L in lens)
fwd = lambda: flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, 2048, 2048, causal=True, window_size=(W, 0))
with torch.no_grad():
    loop("config 3 fwd (fa_fwd_kernel<fp16, 64>)", fwd, seconds, smi, fl)
    z = torch.zeros_like(q)
    loop("config 3 fwd, ZERO inputs", lambda: flash_attn.flash_attn_varlen_func(z, z, z, cu, cu, 2048, 2048, causal=True, window_size=(W, 0)), seconds, smi, fl)
q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
o = fwd()
loop("config 3 bwd (dQ + dK/dV kernels)", lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True), seconds, smi, 2.5 * fl)
# fp8 GQA decode (H 64/8, B 128, 8192 keys): "TFLOP/s" column = TB/s of the cache stream x 1000
f8 = torch.float8_e4m3fn
Bd, Hq, Hk, Dd, L, page = 128, 64, 8, 128, 8192, 256
pps = L // page; nblk = Bd * pps
kc = (torch.randn(nblk, page, Hk, Dd, device="cuda", dtype=torch.float16) * 0.5).to(f8); vc = (torch.randn(nblk, page, Hk, Dd, device="cuda", dtype=torch.float16) * 0.5).to(f8)
bt = torch.randperm(nblk, device="cuda").reshape(Bd, pps).to(torch.int32)
qd = torch.randn(Bd, 1, Hq, Dd, device="cuda", dtype=torch.float16)
ln = torch.full((Bd,), L, dtype=torch.int32, device="cuda")
with torch.no_grad():
    loop("fp8 GQA decode H 64/8 (GB/s in the TFLOP/s column)", lambda: flash_attn.flash_attn_with_kvcache(qd, kc, vc, cache_seqlens=ln, block_table=bt, causal=True, k_descale=1.0, v_descale=1.0),
         seconds, smi, 2.0 * Bd * L * Hk * Dd * 1e3)
    kc16 = torch.randn(nblk, page, 32, Dd, device="cuda", dtype=torch.float16); vc16 = torch.randn_like(kc16)
    q32 = torch.randn(Bd, 1, 32, Dd, device="cuda", dtype=torch.float16)
    loop("fp16 decode H 32/32 (token-major kernel; GB/s)", lambda: flash_attn.flash_attn_with_kvcache(q32, kc16, vc16, cache_seqlens=ln, block_table=bt, causal=True),
         seconds, smi, 2.0 * Bd * L * 32 * Dd * 2 * 1e3)
