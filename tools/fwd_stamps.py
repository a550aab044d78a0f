"""s_memtime sums of fa_fwd_kernel at BASELINE config 3 (variant build: define_variant.py stamps fa_fwd.hip -DFA_FWD_STAMPS [-DFA_EXP_WALK=1]).
  FA_MI355_LIB=$PWD/tools/variants/libfa_stamps.so python tools/fwd_stamps.py"""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn
from flash_attn_mi355 import _lib
B, H, D, W = 64, 32, 64, 512
g = torch.Generator().manual_seed(421)
lens = torch.randint(64, 2049, (B,), generator=g); lens[0] = 2048
cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = lens.cumsum(0); T = int(cu[-1]); cu = cu.cuda()
q, k, v = (torch.randn(T, H, D, device="cuda", dtype=torch.float16) for _ in range(3))
fwd = lambda: flash_attn.flash_attn_varlen_func(q, k, v, cu, cu, 2048, 2048, causal=True, window_size=(W, 0))
buf = (ctypes.c_ulonglong * 8)()
_lib.lib.fa_debug_read_fwd_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
with torch.no_grad():
    for _ in range(3): fwd()
    torch.cuda.synchronize(); _lib.lib.fa_debug_read_fwd_stamps(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fwd(); b.record(); torch.cuda.synchronize()
_lib.lib.fa_debug_read_fwd_stamps(buf, 0)
t = list(buf); n = max(t[5], 1)
names = ["entry->work item", "->geometry", "->first tile in LDS", "tile loop", "epilogue"]
print(f"launch {a.elapsed_time(b):.3f} ms, {t[5]} workgroups, {t[6] / n:.2f} tile steps each; s_memtime ticks per workgroup (wave 0):")
for i, nm in enumerate(names): print(f"  {nm:22s} {t[i] / n:9.0f}")
print(f"  {'total':22s} {sum(t[:5]) / n:9.0f}   loop per step {t[3] / max(t[6], 1):.0f}, of which at the barrier {t[7] / max(t[6], 1):.0f}")
