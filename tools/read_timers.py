"""Development aid: run the bench workload's backward with an FA_TIMERS variant library and print
the per-phase cycle sums of the instrumented wave.
  build:  python tools/define_variant.py tmr fa_bwd.hip -DFA_TIMERS=100 [-DFA_BWD_ASM_OFF ...]
  run:    FA_BWD_ASM=0 FA_MI355_LIB=tools/variants/libfa_tmr.so python tools/read_timers.py [causal|noncausal] [D] [H]"""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn
from flash_attn_mi355 import _lib
torch.manual_seed(1)
causal = (sys.argv[1] != "noncausal") if len(sys.argv) > 1 else True
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2048 // D
B, S = 8, 4096
q, k, v, do = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
for _ in range(2):
    o = flash_attn.flash_attn_func(q, k, v, causal=causal)
    o.backward(do)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
_lib.lib.fa_debug_read_timers.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", _lib.lib.fa_debug_read_timers(buf, 16))
names = ["sd", "sm", "bk", "load_issue", "store", "barrier", "-", "n_steps"]
for ps in range(2):
    t = list(buf[ps * 8: ps * 8 + 8])
    n = max(t[7], 1)
    print(f"pass {ps}: steps {t[7]}  " + "  ".join(f"{names[i]} {t[i]/n:.0f}" for i in range(6)) + f"  total/step {sum(t[:6])/n:.0f}")
