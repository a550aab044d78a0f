"""config 4 at H_k = 8, fp8 cache: the step time only (timing knock-outs of library variants; results are NOT checked)."""
import os, sys, importlib.util
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
import torch, flash_attn
r = [m.config4(flash_attn, torch.device("cuda", 0), torch.float8_e4m3fn, Hk=8) for _ in range(3)]
print(f"lib={os.path.basename(os.environ.get('FA_MI355_LIB', 'product'))}: " + "  ".join(f"{x['ms']:.4f} ms {x['achieved_gbs'] / 1e3:.2f} TB/s" for x in r), flush=True)
