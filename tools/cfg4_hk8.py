"""BASELINE config 4 at H_k = 8 and 32 (fp8 and fp16 KV) through bench.py's own config4(): N timings in one process, for A/B of
library variants (FA_MI355_LIB).  python tools/cfg4_hk8.py [N]"""
import os, sys, importlib.util
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
import torch, flash_attn
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for Hk in (8, 32):
    for dt in (torch.float8_e4m3fn, torch.float16):
        r = [m.config4(flash_attn, dev, dt, Hk=Hk) for _ in range(N)]
        torch.cuda.empty_cache()
        print(f"lib={os.path.basename(os.environ.get('FA_MI355_LIB', 'product'))} Hk {Hk:2d} {'fp8 ' if dt == torch.float8_e4m3fn else 'fp16'}: " +
              "  ".join(f"{x['ms']:.4f} ms {x['achieved_gbs'] / 1e3:.2f} TB/s" for x in r), flush=True)
