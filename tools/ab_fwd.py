"""A/B of two libraries on the forward (bf16 causal H16 D128): python tools/ab_fwd.py tools/variants/libfa_X.so  - three rounds,
medians of 40 evented launches per shape, default library vs the variant, alternating."""
import os, sys, json, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
def one():
    import torch, flash_attn
    torch.manual_seed(1)
    res = {}
    for (tag, B, S, H) in (("c512", 64, 512, 16), ("c1k", 32, 1024, 16), ("c2k", 16, 2048, 16), ("c4k", 8, 4096, 16), ("c8k", 4, 8192, 16)):
        q, k, v = (torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
        with torch.no_grad():
            for _ in range(20): flash_attn.flash_attn_func(q, k, v, causal=True)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
            torch.cuda.synchronize()
            for s, e in evs:
                s.record(); flash_attn.flash_attn_func(q, k, v, causal=True); e.record()
            torch.cuda.synchronize()
        res[tag] = round(sorted(s.elapsed_time(e) for s, e in evs)[20], 4)
    print(json.dumps(res))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    one()
else:
    for rnd in range(3):
        for name, env in (("default", {}), ("variant", {"FA_MI355_LIB": os.path.abspath(sys.argv[1])})):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, **env), capture_output=True, text=True)
            print(f"{name:8s}", [l for l in r.stdout.splitlines() if l.startswith("{")] or r.stderr[-300:], flush=True)
