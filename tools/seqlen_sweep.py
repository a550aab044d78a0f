"""Hand-scheduled vs compiler-scheduled kernels across sequence lengths at a fixed token count (bf16 causal, H16 D128):
where should the dispatch switch?   python tools/seqlen_sweep.py          (spawns itself with FA_*_ASM=0 for the other arm)"""
import json, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
SEQS = tuple(int(x) for x in os.environ.get("SWEEP_SEQS", "256,384,512,768,1024,2048,4096,8192").split(","))
TOKENS = int(os.environ.get("SWEEP_TOKENS", "32768"))


def one():
    import torch, flash_attn
    torch.manual_seed(421)
    H, D, TOK = 16, 128, TOKENS
    res = {}
    for S in SEQS:
        B = TOK // S
        q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        k = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        v = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        do = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)

        def t(fn, n=15):
            for _ in range(3): fn()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            torch.cuda.synchronize()
            for s, e in evs:
                s.record(); fn(); e.record()
            torch.cuda.synchronize()
            ts = sorted(s.elapsed_time(e) for s, e in evs)
            return ts[n // 2]
        with torch.no_grad():
            tf = t(lambda: flash_attn.flash_attn_func(q, k, v, causal=True))
        o1 = flash_attn.flash_attn_func(q, k.detach(), v.detach(), causal=True)
        tdq = t(lambda: torch.autograd.grad(o1, (q,), do, retain_graph=True))
        o2 = flash_attn.flash_attn_func(q, k, v, causal=True)
        tall = t(lambda: torch.autograd.grad(o2, (q, k, v), do, retain_graph=True))
        res[S] = (tf, tdq, tall - tdq)
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
        sys.exit(0)
    arms = {"asm": {}, "compiler": {"FA_FWD_ASM": "0", "FA_BWD_ASM": "0", "FA_BWD_DQ_ASM": "0"}}
    out = {}
    for name, env in arms.items():
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        out[name] = json.loads(line[0]) if line else None
        if not line:
            print(name, "FAILED", r.stderr[-800:])
    print(f"{'S':>6s} | {'fwd asm':>8s} {'comp':>8s} | {'dQ asm':>8s} {'comp':>8s} | {'dK/dV asm':>9s} {'comp':>8s}   (ms, {TOKENS} tokens, H16 D128 bf16 causal)")
    for S in SEQS:
        a, c = out["asm"][str(S)], out["compiler"][str(S)]
        print(f"{S:6d} | {a[0]:8.3f} {c[0]:8.3f} | {a[1]:8.3f} {c[1]:8.3f} | {a[2]:9.3f} {c[2]:8.3f}")
