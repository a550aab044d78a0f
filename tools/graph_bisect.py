"""Which launch of the training step breaks HIP graph capture?  Each variant runs in its own process.
  python tools/graph_bisect.py            (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))


def one(kind):
    import torch, flash_attn
    q = torch.randn(2, 1024, 8, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k, v = (torch.randn(2, 1024, 2, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(2))
    do = torch.randn_like(q)

    def step():
        if kind == "fwd":
            with torch.no_grad():
                return flash_attn.flash_attn_func(q, k, v, causal=True)
        o = flash_attn.flash_attn_func(q, k, v, causal=True)
        if kind == "bwd_dq":
            return torch.autograd.grad(flash_attn.flash_attn_func(q, k.detach(), v.detach(), causal=True), (q,), do)
        if kind == "bwd_dkdv":
            return torch.autograd.grad(flash_attn.flash_attn_func(q.detach(), k, v, causal=True), (k, v), do)
        if kind == "bwd_backward":
            o.backward(do)
            return q.grad
        if kind == "bwd_ret_o":
            return (o,) + torch.autograd.grad(o, (q, k, v), do)
        return torch.autograd.grad(o, (q, k, v), do)

    if kind == "bwd_ref_first":
        ref = [t.clone() for t in step()]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    g.replay()
    torch.cuda.synchronize()
    print("OK", kind, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(sys.argv[1])
    else:
        for kind, env in (("fwd", {}), ("fwd", {"FA_FWD_ASM": "0"}), ("bwd_dq", {}), ("bwd_dq", {"FA_BWD_DQ_ASM": "0"}),
                          ("bwd_dkdv", {}), ("bwd_dkdv", {"FA_BWD_ASM": "0"}), ("bwd", {}), ("bwd_backward", {}), ("bwd_ret_o", {}), ("bwd_ref_first", {}),
                          ("bwd", {"FA_BWD_ASM": "0", "FA_BWD_DQ_ASM": "0", "FA_FWD_ASM": "0"})):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), kind], env=dict(os.environ, **env), capture_output=True, text=True)
            tail = (r.stdout.strip().splitlines() or [""])[-1]
            err = [l for l in r.stderr.splitlines() if "rror" in l or "Fatal" in l][:2]
            print(f"{kind:14s} {env}: rc {r.returncode} {tail} {err}", flush=True)
