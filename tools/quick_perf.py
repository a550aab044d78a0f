"""Quick kernel-level timing of the dense forward/backward (development aid, not bench.py)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flash-attention-v100_amd"))
import flash_attn


def timeit(fn, warm=3, it=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    torch.manual_seed(421)
    cfgs = [(8, 4096, 16, 16, 128, True, torch.bfloat16), (8, 4096, 16, 16, 128, False, torch.bfloat16),
            (8, 4096, 16, 16, 128, True, torch.float16), (4, 8192, 16, 16, 128, True, torch.bfloat16),
            (8, 4096, 32, 32, 64, True, torch.bfloat16), (16, 2048, 64, 8, 128, False, torch.bfloat16)]
    bwd = "--bwd" in sys.argv
    for (B, S, H, Hk, D, causal, dt) in cfgs:
        q = torch.randn(B, S, H, D, device="cuda", dtype=dt, requires_grad=bwd)
        k = torch.randn(B, S, Hk, D, device="cuda", dtype=dt, requires_grad=bwd)
        v = torch.randn(B, S, Hk, D, device="cuda", dtype=dt, requires_grad=bwd)
        flops = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
        med, mn = timeit(lambda: flash_attn.flash_attn_func(q, k, v, causal=causal))
        line = f"B{B} S{S} H{H}/{Hk} D{D} causal={causal} {dt}: fwd {med:.3f} ms (min {mn:.3f}) {flops/med/1e9:.1f} TF (best {flops/mn/1e9:.1f})"
        if bwd:
            o = flash_attn.flash_attn_func(q, k, v, causal=causal)
            do = torch.randn_like(o)
            def fb():
                o = flash_attn.flash_attn_func(q, k, v, causal=causal)
                o.backward(do)
            med2, mn2 = timeit(fb)
            line += f" | fwd+bwd {med2:.3f} ms {3.5*flops/med2/1e9:.1f} TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
