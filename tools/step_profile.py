"""Where do the K timed steps of bench.py lose 2-3 % against the sustained step?  (ms_per_step 2.26 vs 2.20 evented / sustained)

  python tools/step_profile.py [warmup] [steps] [preheat_ms]      (preheat: that many ms of forward launches in front of the warm-up)

Replays bench.py's preparation and step in a fresh process and records a HIP event behind EVERY step (warm-up included) plus
the host time at which each step's launches were queued: per-step GPU ms, the gap between the barrier and the first kernel,
and how far the host runs ahead of the GPU.
"""
import os, sys, time
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import flash_attn

W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
B, H, S, D = 8, 16, 4096, 128
g = torch.Generator(device="cpu").manual_seed(421)
mk = lambda: torch.randn(B, S, H, D, generator=g).to(torch.bfloat16).to(dev)
q, k, v, do = mk(), mk(), mk(), mk()
for t in (q, k, v):
    t.requires_grad_(True)


def step():
    o = flash_attn.flash_attn_func(q, k, v, causal=True)
    o.backward(do)
    q.grad = k.grad = v.grad = None


PRE = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
if PRE > 0:
    with torch.no_grad():
        flash_attn.flash_attn_func(q, k, v, causal=True); torch.cuda.synchronize()
        for _ in range(int(PRE / 0.5) + 1):
            flash_attn.flash_attn_func(q, k, v, causal=True)
    torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(W + K + 2)]
host = []
ev[0].record()
for i in range(W):
    step(); ev[i + 1].record()
torch.cuda.synchronize(); torch.cuda.synchronize()
t0 = time.perf_counter()
ev[W + 1].record()
for i in range(K):
    h0 = time.perf_counter()
    step(); ev[W + 2 + i].record()
    host.append((h0 - t0, time.perf_counter() - t0))
t_q = time.perf_counter() - t0
torch.cuda.synchronize(); torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"warm-up steps (ms): " + " ".join(f"{ev[i].elapsed_time(ev[i + 1]):.3f}" for i in range(W)))
ts = [ev[W + 1 + i].elapsed_time(ev[W + 2 + i]) for i in range(K)]
print("timed steps   (ms): " + " ".join(f"{t:.3f}" for t in ts))
print(f"wall clock {el * 1e3:.3f} ms = {el / K * 1e3:.4f} per step; events first -> last {ev[W + 1].elapsed_time(ev[W + 1 + K]):.3f} ms; "
      f"host finished queueing at {t_q * 1e3:.3f} ms; sum of steps {sum(ts):.3f}")
print("host queue times of steps (start, end ms): " + " ".join(f"({a * 1e3:.2f},{b * 1e3:.2f})" for a, b in host[:6]))

# ---- how long an idle gap resets the ramp: K steps (steady state), sleep, 8 evented steps --------------------------------------
for gap in (0.0, 0.001, 0.005, 0.02, 0.1, 0.5):
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    time.sleep(gap)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
    e[0].record()
    for i in range(8):
        step(); e[i + 1].record()
    torch.cuda.synchronize()
    print(f"idle {gap * 1e3:6.1f} ms, then steps (ms): " + " ".join(f"{e[i].elapsed_time(e[i + 1]):.3f}" for i in range(8)))
