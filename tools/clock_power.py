"""Power and shader clock while each kernel of the step runs in a ~2 s loop.

  python tools/clock_power.py [seconds]          -> text on stdout (committed as profiles/r03_clock_power.txt)

Sampler: a thread reads the SMI every ~50 ms while the main thread keeps the queue full.  Sources tried in order: the
`amdsmi` python module, hwmon / pp_dpm_sclk in sysfs, `rocm-smi --json` / `amd-smi metric --json` subprocesses.
The bare-MFMA comparison point is tools/probes/probe_issue_rate (chip-wide MFMA loop, constant vs random operands).
"""
import glob, json, os, subprocess, sys, threading, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
import torch, flash_attn


class Smi:
    def __init__(self):
        self.kind, self.h = None, None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self.h = amdsmi.amdsmi_get_processor_handles()[0]
            self.amdsmi = amdsmi
            self.read()
            self.kind = "amdsmi"
            return
        except Exception as e:              # noqa: BLE001
            self.err = repr(e)
        cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") +
                       glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
        if cards:
            self.pw = cards[0]
            self.dev = os.path.dirname(os.path.dirname(os.path.dirname(self.pw)))
            self.kind = "sysfs"
            return
        for cmd in (["amd-smi", "metric", "-g", "0", "--power", "--clock", "--json"], ["rocm-smi", "--showpower", "--showclocks", "--json"]):
            try:
                subprocess.run(cmd, capture_output=True, timeout=20, check=True)
                self.cmd, self.kind = cmd, "cli"
                return
            except Exception:               # noqa: BLE001
                pass

    def read(self):
        """-> (watts, sclk MHz) or None"""
        if self.kind == "amdsmi" or (self.kind is None and self.h is not None):
            a = self.amdsmi
            m = a.amdsmi_get_gpu_metrics_info(self.h)
            w = m.get("current_socket_power") or m.get("average_socket_power")
            clk = m.get("current_gfxclks") or m.get("current_gfxclk") or m.get("average_gfxclk_frequency")
            if isinstance(clk, (list, tuple)):
                vals = [c for c in clk if isinstance(c, (int, float)) and 0 < c < 60000]
                clk = (sum(vals) / len(vals), max(vals), min(vals)) if vals else None
            return w, clk
        if self.kind == "sysfs":
            w = int(open(self.pw).read()) / 1e6
            clk = None
            try:
                for line in open(os.path.join(self.dev, "pp_dpm_sclk")):
                    if "*" in line:
                        clk = float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
            except Exception:               # noqa: BLE001
                pass
            return w, clk
        if self.kind == "cli":
            r = subprocess.run(self.cmd, capture_output=True, text=True, timeout=20)
            return r.stdout.strip()[:2000], None
        return None


def loop(name, fn, seconds, smi, flops):
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                samples.append((time.perf_counter(), smi.read()))
            except Exception as e:          # noqa: BLE001
                samples.append((time.perf_counter(), ("error", repr(e))))
            time.sleep(0.05 if smi.kind != "cli" else 0.3)

    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler, daemon=True)
    t0 = time.perf_counter()
    th.start()
    n = 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e.record()
    torch.cuda.synchronize()
    stop.set(); th.join()
    ms = s.elapsed_time(e) / n
    tail = [x for t, x in samples if t - t0 > 0.5 * seconds and x and x[0] != "error"]      # second half: settled state
    print(f"{name}: {n} launches, {ms:.4f} ms each -> {flops / ms / 1e9:.0f} TFLOP/s algorithmic; {len(samples)} SMI samples ({smi.kind})")
    if smi.kind == "cli":
        print("   last sample:", tail[-1][0] if tail else samples[-1:])
        return
    ws = [x[0] for x in tail if isinstance(x[0], (int, float))]
    cs = [x[1] for x in tail if x[1] is not None]
    if ws:
        print(f"   socket power (settled half): mean {sum(ws) / len(ws):.0f} W  min {min(ws):.0f}  max {max(ws):.0f}")
    if cs and isinstance(cs[0], tuple):
        print(f"   gfx clock over XCDs (settled half): mean {sum(c[0] for c in cs) / len(cs):.0f} MHz  max-of-XCDs {max(c[1] for c in cs):.0f}  min-of-XCDs {min(c[2] for c in cs):.0f}")
    elif cs:
        print(f"   sclk (settled half): mean {sum(cs) / len(cs):.0f} MHz  min {min(cs):.0f}  max {max(cs):.0f}")


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    smi = Smi()
    print("SMI source:", smi.kind, getattr(smi, "err", ""))
    idle = smi.read() if smi.kind else None
    print("idle reading:", idle if smi.kind != "cli" else str(idle)[:400])
    torch.manual_seed(421)
    B, S, H, D = 8, 4096, 16, 128
    q, k, v, do = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
    ff = 4.0 * B * H * S * S * D / 2
    with torch.no_grad():
        loop("fwd  (fa_fwd_asm_kernel, causal 4k)", lambda: flash_attn.flash_attn_func(q, k, v, causal=True), seconds, smi, ff)
        loop("fwd  non-causal 4k", lambda: flash_attn.flash_attn_func(q, k, v, causal=False), seconds, smi, 2 * ff)
        z = torch.zeros_like(q)
        loop("fwd  causal 4k, ZERO inputs (same instruction stream, no data toggling)", lambda: flash_attn.flash_attn_func(z, z, z, causal=True), seconds, smi, ff)
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    o = flash_attn.flash_attn_func(q, k, v, causal=True)
    oq = flash_attn.flash_attn_func(q, k.detach(), v.detach(), causal=True)      # frozen K / V: the backward is the dQ kernel alone
    loop("bwd dQ kernel (dq only)", lambda: torch.autograd.grad(oq, (q,), do, retain_graph=True), seconds, smi, 0.5 * ff)
    loop("bwd dQ + dK/dV kernels", lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True), seconds, smi, 2.5 * ff)

    def step():
        oo = flash_attn.flash_attn_func(q, k, v, causal=True)
        oo.backward(do)
        q.grad = k.grad = v.grad = None
    loop("fwd+bwd step", step, seconds, smi, 3.5 * ff)


if __name__ == "__main__":
    main()
