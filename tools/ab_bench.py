"""A/B of library variants with bench.py kernel timings. Usage: ab_bench.py name=path ..."""
import json, os, subprocess, sys
for a in sys.argv[1:]:
    name, path = a.split("=", 1)
    env = dict(os.environ)
    if path != "default":
        env["FA_MI355_LIB"] = os.path.abspath(path)
    for r in range(2):
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), "--no-cpu-baseline"],
                             env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            k = d["kernels"]
            print(f"{name:12s} total {d['value']:7.1f} TF  {d['ms_per_step']:.3f} ms | fwd {k['fwd']['ms']:.3f} dkdv {k['bwd_dkdv']['ms']:.3f} dq {k['bwd_dq']['ms']:.3f}", flush=True)
        except Exception as e:
            print(name, "FAILED", out.stderr[-500:])
