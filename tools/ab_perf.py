"""Interleaved A/B timing of library variants in separate subprocesses per round
(each variant = FA_MI355_LIB path).  Usage: ab_perf.py [--bwd] name=path ..."""
import os, subprocess, sys
args = [a for a in sys.argv[1:] if "=" in a]
flags = [a for a in sys.argv[1:] if "=" not in a]
rounds = 2
res = {}
for r in range(rounds):
    for a in args:
        name, path = a.split("=", 1)
        env = dict(os.environ)
        if path != "default":
            env["FA_MI355_LIB"] = os.path.abspath(path)
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "quick_perf.py")] + flags,
                             env=env, capture_output=True, text=True).stdout
        res.setdefault(name, []).append(out)
for name, outs in res.items():
    print("=====", name)
    for o in outs:
        print(o)
