"""Effective shader clock per kernel from a rocprofv3 pass with GRBM_GUI_ACTIVE (csv output):
clock = counter value / kernel duration.  usage: pmc_clock_summary.py <dir>"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
for sub in ("gui", "sq"):
    files = glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True)
    print(f"== {sub}: {files}")
    rows = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")[:60]
            cname = r.get("Counter_Name")
            val = float(r.get("Counter_Value", 0))
            st, en = r.get("Start_Timestamp"), r.get("End_Timestamp")
            dur = (int(en) - int(st)) if st and en else None
            rows[name][cname].append((val, dur))
    for name, cs in rows.items():
        for cname, lst in cs.items():
            vals = [v for v, _ in lst]
            durs = [t for _, t in lst if t]
            line = f"{name:60s} {cname:26s} n={len(vals):3d} mean={sum(vals) / len(vals):.4g}"
            if durs and len(durs) == len(vals):
                ghz = [v / t for v, t in lst if t]
                line += f"  mean dur {sum(durs) / len(durs) / 1e3:.1f} us  count/ns: mean {sum(ghz) / len(ghz):.3f} min {min(ghz):.3f} max {max(ghz):.3f}"
            print(line)
