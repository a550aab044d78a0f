"""Time line of decode steps from a rocprofv3 kernel trace (CSV), or FETCH_SIZE per kernel from a counter-collection CSV (--pmc).
A "step" = a run of kernels that ends with the attention kernel (+ its combine); prints, per distinct step shape, the kernels in
launch order with duration and the gap to the previous kernel's end (medians over the steps seen)."""
import csv, sys, collections, statistics as st
pmc = sys.argv[1] == "--pmc"
path = sys.argv[2] if pmc else sys.argv[1]
rows = list(csv.DictReader(open(path)))
short = lambda n: n.replace("void fa::", "").replace("(fa::DecArgs)", "").replace("(fa::KArgs)", "").replace("fa::", "")[:90]
if pmc:
    acc = collections.defaultdict(list)
    for r in rows:
        if r.get("Counter_Name") == "FETCH_SIZE" and "fa::" in r["Kernel_Name"]:
            acc[(short(r["Kernel_Name"]), r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    print(f"{'kernel':92s} {'grid':>9s} {'calls':>5s}   FETCH_SIZE median (KiB)   bytes fetched (x 2: gfx950 counts a 128-B request as 64 B)")
    for (k, g), v in sorted(acc.items(), key=lambda x: -st.median(x[1])):
        print(f"{k:92s} {g:>9s} {len(v):5d} {st.median(v):14.0f} KiB  = {2 * st.median(v) * 1024 / 1e6:9.1f} MB")
    sys.exit(0)
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows if "fa::" in r["Kernel_Name"] or "at::" in r["Kernel_Name"]]
ks.sort()
steps, cur = [], []
for i, (s, e, n) in enumerate(ks):
    if not n.startswith(("fa_decode", "decode_combine", "kv_append", "fa_kv")):
        cur = []
        continue
    cur.append((s, e, n))
    nxt = ks[i + 1][2] if i + 1 < len(ks) else ""
    last = n.startswith("decode_combine") or (n.startswith("fa_decode") and not nxt.startswith("decode_combine"))
    if last:
        steps.append(cur); cur = []
shapes = collections.defaultdict(list)
for s in steps:
    shapes[tuple(n for _, _, n in s)].append(s)
for shape, ss in shapes.items():
    if len(ss) < 5:
        continue
    ss = ss[3:]                                        # (warm-up calls)
    print(f"--- {len(ss)} steps")
    tot = []
    for i, n in enumerate(shape):
        dur = st.median([(s[i][1] - s[i][0]) / 1e3 for s in ss])
        gap = st.median([(s[i][0] - s[i - 1][1]) / 1e3 for s in ss]) if i else 0.0
        print(f"   {n:92s} {dur:8.1f} us   gap before {gap:6.1f} us")
    span = st.median([(s[-1][1] - s[0][0]) / 1e3 for s in ss])
    period = st.median([(ss[j + 1][0][0] - ss[j][0][0]) / 1e3 for j in range(len(ss) - 1)]) if len(ss) > 1 else 0
    print(f"   first start -> last end {span:8.1f} us; step period {period:8.1f} us")
