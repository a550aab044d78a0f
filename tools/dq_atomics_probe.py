"""dQ summed out of the dK/dV kernel with fp32 atomics: what does the atomic stream cost?  (round-5 review, task 1a)

  python tools/dq_atomics_probe.py [seconds_per_mode]      -> text on stdout (committed as profiles/r06_dq_atomics.txt)

Runs tools/probes/probe_dq_atomics.hip (the dK/dV launch shape of BASELINE config 2 without its arithmetic: 2048 workgroups,
132 stages each, a 32 x 128 fp32 partial per workgroup and stage = 4.43 GB of atomic operands onto a 268 MB buffer) in its
modes - adds alone, the MFMA stream alone (32 = today's dK/dV stage, 40 = with a fifth GEMM), both together; device scope
and `sc1`; the unit-per-XCD placement of the product grids and a scattered one; plain stores as the floor of the write
pattern - and samples socket power and shader clock (amdsmi) while each mode loops.  Every atomic mode is first VERIFIED
(each add 1.0: row block i must end at i / 4 + 1).
"""
import ctypes, os, subprocess, sys, threading, time
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "probes", "probe_dq_atomics.hip")
LIB = os.path.join(HERE, "probes", "libfa_probe_dq_atomics.so")
BYTES = 128 * 16 * 132 * 32 * 128 * 4


def build():
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", SRC, "-o", LIB], check=True)


def main():
    build()
    if "--build-only" in sys.argv:
        return
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
    lib = ctypes.CDLL(LIB)
    fn = lib.fa_probe_dq_atomics
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)]
    sys.path.insert(0, HERE)
    from clock_power import Smi          # (amdsmi / sysfs / cli reader of tools/clock_power.py)
    smi = Smi()
    print("SMI source:", smi.kind, getattr(smi, "err", ""))
    print(f"atomic operand bytes per launch: {BYTES / 1e9:.3f} GB (2048 workgroups x 132 stages x 16 KiB) onto 268 MB")

    def run(name, mode, mf, verify):
        ms, bad = ctypes.c_float(), ctypes.c_double()
        rc = fn(mode, mf, 3, verify, ctypes.byref(ms), ctypes.byref(bad))          # verification + a first timing
        assert rc == 0, rc
        n = max(3, int(seconds * 1e3 / max(ms.value, 1e-3)))
        samples, stop = [], threading.Event()

        def sampler():
            while not stop.is_set():
                try:
                    samples.append((time.perf_counter(), smi.read()))
                except Exception:                # noqa: BLE001
                    pass
                time.sleep(0.05)
        th = threading.Thread(target=sampler, daemon=True)
        t0 = time.perf_counter()
        th.start()
        ms2, b2 = ctypes.c_float(), ctypes.c_double()
        rc = fn(mode, mf, n, 0, ctypes.byref(ms2), ctypes.byref(b2))
        t1 = time.perf_counter()
        stop.set(); th.join()
        assert rc == 0, rc
        tail = [x for t, x in samples if t - t0 > 0.5 * (t1 - t0) and x]
        ws = [x[0] for x in tail if isinstance(x[0], (int, float))]
        cs = [x[1][0] if isinstance(x[1], tuple) else x[1] for x in tail if x[1] is not None]
        w = f"{sum(ws) / len(ws):.0f} W" if ws else "n/a"
        c = f"{sum(cs) / len(cs):.0f} MHz" if cs else "n/a"
        line = f"{name:<58s} {ms2.value:7.4f} ms"
        if mode & 9:
            line += f"  {BYTES / ms2.value / 1e9:6.2f} TB/s of operands"
        if mode & 2:
            line += f"  {2048 * 4 * 132 * mf * 32768 / ms2.value / 1e9:6.0f} TFLOP/s MFMA"
        line += f"  {w}  {c}  ({n} launches)"
        if verify:
            line += f"  verify: {'ok' if bad.value == 0 else 'WRONG %.3g of the elements' % bad.value}"
        print(line, flush=True)
        return ms2.value

    r = {}
    r["st"] = run("plain stores, same pattern (floor)", 8, 32, 1)
    r["at"] = run("atomics alone, unit per XCD", 1, 32, 1)
    r["at_sc1"] = run("atomics alone, unit per XCD, sc1 (system scope)", 5, 32, 1)
    r["at_sc"] = run("atomics alone, units scattered over the XCDs", 17, 32, 1)
    r["at_sc_sc1"] = run("atomics alone, scattered, sc1", 21, 32, 1)
    r["mf32"] = run("MFMA stream alone, 32 per wave-stage (today's dK/dV)", 2, 32, 0)
    r["mf40"] = run("MFMA stream alone, 40 per wave-stage (fifth GEMM)", 2, 40, 0)
    r["mf32_st"] = run("MFMA 32 + plain stores", 10, 32, 0)
    r["mf40_st"] = run("MFMA 40 + plain stores", 10, 40, 0)
    r["mf40_at"] = run("MFMA 40 + atomics, unit per XCD", 3, 40, 1)
    r["mf40_at_sc1"] = run("MFMA 40 + atomics, unit per XCD, sc1", 7, 40, 1)
    r["mf40_at_sc"] = run("MFMA 40 + atomics, scattered", 19, 40, 1)
    r["mf40_at_sc_sc1"] = run("MFMA 40 + atomics, scattered, sc1", 23, 40, 1)
    print()
    print(f"atomics un-overlapped: {r['at']:.3f} ms; beside the 40-MFMA stream they add {r['mf40_at'] - r['mf40']:+.3f} ms "
          f"({r['mf40_at']:.3f} vs {r['mf40']:.3f}); the fifth GEMM alone adds {r['mf40'] - r['mf32']:+.3f} ms to the bare stream")


if __name__ == "__main__":
    main()
