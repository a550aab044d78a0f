"""Ahead-of-time build of libfa_mi355.so (hand-written HIP for gfx950).

`hipcc` cross-compiles without a GPU, so the library is built in-tree in the build
container and travels to the GPU box with the repository snapshot.  No torch headers are
involved: the library is a plain C-ABI shared object (include/fa_mi355.h)."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "flash_attn_mi355")
LIB = os.path.join(OUT_DIR, "libfa_mi355.so")
SOURCES = ["fa_api.hip", "fa_fwd.hip", "fa_fwd_d256.hip", "fa_fwd_asm.hip", "fa_bwd.hip", "fa_bwd_d256.hip", "fa_bwd_asm.hip", "fa_bwd_dq_asm.hip", "fa_bwd_dq_ds.hip", "fa_kvcache.hip", "fa_decode.hip", "fa_rows.hip"]
GENERATED = [("gen_fwd_asm.py", "fa_fwd_asm_gen.h", []),
             ("gen_bwd_dkdv_asm.py", "fa_bwd_asm_gen.h", []),
             ("gen_bwd_dq_asm.py", "fa_bwd_dq_asm_gen.h", [])]  # (generator, header, arguments): hand-scheduled asm bodies
EXTRA_FLAGS = {"fa_fwd_d256.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}      # per-source compiler switches (see the file's header)
ASM_SOURCES = ("fa_fwd_asm.hip", "fa_bwd_asm.hip", "fa_bwd_dq_asm.hip")   # kernels whose body is one hand-scheduled asm statement
RESOURCES = os.path.join(OUT_DIR, "kernel_resources.json")              # their register / scratch use, checked at build time
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
         "-I" + CSRC, "-Wno-unused-value"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(p.encode())
        h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def _generate():
    """Re-run a kernel-body generator when its header is missing or older than the script."""
    for gen, hdr, gargs in GENERATED:
        g, h = os.path.join(CSRC, gen), os.path.join(CSRC, hdr)
        if not os.path.exists(h) or os.path.getmtime(h) < os.path.getmtime(g):
            txt = subprocess.run([sys.executable, g] + gargs, check=True, stdout=subprocess.PIPE, cwd=CSRC).stdout
            with open(h, "wb") as f:
                f.write(txt)


def _asm_kernel_resources(log):
    """Parse hipcc's -Rpass-analysis=kernel-resource-usage remarks: {kernel: {"vgpr", "agpr", "sgpr", "spill", "scratch", "lds"}}."""
    import re
    res, cur = {}, None
    for line in log.splitlines():
        m = re.search(r"remark: +(?:[^ ]+: )?(Function Name|VGPRs|AGPRs|SGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = res.setdefault(v, {})
        elif cur is not None:
            key = {"VGPRs": "vgpr", "AGPRs": "agpr", "SGPRs": "sgpr", "VGPRs Spill": "spill", "SGPRs Spill": "sgpr_spill",
                   "ScratchSize [bytes/lane]": "scratch", "LDS Size [bytes/block]": "lds"}[k]
            cur[key] = int(v)
    return res


def _check_asm_kernels(all_res):
    """A hand-scheduled body owns its registers: a spill, scratch use or more than 256 + 256 registers means the
    compiler had to work around the asm statement's constraints (and a 257-register kernel does not even load)."""
    bad = []
    for name, r in all_res.items():
        if "asm_kernel" not in name and "ws_kernel" not in name:
            continue
        total_cap = 256 if "ws_kernel" in name else 512            # two waves per SIMD | one
        # (SGPR spills are allowed: they live in lanes of the compiler's own v0..v15 around the asm statement - the paged forward
        #  pins 85 scalar registers -, never in memory; what must not happen is scratch or a VGPR spill)
        if (r.get("spill", 0) or r.get("scratch", 0) or r.get("vgpr", 0) > 256 or r.get("agpr", 0) > 256
                or r.get("vgpr", 0) + r.get("agpr", 0) > total_cap):
            bad.append(f"{name}: {r}")
    if bad:
        raise RuntimeError("asm kernel resource check failed:\n" + "\n".join(bad))


def build(force=False, verbose=False, defines=(), out=None):
    """Compile every HIP source for gfx950 and link libfa_mi355.so. Returns the path.
    `defines` / `out` build an experiment variant next to the product library (A/B runs
    select it with FA_MI355_LIB=<path>)."""
    _generate()
    if defines or out:
        return _build_variant(list(defines), out, verbose)
    bdir = os.path.join(CSRC, "build")
    os.makedirs(bdir, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".py"))]
    deps.append(os.path.join(ROOT, "include", "fa_mi355.h"))
    stamp = os.path.join(bdir, "stamp.txt")
    dig = _digest(deps)
    if (not force and os.path.exists(LIB) and os.path.exists(stamp)
            and open(stamp).read().strip() == dig):
        return LIB
    hipcc = _hipcc()
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        objs.append(obj)
        extra = ["-Rpass-analysis=kernel-resource-usage"]         # every kernel's registers / spills / scratch -> kernel_resources.json
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    resources = {}
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
        resources.update(_asm_kernel_resources(out.decode(errors="replace")))
    _check_asm_kernels(resources)                                # (hand-scheduled bodies: no spill, no scratch; the compiler kernels'
                                                                 #  spill budget is tests/test_build_resources.py + csrc/spill_budget.json)
    import json
    with open(RESOURCES, "w") as f:
        json.dump(resources, f, indent=1, sort_keys=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if out.returncode != 0:
        raise RuntimeError("link failed:\n" + out.stdout.decode(errors="replace"))
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


PROBE_SRC = os.path.join(ROOT, "tools", "probes", "probe_mfma_ceiling.hip")
PROBE_LIB = os.path.join(ROOT, "tools", "probes", "libfa_probe.so")


def build_probe(force=False):
    """Measurement aid next to the product library (never linked into it): the chip-wide random-operand MFMA loop bench.py
    runs after the timed steps for roofline.practical_ceiling (tools/probes/probe_mfma_ceiling.hip)."""
    if (not force and os.path.exists(PROBE_LIB) and os.path.getmtime(PROBE_LIB) >= os.path.getmtime(PROBE_SRC)):
        return PROBE_LIB
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", PROBE_SRC, "-o", PROBE_LIB],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("probe build failed:\n" + r.stdout.decode(errors="replace"))
    return PROBE_LIB


def _build_variant(defines, out, verbose):
    tag = hashlib.sha256(" ".join(defines).encode()).hexdigest()[:8]
    bdir = os.path.join(CSRC, "build", "var_" + tag)
    os.makedirs(bdir, exist_ok=True)
    out = out or os.path.join(OUT_DIR, f"libfa_mi355_{tag}.so")
    hipcc = _hipcc()
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + [d if d.startswith("-") else "-D" + d for d in defines] + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, pr in procs:
        o, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{o.decode(errors='replace')}")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    return out


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":
        # python build.py --variant out.so DEF1 DEF2=3 ...
        print(build(out=os.path.abspath(sys.argv[2]), defines=sys.argv[3:]))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_probe(force="--force" in sys.argv))
