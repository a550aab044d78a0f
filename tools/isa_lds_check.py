"""ISA check: no instruction touches the destination of a transposing LDS read before the counted wait that covers it.

Why (round-5 advisor finding): fa_common.h issues `ds_read_b64_tr_b16` / `_tr_b8` as inline asm (lds_read_tr16_nw / lds_read_tr8_nw)
so that hipcc's waitcnt pass does not put an `s_waitcnt vmcnt(0)` in front of them; the matching wait is a hand-placed
`s_waitcnt lgkmcnt(n)` tied to the assembled fragment (lds_tr_wait).  The compiler does not know the asm outputs are in
flight: if the register coalescer ever copies one (a v_mov between the read and the wait) or hoists a use, an MFMA reads stale
data and nothing fails at build time.  This script disassembles the device code of the built objects and replays every
function linearly with the in-order LDS return queue:

  * every LDS instruction (ds_*) enters the queue - LDS returns in order;
  * `s_waitcnt lgkmcnt(n)` retires all but the n youngest LDS entries.  Scalar memory instructions share the counter and
    return out of order, but they are NOT in the queue: the counter is (outstanding LDS + outstanding SMEM), so "counter <= n"
    bounds the outstanding LDS instructions by n whether the scalar loads have returned or not (pending ones only make the
    wait stricter) - which is also why the hand-placed counts never include them;
  * any other instruction that names a register of an outstanding transposing read's destination is a violation;
  * the replay is linear (fall-through paths): the queue is dropped at s_endpgm / s_branch / s_setpc.

  python tools/isa_lds_check.py            -> per-object summary, exit code 1 on a violation
Used by tests/test_build_resources.py::test_transposed_lds_reads_are_waited_for_before_use.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
LLVM = "/opt/rocm/lib/llvm/bin"
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
WAIT = re.compile(r"lgkmcnt\((\d+)\)")


def disassemble(obj, workdir):
    """device code of a HIP object file -> list of (function, [instruction text ...])"""
    base = os.path.join(workdir, os.path.basename(obj))
    r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={base}.fat", obj], capture_output=True, text=True)
    if r.returncode != 0:
        if "not found" in r.stderr:                 # a source without kernels (fa_api.hip)
            return []
        raise RuntimeError(r.stderr)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={base}.fat", f"--output={base}.co", "--unbundle"], check=True)
    txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", f"{base}.co"], check=True, capture_output=True, text=True).stdout
    funcs, cur = [], None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if cur is None or m.group(1).startswith("_Z"):      # (local labels of the asm bodies are not function starts)
                cur = (m.group(1), [])
                funcs.append(cur)
            continue
        if cur is None or not line.startswith("\t"):
            continue
        ins = line.split("//")[0].strip()
        if ins:
            cur[1].append(ins)
    return funcs


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def is_lgkm(op):
    return op.startswith("ds_")


def check_function(name, code):
    """-> (number of transposing reads, list of violation strings)"""
    queue = []          # outstanding LGKM ops, oldest first: (kind, dst registers or None, text)
    n_tr, bad = 0, []
    for idx, ins in enumerate(code):
        op = ins.split()[0]
        if op == "s_waitcnt":
            m = WAIT.search(ins)
            if m:                                   # (an s_waitcnt without an lgkmcnt field leaves the LGKM queue alone)
                n = int(m.group(1))
                queue = queue[len(queue) - n:] if 0 < n < len(queue) else ([] if n == 0 else queue)
            continue
        if op in ("s_endpgm", "s_branch", "s_setpc_b64"):   # what follows an unconditional jump is not reached from here (the replay is
            queue = []                                       # linear: it covers fall-through paths, i.e. the straight-line tile loops)
            continue
        touched = regs_of(ins)
        for kind, dst, text in queue:
            if kind == "tr" and dst & touched:
                bad.append(f"{name}: `{ins}` (instruction {idx}) names {sorted(dst & touched)} while `{text}` is in flight")
        if is_lgkm(op):
            if "_tr_b" in op:
                n_tr += 1
                dst = regs_of(ins.split(",")[0])
                queue.append(("tr", dst, ins))
            else:
                queue.append(("lds", None, ins))
    return n_tr, bad


def check_objects(objs, workdir):
    report, bad = {}, []
    for obj in objs:
        n_f = n_tr = 0
        for name, code in disassemble(obj, workdir):
            t, b = check_function(name, code)
            n_f += 1
            n_tr += t
            bad += b
        report[os.path.basename(obj)] = (n_f, n_tr)
    return report, bad


def main():
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))
    import build
    build.build()
    bdir = os.path.join(build.CSRC, "build")
    objs = [os.path.join(bdir, s.replace(".hip", ".o")) for s in build.SOURCES]
    with tempfile.TemporaryDirectory() as wd:
        report, bad = check_objects(objs, wd)
    for o, (nf, nt) in report.items():
        print(f"{o:24s} {nf:4d} functions, {nt:6d} transposing LDS reads")
    import collections
    per_fn = collections.Counter(b.split(":")[0] for b in bad)
    asm = {f: n for f, n in per_fn.items() if "asm_kernel" in f}
    real = [b for b in bad if "asm_kernel" not in b.split(":")[0]]
    for b in real[:40]:
        print("VIOLATION", b)
    print(f"{len(real)} violations in compiler-scheduled kernels; {sum(asm.values())} linear-replay hits in {len(asm)} hand-scheduled "
          f"bodies (not meaningful there: branches into unrolled copies and called routines - their generator counts every wait)")
    return 1 if real else 0


if __name__ == "__main__":
    sys.exit(main())
