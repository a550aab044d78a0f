"""The hand-scheduled kernels must own their registers: the build records hipcc's resource remarks for them and fails
on a spill, scratch use or an unloadable register count (a 257-VGPR kernel is rejected by the runtime as invalid ISA)."""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd"))


def test_asm_kernels_have_no_spills_and_fit_the_register_file():
    import build
    build.build()
    res = json.load(open(build.RESOURCES))
    names = {"fa_fwd_asm_kernel": 512, "fa_bwd_dkdv_asm_kernel": 512}
    seen = set()
    for mangled, r in res.items():
        for n, cap in names.items():
            if n in mangled:
                seen.add(n)
                # (SGPR spills go to lanes of the compiler's own VGPRs around the asm statement, not to memory: allowed)
                assert r["spill"] == 0 and r["scratch"] == 0, (mangled, r)
                assert r["vgpr"] <= 256 and r["agpr"] <= 256 and r["vgpr"] + r["agpr"] <= cap, (mangled, r)
    assert seen == set(names), seen


def test_resource_check_rejects_spills():
    import build
    import pytest
    with pytest.raises(RuntimeError):
        build._check_asm_kernels({"_ZN2fa17fa_fwd_asm_kernelIxEEv": {"vgpr": 256, "agpr": 256, "spill": 3, "scratch": 16}})
    with pytest.raises(RuntimeError):
        build._check_asm_kernels({"_ZN2fa16fa_fwd_ws_kernelIxEEv": {"vgpr": 177, "agpr": 80, "spill": 0, "scratch": 0}})
    build._check_asm_kernels({"_ZN2fa13fa_fwd_kernelIxEEv": {"vgpr": 128, "agpr": 0, "spill": 70, "scratch": 300}})   # compiler kernels: not ours to police


def test_generated_headers_are_what_the_generators_emit():
    """The committed *_gen.h asm bodies are machine-written: re-running every generator (build.GENERATED) must reproduce
    them byte for byte - a hand edit of a header, or a generator change without regenerating, fails here (and the
    generators assert their own invariants on the way: LDS read deadlines, register maps, hazard bookkeeping)."""
    import subprocess
    import build
    for gen, hdr, gargs in build.GENERATED:
        out = subprocess.run([sys.executable, os.path.join(build.CSRC, gen)] + gargs, check=True, stdout=subprocess.PIPE,
                             cwd=build.CSRC).stdout
        assert out == open(os.path.join(build.CSRC, hdr), "rb").read(), f"{hdr} is not the output of {gen}"


def test_hazard_checker_inserts_the_gfx950_wait_states():
    """The generators' emitter (gen_fwd_asm.Gen.emit) must separate dependent instructions by the wait states the hardware
    does not interlock: MFMA result -> VALU read 12, VALU result -> MFMA operand 2, trans -> VALU 1, M0 write -> LDS-DMA 1,
    and an LDS read's consumer waits on the exact lgkmcnt of the in-order queue."""
    sys.path.insert(0, os.path.join(ROOT, "flash-attention-v100_amd", "csrc"))
    from gen_fwd_asm import Gen, Ins

    def states_between(g, a, b):
        """wait states (instructions or s_nop counts) emitted strictly between lines a and b"""
        i, j = g.out.index(a), g.out.index(b)
        n = 0
        for line in g.out[i + 1:j]:
            n += int(line.split()[1]) + 1 if line.startswith("s_nop") else 1
        return n

    g = Gen("bf16")
    g.last, g.lds_q, g.srcc_rd, g.out = {}, [], {}, []
    mf = g.mfma("v", 100, "v", 0, "v", 4, True)
    g.emit(mf)
    use = Ins("v_add_f32 v20, v100, v21", "valu", ["v100", "v21"], ["v20"])
    g.emit(use)
    assert states_between(g, mf.txt, use.txt) >= 12
    prod = Ins("v_mov_b32 v4, v30", "valu", ["v30"], ["v4"])
    g.emit(prod)
    mf2 = g.mfma("v", 120, "v", 0, "v", 4, True)
    g.emit(mf2)
    assert states_between(g, prod.txt, mf2.txt) >= 2
    tr = Ins("v_exp_f32 v40, v41", "trans", ["v41"], ["v40"])
    g.emit(tr)
    use2 = Ins("v_add_f32 v42, v40, v40", "valu", ["v40"], ["v42"])
    g.emit(use2)
    assert states_between(g, tr.txt, use2.txt) >= 1
    m0 = Ins("s_add_u32 m0, s49, 4096", "salu", [], ["m0", "scc"])
    g.emit(m0)
    dma = Ins("buffer_load_dwordx4 v22, s[20:23], s54 offen lds", "dma", ["m0", "v22", "s54"], [])
    g.emit(dma)
    assert states_between(g, m0.txt, dma.txt) >= 1
    # in-order LDS returns: three reads in flight, the consumer of the FIRST waits for lgkmcnt(2)
    rds = [Ins(f"ds_read_b128 v[{60 + 4 * i}:{63 + 4 * i}], v24 offset:{1024 * i}", "lds", ["v24"], [f"v{60 + 4 * i + k}" for k in range(4)])
           for i in range(3)]
    for r in rds:
        g.emit(r)
    g.emit(Ins("v_add_f32 v50, v60, v60", "valu", ["v60"], ["v50"]))
    assert "s_waitcnt lgkmcnt(2)" in g.out[g.out.index(rds[2].txt):]


def _resources():
    import build
    build.build()
    return json.load(open(build.RESOURCES))


def test_every_kernel_stays_inside_its_spill_budget():
    """hipcc's resource remarks are recorded for EVERY kernel of the library (build.py: -Rpass-analysis=kernel-resource-usage
    on all sources).  csrc/spill_budget.json holds the spilled-VGPR ceiling of each compiler-scheduled kernel that spills
    today; a kernel that is not listed must not spill, a listed one must not get worse."""
    import build
    res = _resources()
    budget = json.load(open(os.path.join(build.CSRC, "spill_budget.json")))["spill"]
    assert len(res) > 300                                  # all ten sources report, not only the hand-scheduled three
    worse = {n: (r["spill"], budget.get(n, 0)) for n, r in res.items() if r.get("spill", 0) > budget.get(n, 0)}
    assert not worse, f"kernels over their spill budget (spilled, allowed): {worse}"
    stale = [n for n in budget if n not in res]
    assert not stale, f"spill_budget.json lists kernels the build no longer has: {stale}"


def test_kernels_of_the_baseline_configs_do_not_spill():
    """The kernels BASELINE configs 2 - 5 launch: hand-scheduled forward / dK/dV / dQ (config 2, 5), the D = 64 forward and
    backward of config 3, the token-major and head-per-wave decode kernels of config 4 - no VGPR spill, no scratch."""
    res = _resources()
    import subprocess
    names = list(res)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.splitlines()
    # kernel -> spilled VGPRs allowed: none.  (Config 3's dK/dV kernel used to park 5 registers in scratch at three workgroups per
    # CU - the dK / dV store addresses, which hipcc formed in front of the stage loop; its epilogue now builds them from opaque
    # copies of the lane's key and half, profiles/r06_config3.txt section 4.)
    want = {"fa_fwd_asm_kernel<": 0, "fa_bwd_dkdv_asm_kernel<": 0, "fa_bwd_dq_asm_kernel<": 0,
            "fa_fwd_kernel<fa::fp16_tag, 64, 0, false, false, false, 64>": 0, "fa_bwd_dq_kernel<fa::fp16_tag, 64, 0,": 0,
            "fa_bwd_dkdv2_kernel<fa::fp16_tag, 64, 0, false, 64, false>": 0,
            "fa_decode_gemv_tm_kernel<": 0, "decode_combine_kernel": 0, "kv_append_kernel": 0}
    seen = {w: 0 for w in want}
    for n, dm in zip(names, dem):
        for w, allowed in want.items():
            if w in dm:
                seen[w] += 1
                assert res[n].get("spill", 0) <= allowed and res[n].get("scratch", 0) <= 4 * allowed + 4 * (allowed > 0), (dm, res[n])
    assert all(seen.values()), seen


def test_transposed_lds_reads_are_waited_for_before_use():
    """fa_common.h issues ds_read_b64_tr_b16 / _tr_b8 as inline asm with hand-counted `s_waitcnt lgkmcnt(n)` (hipcc would put a
    vmcnt(0) in front of the builtin form); the compiler does not know those asm outputs are in flight.  tools/isa_lds_check.py
    replays the disassembly of every compiler-scheduled kernel with the in-order LDS queue: no instruction may name a
    transposing read's destination before the wait that covers it (a coalescer copy or a hoisted use would read stale data
    silently).  The hand-scheduled *_asm_kernel bodies are not replayed linearly (branches into unrolled copies, called
    routines): their generator's own hazard checker emits and counts every wait (test_hazard_checker_... above)."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build
    import isa_lds_check as chk
    build.build()
    bdir = os.path.join(build.CSRC, "build")
    objs = [os.path.join(bdir, s.replace(".hip", ".o")) for s in build.SOURCES if s not in build.ASM_SOURCES]
    with tempfile.TemporaryDirectory() as wd:
        report, bad = chk.check_objects(objs, wd)
    assert sum(nt for _, nt in report.values()) > 10000          # the reads are there: forward, backward, decode
    assert not bad, bad[:5]
    # the checker does flag the failure it is written for: a copy of the destination in front of the wait
    code = ["ds_read_b64_tr_b16 v[10:11], v5", "ds_read_b64_tr_b16 v[12:13], v5 offset:512", "v_mov_b32_e32 v20, v11",
            "s_waitcnt lgkmcnt(0)", "v_mfma_f32_32x32x16_bf16 a[0:15], v[10:13], v[30:33], a[0:15]"]
    assert len(chk.check_function("synthetic", code)[1]) == 1
    ok = ["ds_read_b64_tr_b16 v[10:11], v5", "ds_read_b64_tr_b16 v[12:13], v5 offset:512", "ds_read_b128 v[40:43], v6",
          "s_load_dword s4, s[0:1], 0x0", "s_waitcnt lgkmcnt(1)", "v_mfma_f32_32x32x16_bf16 a[0:15], v[10:13], v[30:33], a[0:15]"]
    assert chk.check_function("synthetic", ok)[1] == []
