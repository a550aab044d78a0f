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
