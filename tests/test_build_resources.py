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
                assert r["spill"] == 0 and r["sgpr_spill"] == 0 and r["scratch"] == 0, (mangled, r)
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
