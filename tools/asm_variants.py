"""Build variants of the generated forward body (knock-outs / schedule configs) as side libraries and time them.

  build:  asm_variants.py build name1:"--ko=dma" name2:'--cfg={"lds_lead":5}' ...   (runs here, no GPU)
          -> gpurun_lib/libfa_<name>.so   (reuses the product build's other objects)
  time:   asm_variants.py time [name ...]                                           (on the GPU box)
Each variant is timed in its own process (FA_MI355_LIB selects the library)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "flash-attention-v100_amd")
CSRC = os.path.join(PKG, "csrc")
OUT = os.path.join(ROOT, "tools", "variants")
sys.path.insert(0, PKG)


def build(specs):
    import build as b
    b.build()
    os.makedirs(OUT, exist_ok=True)
    bdir = os.path.join(CSRC, "build")
    which = os.environ.get("VARIANT_KERNEL", "asm")          # asm: forward ; bwd: dK/dV ; dq: dQ
    src, gen, macro = {"asm": ("fa_fwd_asm.hip", "gen_fwd_asm.py", "FA_FWD_ASM_GEN_H"),
                       "bwd": ("fa_bwd_asm.hip", "gen_bwd_dkdv_asm.py", "FA_BWD_ASM_GEN_H"),
                       "dq": ("fa_bwd_dq_asm.hip", "gen_bwd_dq_asm.py", "FA_BWD_DQ_ASM_GEN_H")}[which]
    others = [os.path.join(bdir, s.replace(".hip", ".o")) for s in b.SOURCES if s != src]
    procs = []
    for spec in specs:
        name, _, args = spec.partition(":")
        hdr = os.path.join(OUT, f"gen_{name}.h")
        txt = subprocess.run([sys.executable, os.path.join(CSRC, gen)] + (args.split(" ") if args else []),
                             check=True, stdout=subprocess.PIPE, cwd=CSRC).stdout
        open(hdr, "wb").write(txt)
        obj = os.path.join(OUT, f"asm_{name}.o")
        cmd = [b._hipcc()] + b.FLAGS + [f'-D{macro}="{hdr}"', "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((name, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for name, obj, pr in procs:
        o, _ = pr.communicate()
        if pr.returncode:
            raise RuntimeError(o.decode(errors="replace")[-3000:])
        lib = os.path.join(OUT, f"libfa_{name}.so")
        subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others, check=True)
        print("built", lib)


def one():
    import torch
    import flash_attn
    torch.manual_seed(421)
    res = {}
    for (tag, B, S, H, causal) in (("causal4k", 8, 4096, 16, True), ("full4k", 8, 4096, 16, False), ("causal8k", 4, 8192, 16, True)):
        q, k, v = (torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
        for _ in range(5):
            flash_attn.flash_attn_func(q, k, v, causal=causal)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for s, e in evs:
            s.record(); flash_attn.flash_attn_func(q, k, v, causal=causal); e.record()
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) for s, e in evs)
        fl = 4.0 * B * H * S * S * 128 * (0.5 if causal else 1.0)
        res[tag] = (ts[len(ts) // 2], fl / ts[len(ts) // 2] / 1e9)
    print(json.dumps(res))


def one_bwd():
    """dK/dV kernel alone (+ the preprocess kernel: only dk, dv are asked for)"""
    import torch
    import flash_attn
    torch.manual_seed(421)
    res = {}
    for (tag, B, S, H, Hk, causal) in (("causal4k", 8, 4096, 16, 16, True), ("full4k", 8, 4096, 16, 16, False), ("gqa4k", 4, 4096, 32, 8, True)):
        q = torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        k, v = (torch.randn(B, S, Hk, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(2))
        do = torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16)
        o = flash_attn.flash_attn_func(q.detach(), k, v, causal=causal)      # dk, dv only: preprocess + dK/dV kernel
        if True:                                # dK/dV alone (+ the preprocess kernel): only dk, dv are asked for
            for _ in range(3):
                torch.autograd.grad(o, (k, v), do, retain_graph=True)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                torch.autograd.grad(o, (k, v), do, retain_graph=True)
            e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        fl = 8.0 * B * H * S * S * 128 * (0.5 if causal else 1.0)
        res[tag] = (ms, fl / ms / 1e9)
    print(json.dumps(res))


def one_dq():
    """dQ kernel alone (K / V frozen: only dq is asked for)"""
    import torch
    import flash_attn
    torch.manual_seed(421)
    res = {}
    for (tag, B, S, H, Hk, causal) in (("causal4k", 8, 4096, 16, 16, True), ("full4k", 8, 4096, 16, 16, False), ("causal8k", 4, 8192, 16, 16, True)):
        q = torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        k, v = (torch.randn(B, S, Hk, 128, device="cuda", dtype=torch.bfloat16) for _ in range(2))
        do = torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16)
        o = flash_attn.flash_attn_func(q, k, v, causal=causal)
        for _ in range(3):
            torch.autograd.grad(o, (q,), do, retain_graph=True)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for s, e in evs:
            s.record(); torch.autograd.grad(o, (q,), do, retain_graph=True); e.record()
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) for s, e in evs)
        fl = 6.0 * B * H * S * S * 128 * (0.5 if causal else 1.0)          # executed: 3 GEMMs
        res[tag] = (ts[len(ts) // 2], fl / ts[len(ts) // 2] / 1e9)
    print(json.dumps(res))


def time_all(names):
    if not names:
        names = sorted(f[6:-3] for f in os.listdir(OUT) if f.startswith("libfa_") and f.endswith(".so"))
    names = ["default"] + names
    for rnd in range(2):
        for n in names:
            env = dict(os.environ)
            if n != "default":
                env["FA_MI355_LIB"] = os.path.join(OUT, f"libfa_{n}.so")
            mode = {"bwd": "one_bwd", "dq": "one_dq"}.get(os.environ.get("VARIANT_KERNEL"), "one")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), mode], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(n, "FAILED", r.stderr[-500:])
                continue
            d = json.loads(line[0])
            print(f"{n:28s} " + "  ".join(f"{k}: {v[0]:.3f} ms {v[1]:6.0f} TF" for k, v in d.items()), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "one":
        one()
    elif sys.argv[1] == "one_bwd":
        one_bwd()
    elif sys.argv[1] == "one_dq":
        one_dq()
    else:
        time_all(sys.argv[2:])

