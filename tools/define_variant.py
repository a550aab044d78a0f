"""Side library with one source recompiled under extra -D flags (the other objects come from the product build).
  python tools/define_variant.py <name> <source.hip> -DFOO=1 ...   ->  tools/variants/libfa_<name>.so
  on the GPU:  FA_MI355_LIB=tools/variants/libfa_<name>.so python tools/bench_headdims.py 256"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "flash-attention-v100_amd")
CSRC = os.path.join(PKG, "csrc")
OUT = os.path.join(ROOT, "tools", "variants")
sys.path.insert(0, PKG)
import build as b
name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build()
os.makedirs(OUT, exist_ok=True)
bdir = os.path.join(CSRC, "build")
others = [os.path.join(bdir, s.replace(".hip", ".o")) for s in b.SOURCES if s != src]
obj = os.path.join(OUT, f"def_{name}.o")
subprocess.run([b._hipcc()] + b.FLAGS + flags + ["-c", os.path.join(CSRC, src), "-o", obj], check=True)
lib = os.path.join(OUT, f"libfa_{name}.so")
subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others, check=True)
print("built", lib)
