"""Print registers / spills / scratch of every kernel in a .hip source (hipcc -Rpass-analysis).
Usage: kernel_resources.py csrc/fa_bwd.hip [-DNAME ...]"""
import os, re, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
       "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "flash-attention-v100_amd", "csrc"),
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: +(.*?): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("void ", "").replace("fa::", "")
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
print(f"{'kernel':75s} {'VGPR':>5s} {'AGPR':>5s} {'spill':>6s} {'scratch':>8s} {'occ':>4s}")
for k, r in rows.items():
    print(f"{k[:75]:75s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('VGPRs Spill','?'):>6s} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('Occupancy [waves/SIMD]','?'):>4s}")
