"""per-kernel register / spill / LDS table of one .hip file as hipcc compiles it for gfx950 (-Rpass-analysis=kernel-resource-usage); developer tool,
runs without a GPU.  usage: python tools/kernel_resources.py open_clip_amd/csrc/gemm_nt5.hip [-DOCN_DEV_BUILD]"""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Rpass-analysis=kernel-resource-usage",
       "-c", src, "-o", "/dev/null"] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for ln in err.splitlines():
    m = re.search(r"remark: (?:.*?:\d+:\d+: )?\s*([A-Za-z \[\]/]+): (.+?) \[-Rpass", ln)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:70]}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'kernel':70s} VGPR AGPR  SGPR spillS spillV  occ  LDS")
for r in rows:
    print(f"{r['name']:70s} {r.get('VGPRs','?'):>4s} {r.get('AGPRs','?'):>4s} {r.get('SGPRs','?'):>5s} {r.get('SGPRs Spill','?'):>6s} {r.get('VGPRs Spill','?'):>6s} "
          f"{r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('LDS Size [bytes/block]','?'):>5s}")
