"""Print VGPR/AGPR/scratch/LDS/occupancy per kernel from hipcc's resource-usage remarks."""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only",
       "-munsafe-fp-atomics", "-ffp-contract=off", "-mllvm", "-pragma-unroll-threshold=200000",
       "-Rpass-analysis=kernel-resource-usage", *sys.argv[3:],
       "-x", "hip", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    txt = m.group(1).strip()
    if txt.startswith("Function Name:"):
        cur = txt.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in txt:
        k, v = txt.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem)
    if flt and not re.search(flt, dem):
        continue
    print(f"{dem:50s} VGPR {r.get('VGPRs'):>4} AGPR {r.get('AGPRs'):>4} scratch {r.get('ScratchSize [bytes/lane]'):>5} "
          f"occ {r.get('Occupancy [waves/SIMD]'):>2} LDS {r.get('LDS Size [bytes/block]'):>6} SGPR {r.get('TotalSGPRs')}")
