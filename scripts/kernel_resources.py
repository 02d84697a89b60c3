"""Register / LDS / occupancy table of the kernels of one translation unit, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage): python scripts/kernel_resources.py convex.hip 'k_body<2, 2, 2' [extra flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "mundy_amd", "csrc", sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wno-unused-function",
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, []
for ln in err.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("mhip::", "").replace("void ", "")}
        rows.append(cur)
        continue
    for key, rx in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                    ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(rx, ln)
        if m and cur is not None:
            cur[key] = int(m.group(1))
for r in rows:
    if pat in r["name"]:
        print("%-70s VGPR %3d AGPR %3d SGPR %3d scratch %3d LDS %6d occupancy %d" % (r["name"][:70], r.get("vgpr", -1), r.get("agpr", -1), r.get("sgpr", -1), r.get("scratch", -1), r.get("lds", -1), r.get("occ", -1)))
