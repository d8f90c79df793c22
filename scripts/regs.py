"""register / occupancy table of the sweep kernels of a build: python scripts/regs.py [extra hipcc flags...]"""
import os, re, subprocess, sys
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ttcr_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math"] + sys.argv[1:] + \
      ["-Rpass-analysis=kernel-resource-usage", "fsm_capi.hip", "-o", "/tmp/regs_probe.so"]
out = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True)
txt = out.stderr + out.stdout
if out.returncode != 0:
    print(txt[-4000:]); sys.exit(1)
keys = ("TotalSGPRs", "VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]")
rows, cur = [], None
for l in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    for k in keys:
        m = re.search(re.escape(k) + r": (\d+)", l)
        if m and cur is not None and k not in cur: cur[k] = int(m.group(1))
for r in rows:
    if "fsm_sweep_persistent" not in r["name"]: continue
    d = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    d = d.replace("ttcr_amd::", "").replace("void ", "").split("(")[0]
    print("%-72s vgpr %4d occ %d sgpr_spill %4d scratch %4d lds %d" % (d, r.get("VGPRs", -1), r.get("Occupancy [waves/SIMD]", -1), r.get("SGPRs Spill", -1), r.get("ScratchSize [bytes/lane]", -1), r.get("LDS Size [bytes/block]", -1)))
