"""Extract one kernel from the device assembly of the library and summarise it.
usage: asm_kernel.py <mangled-name-prefix> [out.s] [extra hipcc flags...]   (compiles ttcr_amd/csrc/fsm_capi.hip with -S)"""
import os, re, subprocess, sys
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ttcr_amd", "csrc")
prefix = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/kernel.s"
flags = sys.argv[3:]
asm = "/tmp/_all.s"
if not os.environ.get("REUSE_ASM") or not os.path.exists(asm):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           "--cuda-device-only", "-S", "fsm_capi.hip", "-o", asm] + flags, cwd=csrc, stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().endswith(":") or (l.startswith(prefix) and ": ;" in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
open(out, "w").write("\n".join(body))
ins = [l.strip() for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
def cnt(p): return sum(1 for l in ins if re.match(p, l))
print(lines[start].split(":")[0])
print("instructions", len(ins), "| scratch", cnt(r"scratch_"), "| v_readlane", cnt(r"v_readlane"), "v_writelane", cnt(r"v_writelane"),
      "| s_barrier", cnt(r"s_barrier"), "| ds_", cnt(r"ds_"), "| global/flat", cnt(r"(global|flat)_"), "| v_*f64", cnt(r"v_\w+_f64"))
for i, l in enumerate(body):
    if "scratch_" in l: print("  line", i, l.strip())
