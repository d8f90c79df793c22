"""HBM traffic of the sweep kernel of one profiled command from the two PMC passes (FETCH_SIZE / WRITE_SIZE per dispatch, separate rocprofv3
--pmc runs, no trace domains), calibrated IN THE SAME RUN: every profiled process also runs torch.add(a, 1.0, out=b) on 2^28 floats (a coalesced
16-byte-per-lane elementwise kernel of known size: 2^30 bytes read, 2^30 written), and the factor between the bytes that kernel moved and what
the counters report for it is what the sweep kernel's counters are multiplied with (gfx950: FETCH_SIZE counts a 128-byte request as 64 bytes
-- the guide's x2; WRITE_SIZE counts true).  Tagged with the hash of the kernel sources (ttcr_amd.build.source_hash) so that bench.py only
reports it for the library it was measured with.
usage: pmc_to_json.py <dir with <tag>_{FETCH,WRITE}_SIZE_summary.csv> <tag> <size> <sources> <out.json> [kernel substring = fsm_sweep_persistent]"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ttcr_amd.build import source_hash

d, tag, size, sources, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
want = sys.argv[6] if len(sys.argv) > 6 else "fsm_sweep_persistent"
vals, disp, calib, kern = {}, {}, {}, None
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(os.path.join(d, f"{tag}_{c}_summary.csv"))):
        if want in r["Kernel_Name"] and c not in vals:
            vals[c] = float(r["PerDispatch_KB"]); disp[c] = int(r["Dispatches"]); kern = r["Kernel_Name"]
        # the calibration kernel: torch.add(a, 1.0, out=b) on 2^28 floats -- the elementwise kernel that WRITES 1 GiB per dispatch
        if "elementwise_kernel" in r["Kernel_Name"] and "Fill" not in r["Kernel_Name"] and c == "WRITE_SIZE" and abs(float(r["PerDispatch_KB"]) / float(1 << 20) - 1.0) < 0.25 and "WRITE_SIZE" not in calib:
            calib["WRITE_SIZE"] = (float(r["PerDispatch_KB"]), int(r["Dispatches"]), r["Kernel_Name"])
if "WRITE_SIZE" in calib:   # the same kernel in the FETCH_SIZE pass
    for r in csv.DictReader(open(os.path.join(d, f"{tag}_FETCH_SIZE_summary.csv"))):
        if r["Kernel_Name"] == calib["WRITE_SIZE"][2]:
            calib["FETCH_SIZE"] = (float(r["PerDispatch_KB"]), int(r["Dispatches"]), r["Kernel_Name"])
gib_kb = float(1 << 20)
rf = gib_kb / calib["FETCH_SIZE"][0] if "FETCH_SIZE" in calib else None
wf = gib_kb / calib["WRITE_SIZE"][0] if "WRITE_SIZE" in calib else None
use_rf = rf if rf is not None else 2.0
use_wf = wf if wf is not None else 1.0
rec = {"source_hash": source_hash(), "size": size, "sources": sources, "kernel": kern, "dispatches": disp,
       "fetch_kb_per_launch": vals["FETCH_SIZE"], "write_kb_per_launch": vals["WRITE_SIZE"],
       "calibration": {"kernel": (calib.get("WRITE_SIZE", (None, None, None))[2] or "")[:160] or None, "bytes_read_per_call": 1 << 30, "bytes_written_per_call": 1 << 30,
                       "FETCH_SIZE_kb_reported": calib.get("FETCH_SIZE", (None,))[0], "WRITE_SIZE_kb_reported": calib.get("WRITE_SIZE", (None,))[0],
                       "read_factor_measured": rf, "write_factor_measured": wf},
       "read_factor_used": use_rf, "write_factor_used": use_wf,
       "traffic_bytes_per_launch": (use_rf * vals["FETCH_SIZE"] + use_wf * vals["WRITE_SIZE"]) * 1024.0,
       "note": "counter values x the factors the 1 GiB copy of the same run gives (bytes it moved / bytes reported); without a calibration kernel in "
               "the run: the guide's x2 for FETCH_SIZE on gfx950, x1 for WRITE_SIZE"}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
