"""profiles/rNN/traffic.json from the two PMC summaries of scripts/pmc_run.sh (FETCH_SIZE / WRITE_SIZE per dispatch of the
sweep kernel), tagged with the hash of the kernel sources (ttcr_amd.build.source_hash) so that bench.py only reports it
for the library it was measured with.  usage: pmc_to_json.py <pmc_dir> <tag> <size> <sources> <out.json>"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ttcr_amd.build import source_hash

d, tag, size, sources, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
vals = {}
calib = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(os.path.join(d, f"{tag}_{c}_summary.csv"))):
        if "fsm_sweep_persistent" in r["Kernel_Name"] and c not in vals:
            vals[c] = float(r["PerDispatch_KB"]); kern = r["Kernel_Name"]
        if "fsm_shear_slowness" in r["Kernel_Name"]:
            calib[c] = float(r["PerDispatch_KB"])
rec = {"source_hash": source_hash(), "size": size, "sources": sources, "kernel": kern,
       "fetch_kb_per_launch": vals["FETCH_SIZE"], "write_kb_per_launch": vals["WRITE_SIZE"],
       "calibration": {"kernel": "fsm_shear_slowness", "bytes_read_per_call": size ** 3 * 4, "FETCH_SIZE_kb_reported": calib.get("FETCH_SIZE"),
                       "bytes_written_per_call": size ** 3 * 4, "WRITE_SIZE_kb_reported": calib.get("WRITE_SIZE")},
       "traffic_bytes_per_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
       "note": "gfx950: FETCH_SIZE counts 128-B requests at 64 B (x2 correction, see the calibration kernel of the same run)"}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
