"""Per-chunk timeline of a chain of patches from an FSM_ENABLE_PROF trace (TTCR_FSM_PROF_TRACE file): for direction D,
the patches (TJ, TK) = (m, 0) for m = 0.. (the J ramp), the five stamps of their first chunks:
top (loop entry), wait (upwind counters seen + barrier), stage, march, write-back -- all relative to the sweep's start.
usage: chunk_trace.py trace.bin [dir] [n_patches] [n_chunks]"""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64)
D = int(sys.argv[2]) if len(sys.argv) > 2 else 4
NP = int(sys.argv[3]) if len(sys.argv) > 3 else 6
NCH = int(sys.argv[4]) if len(sys.argv) > 4 else 6
units = raw[:4 * 65536].reshape(-1, 4)
rec = raw[4 * 65536:4 * 65536 + 8192 * 400].reshape(8192, 80, 5)
TJ = (units[:, 3] & 0xffff).astype(int); TK = ((units[:, 3] >> 16) & 0xffff).astype(int); d = ((units[:, 3] >> 32) & 0xff).astype(int)
ok = units[:, 2] > 0
sel = np.nonzero(ok & (d == D))[0]
t0 = units[sel, 0].min()
us = lambda v: (float(v) - float(t0)) * 0.01
for axis in ("J", "K", "diag"):
    print(f"--- direction {D}, ramp along {axis}")
    for m in range(NP):
        want = (m, 0) if axis == "J" else ((0, m) if axis == "K" else (m, m))
        hit = [i for i in sel if (TJ[i], TK[i]) == want and i < 8192]
        if not hit:
            continue
        i = hit[0]
        line = f"patch {want}: entry {us(units[i,0]):8.1f} loop {us(units[i,1]):8.1f} | "
        for c in range(NCH):
            r = rec[i, c]
            if r[3] == 0:
                break
            wb = rec[i, c + 1][4] if c + 1 < 80 else 0   # (the chunk counter moves before the write-back stamp)
            line += f"c{c}: top {us(r[0]):7.1f} wait {us(r[1]):7.1f} stage {us(r[2]):7.1f} march {us(r[3]):7.1f} wb {us(wb):7.1f} | "
        print(line)
