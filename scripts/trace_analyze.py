"""Unit trace of an FSM_ENABLE_PROF build (TTCR_FSM_PROF=1 TTCR_FSM_PROF_TRACE=file): per work unit
(direction, patch, batch entry) the entry / first-chunk / exit times on the 100 MHz clock.
Prints per direction: first entry, first chunk, last exit; the ramp (first-chunk time against the patch
anti-diagonal); unit run times; and the number of units inside their chunk loop over time."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64)[:4 * 65536].reshape(-1, 4)   # (the chunk records of the first units follow)
a = a[a[:, 2] > 0]
t0 = a[:, 0].min()
ent = (a[:, 0] - t0) * 0.01   # us
beg = (a[:, 1] - t0) * 0.01
end = (a[:, 2] - t0) * 0.01
TJ = (a[:, 3] & 0xffff).astype(int); TK = ((a[:, 3] >> 16) & 0xffff).astype(int)
d = ((a[:, 3] >> 32) & 0xff).astype(int); z = ((a[:, 3] >> 40) & 0xff).astype(int); nch = (a[:, 3] >> 48).astype(int)
print(f"units {len(a)}  span {end.max():.1f} us")
for dd in sorted(set(d)):
    m = d == dd
    run = end[m] - beg[m]
    wait = beg[m] - ent[m]
    md = TJ[m] + TK[m]
    # ramp: median first-chunk time per anti-diagonal, linear fit
    ds = sorted(set(md))
    med = np.array([np.median(beg[m][md == q]) for q in ds])
    slope = np.polyfit(ds, med, 1)[0] if len(ds) > 2 else 0.0
    print(f"dir {dd}: units {m.sum()} shortcut {(nch[m]==0xffff).sum()} no-chunk {(nch[m]==0).sum()} chunks/evaluating unit {np.mean(nch[m][(nch[m]>0)&(nch[m]<0xffff)]) if ((nch[m]>0)&(nch[m]<0xffff)).any() else 0:.1f} | resident: median {np.median(end[m]-ent[m]):.1f} us")
    print(f"dir {dd}: entry {ent[m].min():9.1f}  first chunk {beg[m].min():9.1f}  last exit {end[m].max():9.1f} | "
          f"ramp {slope:6.2f} us/diagonal | run: median {np.median(run):7.1f} min {run.min():7.1f} max {run.max():7.1f} | "
          f"wait before start: median {np.median(wait):8.1f} max {wait.max():8.1f}")
# concurrency
T = np.linspace(0, end.max(), 41)
act = [(int(np.sum((beg <= t) & (end > t))), int(np.sum((ent <= t) & (beg > t)))) for t in T]
print("time(us): running / resident-but-waiting-for-previous-sweep")
print("  ".join(f"{t:.0f}:{r}/{w}" for t, (r, w) in zip(T, act)))
# slot time by kind of unit (SKIP kernels): units that evaluated no chunk at all only relay progress
res = end - ent
none = (nch == 0) | (nch == 0xffff)
some = ~none
print(f"slot time: all units {res.sum()/1e3:.1f} ms | units without an evaluated chunk: {none.sum()} units, {res[none].sum()/1e3:.1f} ms "
      f"(median {np.median(res[none]) if none.any() else 0:.1f} us, 90 % {np.percentile(res[none], 90) if none.any() else 0:.1f} us) | "
      f"units with chunks: {some.sum()} units, {res[some].sum()/1e3:.1f} ms, {nch[some].sum()} chunks = {res[some].sum()/max(nch[some].sum(),1):.2f} us of residence per evaluated chunk")
