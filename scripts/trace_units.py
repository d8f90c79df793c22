"""Unit trace of a trace-only build (scripts/devbuild.sh trace -DFSM_ENABLE_PROF=2; run with TTCR_FSM_PROF=1
TTCR_FSM_PROF_TRACE=file): one record per work unit of the LAST sweep-iteration launch (up to 2^20 units): entry, entry ->
first chunk, ticks spent polling upwind progress, exit (100 MHz clock), patch, direction, batch entry, evaluated chunks.
Prints where the slot time of the launch goes.   usage: trace_units.py file [slots]"""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
a = a[a[:, 2] > 0]
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 768
t0 = a[:, 0].min()
ent = (a[:, 0] - t0) * 0.01                          # us
pre = (a[:, 1] & np.uint64(0xffffffff)) * 0.01       # entry -> first chunk (ticket, tables, previous sweep's patches)
poll = (a[:, 1] >> np.uint64(32)) * 0.01             # thread 0 polling upwind progress inside the chunk loop
end = (a[:, 2] - t0) * 0.01
d = ((a[:, 3] >> np.uint64(32)) & np.uint64(0xff)).astype(int)
z = ((a[:, 3] >> np.uint64(40)) & np.uint64(0xff)).astype(int)
nch = (a[:, 3] >> np.uint64(48)).astype(int)
res = end - ent
span = end.max()
short = nch == 0xffff
none = nch == 0
some = ~short & ~none
print(f"units {len(a)}  span {span/1e3:.2f} ms  slot time {res.sum()/1e3:.1f} ms = {res.sum()/span:.0f} slots busy on average (of {slots})")
print(f"  whole-sweep shortcut: {short.sum()} units, {res[short].sum()/1e3:.1f} ms of slot time (median {np.median(res[short]) if short.any() else 0:.2f} us)")
print(f"  no chunk evaluated:   {none.sum()} units, {res[none].sum()/1e3:.1f} ms (median {np.median(res[none]) if none.any() else 0:.2f} us, 90 % {np.percentile(res[none], 90) if none.any() else 0:.2f}); "
      f"of it before the chunk loop {pre[none].sum()/1e3:.1f} ms, polling {poll[none].sum()/1e3:.1f} ms")
if some.any():
    run = res[some] - pre[some] - poll[some]
    print(f"  units with chunks:    {some.sum()} units, {res[some].sum()/1e3:.1f} ms: before the chunk loop {pre[some].sum()/1e3:.1f} ms, polling {poll[some].sum()/1e3:.1f} ms, "
          f"rest {run.sum()/1e3:.1f} ms = {run.sum()/nch[some].sum():.2f} us per evaluated chunk ({nch[some].sum()} chunks, {nch[some].mean():.1f} per unit)")
print("per direction: units with chunks / evaluated chunks / first entry / last exit (ms) / slot time (ms) of which polling, before-loop")
for dd in sorted(set(d)):
    m = d == dd
    print(f"  dir {dd}: {np.sum(m & some):6d} {nch[m & some].sum():8d}  {ent[m].min()/1e3:8.2f} {end[m].max()/1e3:8.2f}  {res[m].sum()/1e3:8.1f} {poll[m].sum()/1e3:8.1f} {pre[m].sum()/1e3:8.1f}")
T = np.linspace(0, span, 33)[:-1]
order = np.argsort(ent)
print("time (ms): resident units | evaluating share of the slot time in the window")
w = span / 32
for t in T:
    inw = (ent < t + w) & (end > t)
    ov = np.minimum(end[inw], t + w) - np.maximum(ent[inw], t)
    ovs = ov[some[inw]].sum()
    print(f"  {t/1e3:7.2f}: {ov.sum()/w:6.0f} | {ovs/max(ov.sum(),1e-9):.2f}")
