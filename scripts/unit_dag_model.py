"""Model of one sweep-iteration of a lone 3-D source in the persistent kernel: the (sweep, patch) work units with
their chunk-level dependencies (upwind patches of the same sweep, the 3 x 3 patches of the previous sweep), admitted to a fixed
number of workgroup slots in ticket order.  Inputs are measured quantities (time of a chunk, hand-off latency, unit set-up);
the output is the makespan of the schedule -- how much of the lone source's time is the dependency structure itself.
usage: unit_dag_model.py [n=512] [slots=512] [t_chunk_us=4.1] [handoff_us=1.5] [setup_us=6] [order=time|sweep]"""
import heapq, sys
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 512
tc = float(sys.argv[3]) if len(sys.argv) > 3 else 4.1
lam = float(sys.argv[4]) if len(sys.argv) > 4 else 1.5
setup = float(sys.argv[5]) if len(sys.argv) > 5 else 6.0
order = sys.argv[6] if len(sys.argv) > 6 else "time"
P, C = 16, 8
npj = npk = (n + P - 1) // P
NF = n


def unit_levels(TJ, TK):
    j0, k0 = TJ * P, TK * P
    jm, km = min(j0 + P, n) - 1, min(k0 + P, n) - 1
    Ls, Le = j0 + k0, jm + km + NF - 1
    m = TJ + TK
    Lcf = Ls - ((Ls - m) % C)
    return Lcf, (Le - Lcf) // C + 1


def flips(d):
    return (d >> 1) & 1, (d >> 2) & 1


# ticket order: sweep by sweep (anti-diagonals), or by the start time on an unbounded machine (what the library does for a lone source)
units = []
for d in range(8):
    for m in range(npj + npk - 1):
        for TK in range(max(0, m - npj + 1), min(m, npk - 1) + 1):
            units.append((d, m - TK, TK))
if order == "time":
    ts = {}
    tf = {}
    hop = (P + C - 1) / C + 1.0
    for (d, TJ, TK) in units:
        t = 0.0
        if TJ > 0: t = max(t, ts[(d, TJ - 1, TK)] + hop)
        if TK > 0: t = max(t, ts[(d, TJ, TK - 1)] + hop)
        if d > 0:
            rj, rk = flips(d); prj, prk = flips(d - 1)
            for a in (-1, 0, 1):
                for b in (-1, 0, 1):
                    tj, tk = TJ + a, TK + b
                    if 0 <= tj < npj and 0 <= tk < npk:
                        pj = npj - 1 - tj if rj != prj else tj
                        pk = npk - 1 - tk if rk != prk else tk
                        t = max(t, tf[(d - 1, pj, pk)])
        ts[(d, TJ, TK)] = t
        tf[(d, TJ, TK)] = t + unit_levels(TJ, TK)[1] + 1.0
    units.sort(key=lambda u: ts[u])   # (stable: ties keep sweep / anti-diagonal order)
if order == "tail":
    # longest path from a unit's start to the end of the iteration (in chunk times): the units with the longest tail first
    tail = {}
    hop = (P + C - 1) / C + 1.0
    succ_next = {}
    for d in range(1, 8):
        rj, rk = flips(d); prj, prk = flips(d - 1)
        for TJ in range(npj):
            for TK in range(npk):
                for a in (-1, 0, 1):
                    for b in (-1, 0, 1):
                        tj, tk = TJ + a, TK + b
                        if 0 <= tj < npj and 0 <= tk < npk:
                            pj = npj - 1 - tj if rj != prj else tj
                            pk = npk - 1 - tk if rk != prk else tk
                            succ_next.setdefault((d - 1, pj, pk), []).append((d, TJ, TK))
    for (d, TJ, TK) in reversed(units):
        dur = unit_levels(TJ, TK)[1] + 1.0
        t = dur
        if TJ + 1 < npj: t = max(t, hop + tail[(d, TJ + 1, TK)])
        if TK + 1 < npk: t = max(t, hop + tail[(d, TJ, TK + 1)])
        for v in succ_next.get((d, TJ, TK), ()): t = max(t, dur + tail[v])
        tail[(d, TJ, TK)] = t
    units.sort(key=lambda u: -tail[u])

iters = int(sys.argv[7]) if len(sys.argv) > 7 else 1   # > 1: re-sort the tickets by the start times of the previous schedule and repeat
for it in range(iters):
  fin = {}       # unit -> finish times of its chunks
  lcf = {}
  first = {}
  free = [0.0] * slots
  heapq.heapify(free)
  admit_prev = 0.0
  busy = 0.0
  for (d, TJ, TK) in units:
      Lcf, nch = unit_levels(TJ, TK)
      a = max(heapq.heappop(free), admit_prev)      # tickets are taken in order
      admit_prev = a
      t0 = a + setup
      if d > 0:                                      # previous sweep: the 3 x 3 patches around this one (its own partition) finished
          rj, rk = flips(d); prj, prk = flips(d - 1)
          for da in (-1, 0, 1):
              for db in (-1, 0, 1):
                  tj, tk = TJ + da, TK + db
                  if 0 <= tj < npj and 0 <= tk < npk:
                      pj = npj - 1 - tj if rj != prj else tj
                      pk = npk - 1 - tk if rk != prk else tk
                      t0 = max(t0, fin[(d - 1, pj, pk)][-1] + lam)
      dep = np.full(nch, t0)
      for up in ((TJ - 1, TK), (TJ, TK - 1)):
          if up[0] < 0 or up[1] < 0: continue
          f = fin[(d, up[0], up[1])]
          ul = lcf[(d, up[0], up[1])]
          # chunk c (levels from Lcf + 8 c) needs the upwind chunk that ends at level Lcf + 8 c + 7
          cu = (Lcf + C * np.arange(nch) - 1 - ul) // C
          cu = np.clip(cu, 0, len(f) - 1)
          dep = np.maximum(dep, f[cu] + lam)
      c = np.arange(nch)
      start = np.maximum.accumulate(dep - c * tc) + c * tc     # start[c] = max(start[c-1] + tc, dep[c])
      f = start + tc
      fin[(d, TJ, TK)] = f
      first[(d, TJ, TK)] = start[0]
      lcf[(d, TJ, TK)] = Lcf
      busy += nch * tc
      heapq.heappush(free, f[-1])
      # (units of sweep d are only needed by sweeps d and d + 1)
  end = max(v[-1] for v in fin.values())
  print(f'  pass {it}: makespan {end/1e3:.3f} ms')
  units.sort(key=lambda u: first[u])
end = max(v[-1] for v in fin.values())
print(f"n={n} slots={slots} chunk {tc} us hand-off {lam} us set-up {setup} us order={order}: makespan {end/1e3:.3f} ms per sweep-iteration; "
      f"work {busy/1e3:.1f} ms of slot time = {busy/slots/1e3:.3f} ms if nothing waited; critical path only (unbounded slots): run with slots=100000")
