"""Does it pay to pair spatially close sources (a chunk of a pair is evaluated when EITHER source needs it)?
512^3 x 64 sources of the bench, skipping on, the library's own pairing switched off (option pair_sources = 0) and the call
order permuted instead: as given / greedy nearest neighbour (what the library does) / the same improved by 2-opt swaps.
usage: pairing_probe.py [n] [S]"""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512; S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dx = 20.0 / (n - 1); x = np.arange(n) * dx
s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * x))[None, None, :], (n, n, n))).astype(np.float32)
src = cases.mt_sources(S)[:S]
def greedy_pairs(p):
    left = list(range(len(p))); order = []
    while left:
        a = left.pop(0)
        if not left: order.append(a); break
        d = [np.sum((p[a] - p[b]) ** 2) for b in left]
        b = left.pop(int(np.argmin(d)))
        order += [a, b]
    return np.array(order)
def two_opt(p, order, power=1.0):
    pairs = [list(order[k:k + 2]) for k in range(0, len(order) - 1, 2)]
    d = lambda a, b: np.sum((p[a] - p[b]) ** 2) ** (power / 2)
    improved = True
    while improved:
        improved = False
        for i, j in itertools.combinations(range(len(pairs)), 2):
            (a, b), (c, e) = pairs[i], pairs[j]
            cur = d(a, b) + d(c, e)
            alt = [((a, c), (b, e)), ((a, e), (b, c))]
            for (p1, p2) in alt:
                if d(*p1) + d(*p2) < cur - 1e-12:
                    pairs[i], pairs[j] = list(p1), list(p2); improved = True; break
    out = [v for pr in pairs for v in pr]
    if len(order) % 2: out.append(order[-1])
    return np.array(out)
g = ttcr_amd.Grid3d(x, x, x, n_threads=S, cell_slowness=0, method='FSM', tt_from_rp=0, weno=0, dtype=np.float32)
g.set_slowness(s); g.set_option('pair_sources', 0)
rcv = np.zeros((S, 3))
p = src[:, -3:]
gr = greedy_pairs(p)
for name, o in (("call order", np.arange(S)), ("greedy nearest neighbour", gr), ("greedy + 2-opt, sum of distances", two_opt(p, gr, 1.0)),
                ("greedy + 2-opt, sum of squared distances", two_opt(p, gr, 2.0)), ("greedy + 2-opt, max-norm distance", None)):
    if o is None:
        pm = p.copy()
        pairs = [list(gr[k:k + 2]) for k in range(0, S - 1, 2)]
        dd = lambda a, b: np.max(np.abs(pm[a] - pm[b]))
        improved = True
        while improved:
            improved = False
            for i, j in itertools.combinations(range(len(pairs)), 2):
                (a, b), (c, e) = pairs[i], pairs[j]
                cur = dd(a, b) + dd(c, e)
                for (p1, p2) in (((a, c), (b, e)), ((a, e), (b, c))):
                    if dd(*p1) + dd(*p2) < cur - 1e-12:
                        pairs[i], pairs[j] = list(p1), list(p2); improved = True; break
        o = np.array([v for pr in pairs for v in pr])
    best = None
    for r in range(2):
        g.raytrace(src[o], rcv); tm = g.timing()
        if best is None or tm['sweep_ms'] < best['sweep_ms']: best = tm
    tot = sum(np.sqrt(np.sum((p[o[k]] - p[o[k + 1]]) ** 2)) for k in range(0, S - 1, 2))
    print(f"{name:42s}: sweeps {best['sweep_ms']:.2f} ms, evaluated {best['evaluated_updates'] / max(best['node_updates'], 1):.4f}, sum of pair distances {tot:.1f}", flush=True)
