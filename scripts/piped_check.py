#!/usr/bin/env python
"""Pipelined sweep kernel (fsm_piped_kernels.h) against the four-wave kernel: fields, iteration counts, change history on random
shapes / models / sources (bit-identical), then the time of a lone 512^3 and 256^3 source with either kernel.
  python scripts/piped_check.py [--no-time] [--cases N] [--seed S]     """
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def solve(shape, s, src, slab, n_threads=1, weno=0, fixed=0):
    import ttcr_amd
    nx, ny, nz = shape
    dx = 0.37
    g = ttcr_amd.Grid3d(np.arange(nx) * dx, np.arange(ny) * dx, np.arange(nz) * dx, n_threads=n_threads, cell_slowness=0,
                        method="FSM", tt_from_rp=0, weno=weno, dtype=np.float32)
    g.set_option("piped", slab); g.set_option("skip", 0)
    if fixed: g.set_option("fixed_iters", fixed)
    g.set_slowness(s)
    rcv = np.array([[0.0, 0.0, 0.0]])
    out = []
    g.raytrace(np.repeat(src, 1, axis=0), np.tile(rcv, (src.shape[0], 1)))
    for i in range(min(n_threads, src.shape[0])):
        out.append((g.get_grid_traveltimes(i).copy(), g.get_niter(i), tuple(g.get_changes(i)[0])))
    return out, g.timing(), g.last_kernel()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-time", action="store_true")
    ap.add_argument("--cases", type=int, default=24)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad = 0
    for c in range(args.cases):
        nx = int(rng.choice([5, 16, 23, 32, 40, 64, 71, 128]))
        ny, nz = (int(v) for v in rng.integers(5, 150, 2)) if c % 3 else (int(v) for v in rng.choice([8, 16, 17, 63, 64, 65, 128, 129], 2))
        shape = (nx, ny, nz)
        kind = c % 4
        if kind == 0: s = np.full(shape, 0.4, np.float32)
        elif kind == 1: s = rng.uniform(0.25, 1.0, shape).astype(np.float32)
        elif kind == 2: s = np.broadcast_to((1.0 / (1.0 + 0.1 * np.arange(shape[2]) * 0.37)).astype(np.float32), shape).copy()
        else:
            b = rng.uniform(0.25, 1.0, tuple((v + 7) // 8 for v in shape)).astype(np.float32)
            s = np.repeat(np.repeat(np.repeat(b, 8, 0), 8, 1), 8, 2)[:shape[0], :shape[1], :shape[2]].copy()
        nsrc = 1 if c % 5 else 3
        hi = (np.array(shape) - 1) * 0.37
        src = rng.uniform(0, 1, (nsrc, 3)) * hi
        if c % 7 == 0: src[0] = np.round(src[0] / 0.37) * 0.37     # on a node
        if c % 11 == 3: src[0] = rng.integers(0, 2, 3) * (np.array(shape) - 2) * 0.37       # in / next to a corner
        weno = 1 if c % 6 == 5 else 0
        ref, _, k0 = solve(shape, s, src, 0, n_threads=nsrc, weno=weno)
        got, _, k1 = solve(shape, s, src, 1, n_threads=nsrc, weno=weno)
        ok = all(np.array_equal(a[0], b[0]) and a[1] == b[1] for a, b in zip(ref, got))
        chg = all(len(a[2]) == len(b[2]) and np.allclose(a[2], b[2], rtol=1e-6) for a, b in zip(ref, got))
        print(f"case {c}: shape {shape} model {kind} sources {nsrc} weno {weno} niter {[a[1] for a in ref]} [{k1}] -> {'ok' if ok else 'FIELDS DIFFER'}"
              f"{'' if chg else ' (change history differs: %s vs %s)' % (ref[0][2], got[0][2])}", flush=True)
        if "piped" not in k1 and not weno:   # (weno: the last kernel launched is the WENO stage's)
            print("   (the pipelined kernel did not run)"); bad += 1
        if not ok:
            bad += 1
            for a, b in zip(ref, got):
                d = np.argwhere(a[0] != b[0])
                if len(d):
                    print("   first differing nodes:", d[:5].tolist(), "of", len(d), " values", [(float(a[0][tuple(i)]), float(b[0][tuple(i)])) for i in d[:3]], "niter", a[1], b[1])
    print("cases with differences:", bad)
    if not args.no_time and not bad:
        import cases
        for n in (256, 512):
            z = np.arange(n) * (20.0 / (n - 1))
            s = np.ascontiguousarray(np.broadcast_to((1.0 / (1.0 + 0.1 * z)).astype(np.float32), (n, n, n)))
            src = cases.mt_sources(1) * (n - 1) * 0.37 / 20.0
            res = {}
            for slab in (0, 1):
                best = None
                for rep in range(4):
                    out, tm, kn = solve((n, n, n), s, src, slab, fixed=2)
                    ms = tm["sweep_ms"] / 2
                    best = ms if best is None else min(best, ms)
                res[slab] = out
                print(f"n={n} slab={slab} [{kn}]: {best:.3f} ms per sweep-iteration ({104.0 * n ** 3 / (best * 1e-3) / 8e12:.3f} of the roofline)", flush=True)
            same = np.array_equal(res[0][0][0], res[1][0][0])
            print(f"n={n}: fields {'identical' if same else 'DIFFER'}", flush=True)
            bad += 0 if same else 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
