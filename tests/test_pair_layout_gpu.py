"""The field layout of grids in the pairing window follows the model from call to call (option "pair_layout", GridT::choose_layout in
ttcr_amd/csrc/fsm_capi.hip) -- and nothing a caller sees depends on it (-m gpu).  TTCR_FSM_PAIR_UNITS lowers the pairing threshold so that a
small grid sits in the window: 96^3 nodes = 36 patches, 6 slots = 216 units, threshold 100 -> window (100, 250]."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import cases, ttcr_amd
n, S = 96, 6
dx = 20.0 / (n - 1)
x = np.arange(n) * dx
smooth = np.full((n, n, n), 0.4, dtype=np.float32)          # a constant model: the second sweep-iteration evaluates next to nothing
rough = np.random.default_rng(5).uniform(0.25, 1.0, (n, n, n)).astype(np.float32)
srcs = cases.mt_sources(S)
rcv1 = cases.rcv_lattice3d(n=5)
src_rows, rcv_rows = np.repeat(srcs, rcv1.shape[0], axis=0), np.tile(rcv1, (S, 1))
def run(g, model, calls):
    g.set_slowness(model)
    out = []
    for _ in range(calls):
        tt = g.raytrace(src_rows, rcv_rows)
        tm = g.timing()
        out.append(dict(kernel=g.last_kernel(), frac=tm["evaluated_updates"] / max(tm["node_updates"], 1), tt=tt.tolist(), niter=[g.get_niter(i) for i in range(S)],
                        fields=[float(np.sum(g.get_grid_traveltimes(i).astype(np.float64))) for i in range(S)],
                        f0=g.get_grid_traveltimes(3).ravel()[::97].tolist()))
    return out
res = {}
for name, lay in (("auto", -1), ("pairs", 1), ("single", 0)):
    g = ttcr_amd.Grid3d(x, x, x, n_threads=S, cell_slowness=0, method='FSM', tt_from_rp=0, weno=0, dtype=np.float32)
    g.set_option("pair_layout", lay)
    g.set_option("skip", 1)
    res[name] = run(g, smooth, 3) + run(g, rough, 3) + run(g, smooth, 2)
    # a call that restarts ONE slot must leave the other slots' fields where they are, whatever the layout did before
    before = g.get_grid_traveltimes(1).copy()
    g.raytrace(srcs[4:5], rcv1[:1], thread_no=2)
    res[name].append(dict(kept=bool(np.array_equal(before, g.get_grid_traveltimes(1)))))
print("LAYOUT_WORKER " + json.dumps(res))
"""


def test_layout_follows_the_model_and_results_do_not_depend_on_it(tmp_path, capsys):
    script = tmp_path / "layout_worker.py"
    script.write_text(WORKER % dict(root=ROOT))
    # (a 96^3 constant model evaluates 0.59 of its updates, the 512^3 gradient model of the bench 0.41: the test moves the lower threshold
    # to 0.65 so that its small smooth model counts as smooth)
    LO, HI = 0.65, 0.80
    env = dict(os.environ, TTCR_FSM_PAIR_UNITS="100", TTCR_FSM_LAYOUT_LO=str(LO), TTCR_FSM_LAYOUT_HI=str(HI))
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("LAYOUT_WORKER ")][-1][len("LAYOUT_WORKER "):])
    kern = {k: [c.get("kernel") for c in v[:-1]] for k, v in res.items()}
    with capsys.disabled():
        print("\n[pair_layout] kernels call by call (3 x smooth, 3 x rough, 2 x smooth):")
        for k, v in kern.items():
            print("   ", k, [("pairs" if ",1,2,true" in q else "single") for q in v])
    pairs = lambda q: ",1,2,true" in q
    assert all(pairs(q) for q in kern["pairs"]) and not any(pairs(q) for q in kern["single"])
    a = [pairs(q) for q in kern["auto"]]
    f = [c["frac"] for c in res["auto"][:-1]]
    with capsys.disabled():
        print("    evaluated fraction of the auto grid's calls:", [round(v, 3) for v in f])
    # created in pairs; every call that restarts all slots takes the layout the rule gives for the evaluated fraction of the call before:
    # pairs -> one field per workgroup below LO, back to pairs above HI
    assert a[0]
    for i in range(1, 8):
        want = (f[i - 1] >= LO) if a[i - 1] else (f[i - 1] > HI)
        assert a[i] == want, (i, a, f)
    # ... and the two models of this test drive it both ways
    assert not all(a) and any(a[3:6]), (a, f)
    # every call of every layout: the same receiver traveltimes, iteration counts and fields
    for i in range(8):
        ref = res["pairs"][i]
        for k in ("auto", "single"):
            got = res[k][i]
            assert got["tt"] == ref["tt"] and got["niter"] == ref["niter"] and got["fields"] == ref["fields"] and got["f0"] == ref["f0"], (k, i)
    assert all(v[-1]["kept"] for v in res.values())


SKIP_WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import cases, ttcr_amd
n = 64
dx = 20.0 / (n - 1)
x = np.arange(n) * dx
smooth = np.full((n, n, n), 0.4, dtype=np.float32)
rough = np.random.default_rng(5).uniform(0.25, 1.0, (n, n, n)).astype(np.float32)
src = cases.mt_sources(1)
rcv = cases.rcv_lattice3d(n=5)
grids = {}
for name, sk in (("auto", -1), ("off", 0), ("on", 1)):
    g = ttcr_amd.Grid3d(x, x, x, n_threads=1, cell_slowness=0, method='FSM', tt_from_rp=0, weno=0, dtype=np.float32)
    g.set_option("skip", sk)
    grids[name] = g
out = []
for model, calls in ((smooth, 3), (rough, 11), (smooth, 2)):
    for g in grids.values(): g.set_slowness(model)
    for _ in range(calls):
        rec = {}
        for name, g in grids.items():
            tt = g.raytrace(np.repeat(src, rcv.shape[0], axis=0), rcv)
            tm = g.timing()
            rec[name] = dict(kernel=g.last_kernel(), frac=tm["evaluated_updates"] / max(tm["node_updates"], 1), niter=g.get_niter(0),
                             tt=tt.tolist(), s=float(np.sum(g.get_grid_traveltimes(0).astype(np.float64))))
        out.append(rec)
print("SKIP_WORKER " + json.dumps(out))
"""


def test_exact_skipping_follows_the_model_in_the_probe_window(tmp_path, capsys):
    """launches of skip_probe_min ... skip_units_min - 1 work units (fp32, 3-D, first order): exact skipping stays on while the grid's last
    skipping solve evaluated less than 0.6 of its node updates, goes off above, and is tried again after eight solves without
    (GridT::skip_default) -- and no result depends on it.  The two thresholds are lowered so that a 64^3 grid (16 patches) sits in the window."""
    script = tmp_path / "skip_worker.py"
    script.write_text(SKIP_WORKER % dict(root=ROOT))
    env = dict(os.environ, TTCR_FSM_SKIP_PROBE_UNITS="8", TTCR_FSM_SKIP_UNITS="32")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("SKIP_WORKER ")][-1][len("SKIP_WORKER "):])
    skipping = lambda k: k.split(",")[5] == "true"
    on = [skipping(c["auto"]["kernel"]) for c in out]
    f = [c["auto"]["frac"] for c in out]
    with capsys.disabled():
        print("\n[skip probe] auto grid, call by call (3 x constant, 11 x rough, 2 x constant): skipping", ["on" if v else "off" for v in on])
        print("    evaluated fraction:", [round(v, 3) for v in f])
    assert all(skipping(c["on"]["kernel"]) for c in out) and not any(skipping(c["off"]["kernel"]) for c in out)
    # the rule, replayed
    want, off_calls = True, 0
    for i, c in enumerate(out):
        assert on[i] == want, (i, on, f)
        if on[i]:
            want, off_calls = f[i] < 0.6, 0
        else:
            off_calls += 1
            if off_calls >= 8:
                want, off_calls = True, 0
    assert not all(on) and on[0] and any(on[4:14])   # the rough model switches it off, the probe after eight calls switches it on once
    for c in out:
        for k in ("off", "on"):
            assert c["auto"]["tt"] == c[k]["tt"] and c["auto"]["niter"] == c[k]["niter"] and c["auto"]["s"] == c[k]["s"]
