"""time of set_slowness (device pointer: the four sheared copies are what is timed) on an n^3 fp32 grid, lines vs scatter kernel;
and that both give the same solve.  usage: set_slowness_time.py n"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, ttcr_amd, cases
n = int(sys.argv[1]); dx = 20.0 / (n - 1); x = np.arange(n) * dx
s = torch.rand(n ** 3, dtype=torch.float32, device='cuda') * 0.5 + 0.5
res = {}
for mode in ('lines', 'scatter'):
    if mode == 'scatter': os.environ['TTCR_FSM_SHEAR_SCATTER'] = '1'
    else: os.environ.pop('TTCR_FSM_SHEAR_SCATTER', None)
    g = ttcr_amd.Grid3d(x, x, x, n_threads=1, cell_slowness=0, method='FSM', tt_from_rp=0, weno=0, dtype=np.float32)
    g.set_slowness_device(s.data_ptr(), s.numel())
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): g.set_slowness_device(s.data_ptr(), s.numel())
    torch.cuda.synchronize(); el = (time.perf_counter() - t) / 5
    src = cases.mt_sources(1); rcv = cases.rcv_lattice3d()
    res[mode] = g.raytrace(np.repeat(src, rcv.shape[0], axis=0), rcv)
    print(f"n={n} {mode}: set_slowness_device {el*1e3:.3f} ms (copy + 4 sheared copies)", flush=True)
assert np.array_equal(res['lines'], res['scatter'])
print("same receiver traveltimes")
