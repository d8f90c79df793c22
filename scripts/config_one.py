"""One configuration per process (for rocprofv3): config_one.py <C2|C4|C5|W1|W8|S1|S1d|E8> [reps]
C2: 256^3 nodes gradient, 1 source, fp32;  C4: 256^3 cells layers, 8 sources;  C5: Grid2d 4096^2, 16 sources;
W1 / W8: the ttcrpy default (WENO) on 256^3 nodes with 1 / 8 sources;  S1 / E8: 512^3, 1 / 8 sources (S1: every chunk evaluated, option skip = 0; S1d: the library's default path).  Prints sweep ms and the roofline
fraction.  TTCR_FSM_ARITH=1 in the environment: the tolerance-grade arithmetic.  Every process first copies 1 GiB with torch (three times): the
calibration kernel of the HBM counters in the same rocprofv3 run (a coalesced 16-byte-per-lane copy of known size)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
cfg = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
import torch
_a = torch.ones(1 << 28, dtype=torch.float32, device='cuda'); _b = torch.empty_like(_a)
for _ in range(3): torch.add(_a, 1.0, out=_b)   # (an elementwise kernel, 16 bytes per lane: Tensor.copy_ of a contiguous tensor is a runtime copy instead)
torch.cuda.synchronize(); del _a, _b; torch.cuda.empty_cache()
rc3 = cases.rcv_lattice3d()
if cfg in ('C2', 'W1', 'W8', 'S1', 'S1d', 'E8'):
    n = 512 if cfg in ('S1', 'S1d', 'E8') else 256; dx = 20.0 / (n - 1); x = np.arange(n) * dx
    s = np.ascontiguousarray(np.broadcast_to((1 / (1 + 0.1 * x))[None, None, :], (n, n, n)), dtype=np.float32)
    ns = 8 if cfg in ('W8', 'E8') else 1
    g = ttcr_amd.Grid3d(x, x, x, n_threads=ns, cell_slowness=0, method='FSM', tt_from_rp=0, weno=int(cfg[0] == 'W'), dtype=np.float32)
    g.set_slowness(s)
    if cfg == 'S1': g.set_option('skip', 0)   # the evaluate-all kernel the roofline figures are about (S1d: the library's own choice)
    srcs = cases.mt_sources(64)[:ns]; nodes, bpn = n ** 3, 104.0
    src, rcv = np.repeat(srcs, len(rc3), axis=0), np.tile(rc3, (ns, 1))
elif cfg == 'C4':
    nc = 256; dx = 20.0 / nc; x = np.arange(nc + 1) * dx
    sc = np.ascontiguousarray(np.broadcast_to((1 / (1 + 0.1 * (np.floor(np.arange(nc) * dx) + 0.5)))[None, None, :], (nc, nc, nc)), dtype=np.float32)
    ns = 8
    g = ttcr_amd.Grid3d(x, x, x, n_threads=ns, cell_slowness=1, method='FSM', tt_from_rp=0, weno=0, dtype=np.float32)
    g.set_slowness(sc)
    srcs = cases.mt_sources(64)[:ns]; nodes, bpn = 257 ** 3, 104.0
    src, rcv = np.repeat(srcs, len(rc3), axis=0), np.tile(rc3, (ns, 1))
elif cfg == 'C5':
    n = 4096; dx = 20.0 / (n - 1); x = np.arange(n) * dx
    s2 = np.ascontiguousarray(np.broadcast_to((1 / (1 + 0.1 * x))[None, :], (n, n)), dtype=np.float32)
    ns = 16
    g = ttcr_amd.Grid2d(x, x, n_threads=ns, cell_slowness=0, method='FSM', weno=0, dtype=np.float32)
    g.set_slowness(s2)
    srcs = cases.mt_sources(16, ndim=2); rc = np.stack([np.zeros(21), np.linspace(0, 20, 21)], axis=1); nodes, bpn = n * n, 56.0
    src, rcv = np.repeat(srcs, len(rc), axis=0), np.tile(rc, (ns, 1))
else:
    raise SystemExit("unknown configuration")
best = None
for _ in range(reps):
    g.raytrace(src, rcv); tm = g.timing()
    if best is None or tm['sweep_ms'] < best['sweep_ms']: best = tm
it = sum(g.get_niter(i) + g.get_niterw(i) for i in range(ns))
ev = best['evaluated_updates'] / max(best['node_updates'], 1)
print(f"{cfg}: sweeps {best['sweep_ms']:.3f} ms, launches {best['kernel_launches']}, sweep-iterations (all sources) {it}, evaluated {ev:.3f}, "
      f"{nodes * it / best['sweep_ms'] / 1e3:.0f} Mnodes/s per sweep-iteration, contract fraction {bpn * nodes * it / best['sweep_ms'] / 1e6 / 8000:.4f}, "
      f"evaluated-priced fraction {ev * bpn * nodes * it / best['sweep_ms'] / 1e6 / 8000:.4f} of 8 TB/s", flush=True)
