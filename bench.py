#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X FSM eikonal solver.

Metric (BASELINE.json): Mnodes/s per sweep-iteration on a 512^3 fp32 grid, plus sources/s.
Workload = config[2] of BASELINE.json at every N (strong scaling): the 64 sources drawn by the
reference's generator (mt19937_64(12345), tests/accuracy_grid3d.cpp:352-360) on the 512^3-node
gradient model s = 1/(1 + 0.1 z) over [0,20]^3 km, fp32, first-order FSM (weno=False,
tt_from_rp=False, eps=1e-5, maxit=50), 441 receivers (rcv.dat lattice), block-distributed over
the N GPUs like get_blk_size (64 on one GPU, 8 per GPU on eight).  One "step" = one full solve
(init + sweep iterations to convergence + receiver interpolation) of the rank's sources.  Inputs are
resident in HBM before the timed region (slowness is broadcast over RCCL and handed to the
solver as a device pointer); the receiver traveltimes are all-gathered over RCCL inside
every step (a few KB).

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_NODE_ITER = 104.0  # 8 sweeps x 12 B + 8 B snapshot, fp32


def gradient_slowness_f32(n, dx):
    z = np.arange(n, dtype=np.float64) * dx
    s = (1.0 / (1.0 + 0.1 * z)).astype(np.float32)
    return s


def cpu_baseline(n=256, n_src=3):
    """The CPU restatement (oracle/, bit-exact vs the compiled reference) timed on this box's
    host cores -- ONE core, like the reference's serial per-source sweep -- on a bounded sample
    of the same workload: the same gradient model and source generator at n^3 nodes."""
    import cases
    from oracle import oracle as O

    dx = 20.0 / (n - 1)
    s = np.repeat(gradient_slowness_f32(n, dx), n * n)
    srcs = cases.mt_sources(64)[:n_src]
    t = time.perf_counter()
    iters = 0
    for p in srcs:
        r = O.solve3d(np.float32, (n - 1,) * 3, dx, (0, 0, 0), s, [p])
        iters += r["niter"]
    el = time.perf_counter() - t
    out = {"value": round(n ** 3 * iters / el / 1e6, 3), "unit": "Mnodes/s per sweep-iteration", "cores": 1,
           "kind": "port",
           "sample": f"{n_src} sources (first of the mt19937_64(12345) set) on the {n}^3-node gradient model, fp32, "
                     f"{iters} sweep-iterations, {el:.1f} s on 1 of {os.cpu_count()} host cores"}
    # (ii) source-parallel on the host cores, one source per thread like the reference's n_threads
    #      pool (ttcr/Grid3D.h:821-832); ctypes releases the GIL for the duration of a solve
    try:
        from concurrent.futures import ThreadPoolExecutor
        nthr = max(1, os.cpu_count() or 1)   # ALL host cores, one source per thread
        many = cases.mt_sources(max(64, nthr))[:nthr]
        t = time.perf_counter()
        with ThreadPoolExecutor(max_workers=nthr) as ex:
            its = list(ex.map(lambda p: O.solve3d(np.float32, (n - 1,) * 3, dx, (0, 0, 0), s, [p])["niter"], many))
        el = time.perf_counter() - t
        out["all_cores_value"] = round(n ** 3 * sum(its) / el / 1e6, 3)
        out["all_cores"] = nthr
        out["all_cores_sample"] = f"{nthr} sources, one per thread, {n}^3 nodes, {sum(its)} sweep-iterations in {el:.1f} s"
    except Exception as e:
        out["all_cores_error"] = str(e)[:200]
    # the unmodified compiled reference, when its build travelled with the tree, on the SAME sample as the restatement
    # (the first source on the n^3-node model): SURVEY 8(d)'s calibration ratio is taken on one sample, one core each
    try:
        if O.have_ref():
            t = time.perf_counter()
            r1 = O.solve3d(np.float32, (n - 1,) * 3, dx, (0, 0, 0), s, [srcs[0]])
            el_port = time.perf_counter() - t
            t = time.perf_counter()
            r = O.ref_solve3d(np.float32, (n - 1,) * 3, dx, (0, 0, 0), s, [srcs[0]])
            el = time.perf_counter() - t
            out["reference_value"] = round(n ** 3 * r["niter"] / el / 1e6, 3)
            out["reference_sample"] = (f"unmodified reference (oracle/_ref), source 0 on the same {n}^3-node model, {r['niter']} sweep-iterations, "
                                       f"{el:.1f} s on 1 core (grid construction included, as in ttcrpy)")
            out["restatement_same_sample_value"] = round(n ** 3 * r1["niter"] / el_port / 1e6, 3)
            out["calibration_ratio_restatement_over_reference"] = round(el / el_port, 2)
    except Exception as e:  # the baseline must never take the bench down
        out["reference_error"] = str(e)[:200]
    return out


PROFILE_DIRS = ("r06", "r05", "r04", "r03", "r02")


def profiled_traffic(n, n_src_rank0, world):
    """HBM bytes per launch (one sweep-iteration of the batch) from the committed rocprofv3 PMC passes of this very command
    (profiles/rNN/traffic.json, written by scripts/pmc_run.sh + scripts/pmc_to_json.py: separate --pmc passes, KB units,
    the factors of the two counters calibrated by a 1 GiB copy in the same run).  bench.py cannot profile itself; the record is
    tagged with the hash of the kernel sources it was taken with and is only reported when that hash is the one of the
    library being benchmarked now -- otherwise `traffic` is null rather than stale.  Returns (bytes, source, reads, writes)."""
    try:
        from ttcr_amd.build import source_hash
        note = None
        for d in PROFILE_DIRS:
            path = os.path.join(ROOT, "profiles", d, "traffic.json")
            if not os.path.exists(path):
                continue
            with open(path) as f:
                rec = json.load(f)
            if rec.get("source_hash") != source_hash():
                note = note or "profiles/%s/traffic.json is from other kernel sources (%s): not reported" % (d, rec.get("source_hash"))
                continue
            if not (n == rec["size"] and n_src_rank0 == rec["sources"] and world == 1 and os.environ.get("TTCR_FSM_MODE", "2") == "2"):
                return None, None, None, None
            # (counter values x the factors the calibration copy of the same rocprofv3 run gave: scripts/pmc_to_json.py; records of earlier
            # rounds carry none: the guide's x2 for FETCH_SIZE on gfx950)
            rd = float(rec.get("read_factor_used", 2.0)) * rec["fetch_kb_per_launch"] * 1024.0
            wr = float(rec.get("write_factor_used", 1.0)) * rec["write_kb_per_launch"] * 1024.0
            return rd + wr, "profiles/%s/traffic.json" % d, rd, wr
        return None, note, None, None
    except Exception:
        return None, None, None, None


def measured_copy_bandwidth(dev, reps=5):
    """read + write bandwidth of a plain device-to-device copy of 1 GiB on this GPU (SURVEY 8d asks for the roofline
    fraction against a measured figure as well as against the 8 TB/s peak)"""
    import torch

    a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    a.fill_(1.0)
    b.copy_(a)
    torch.add(a, 1.0, out=b)   # (the calibration kernel of the HBM counters when this command is profiled: scripts/pmc_to_json.py)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    return 2.0 * a.numel() * 4 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def single_source_leg(n, dx, x, s_dev, local_rank, reps=5, arith=0):
    """The case north_star's roofline target is written for: ONE source on the same grid, same run, HIP events around the
    sweep launches of the solve (the first of the benchmark's sources, run to convergence).  arith = 1: the tolerance-grade
    arithmetic (option "arith"), with the RMS difference of its field from the default mode's (= the reference's, bit for bit)."""
    import cases
    import ttcr_amd

    g1 = ttcr_amd.Grid3d(x, x, x, n_threads=1, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=np.float32, device=local_rank)
    g1.set_slowness_device(s_dev.data_ptr(), s_dev.numel())
    src = cases.mt_sources(1)
    rcv = cases.rcv_lattice3d()
    extra = {}
    if arith:
        g1.raytrace(np.repeat(src, rcv.shape[0], axis=0), rcv)
        ref, it_ref = g1._flat_tt(0).astype(np.float64), g1.get_niter()
        g1.set_option("arith", 1)
        g1.raytrace(np.repeat(src, rcv.shape[0], axis=0), rcv)
        d = g1._flat_tt(0).astype(np.float64) - ref
        extra = {"arith": 1, "rms_vs_default_mode_s": float(np.sqrt(np.mean(d * d))), "max_abs_vs_default_mode_s": float(np.max(np.abs(d))),
                 "tolerance_rms_s": 1e-5, "sweep_iterations_default_mode": it_ref,
                 "accuracy_note": "default mode = the reference bit for bit (tests/test_baseline_configs_gpu.py); tolerance: BASELINE.json north_star"}
        del ref, d
    # the library's own choice for this grid first (exact skipping follows the model for launches of 1 024 ... 2 047 work units), then the
    # evaluate-all kernel the roofline figures of this leg are about (option skip = 0: algorithmic bytes and evaluated work coincide)
    g1.raytrace(np.repeat(src, rcv.shape[0], axis=0), rcv)
    g1.raytrace(np.repeat(src, rcv.shape[0], axis=0), rcv)
    td = g1.timing()
    extra["library_default_path"] = {"ms_of_sweep_launches_per_solve": round(td["sweep_ms"], 3), "ms_per_sweep_iteration": round(td["sweep_ms"] / max(g1.get_niter(), 1), 4),
                                     "evaluated_fraction": round(td["evaluated_updates"] / max(td["node_updates"], 1), 4), "kernel": g1.last_kernel()}
    g1.set_option("skip", 0)
    g1.raytrace(np.repeat(src, rcv.shape[0], axis=0), rcv)   # warm-up (graph capture)
    ms, its, ev = 0.0, 0, 0
    t = time.perf_counter()
    for _ in range(reps):
        g1.raytrace(np.repeat(src, rcv.shape[0], axis=0), rcv)
        tm = g1.timing()
        ms += tm["sweep_ms"]
        its += g1.get_niter()
        ev += tm["evaluated_updates"]
    wall = time.perf_counter() - t
    per_it = ms / its
    achieved = BYTES_PER_NODE_ITER / 8.0 * ev / (ms * 1e-3) / 1e9
    kern = g1.last_kernel()
    del g1
    out = {"ms_per_sweep_iteration": round(per_it, 4), "frac": round(achieved / HBM_PEAK_GBS, 4), "achieved_GBs": round(achieved, 1),
           "Mnodes_per_s_per_sweep_iteration": round(n ** 3 / per_it / 1e3, 1), "sweep_iterations": its // reps,
           "ms_per_solve_wall": round(wall / reps * 1e3, 3), "solves": reps, "kernel": kern,
           "note": "1 source (first of the set) on the same grid, same process; HIP events on the library's stream"}
    out.update(extra)
    return out


def small_batch_leg(n, dx, x, s_dev, local_rank, n_src, reps, ms64_per_step, arith=0):
    """What ONE GPU of an 8-GPU node runs when the 64 sources of the workload are sharded (Grid3D::raytrace's block distribution,
    ttcr/Grid3D.h:451-465, 810-853): n_src sources on the same grid, same process.  Whole steps (wall clock) and sweep launches."""
    import cases
    import ttcr_amd

    g = ttcr_amd.Grid3d(x, x, x, n_threads=n_src, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0, dtype=np.float32, device=local_rank)
    g.set_slowness_device(s_dev.data_ptr(), s_dev.numel())
    if arith:
        g.set_option("arith", 1)
    src = cases.mt_sources(64)[:n_src]
    rcv = cases.rcv_lattice3d()
    sr, rr = np.repeat(src, rcv.shape[0], axis=0), np.tile(rcv, (n_src, 1))
    g.raytrace(sr, rr)
    ms, ev, its, launches = 0.0, 0, 0, 0
    t = time.perf_counter()
    for _ in range(reps):
        g.raytrace(sr, rr)
        tm = g.timing()
        ms += tm["sweep_ms"]; ev += tm["evaluated_updates"]; its += tm["node_updates"] // 8; launches += tm["kernel_launches"]
    wall = (time.perf_counter() - t) / reps * 1e3
    achieved = BYTES_PER_NODE_ITER / 8.0 * ev / (ms * 1e-3) / 1e9
    out = {"sources": n_src, "ms_per_step_wall": round(wall, 3), "ms_of_sweep_launches_per_step": round(ms / reps, 3),
           "Mnodes_per_s_per_sweep_iteration": round(its / (wall * reps * 1e-3) / 1e6, 1),
           "frac": round(achieved / HBM_PEAK_GBS, 4), "evaluated_fraction": round(ev / max(its * 8, 1), 4),
           "sweep_iterations": sorted({g.get_niter(i) for i in range(n_src)}), "kernel": g.last_kernel(), "steps": reps}
    if ms64_per_step:
        out["projected_strong_scaling_efficiency_8_gpus"] = round(ms64_per_step / (8.0 * wall), 3)
        out["note"] = ("projection from one GPU: t(64 sources) / (8 x t(8 sources)), whole steps -- what the strong-scaling run of the "
                       "workload would show on 8 GPUs if nothing else were lost; not a measurement of 8 GPUs")
    return out, g


def heterogeneous_leg(g, n, reps=2):
    """The same 8-source grid on a model where every sweep-iteration does real work: uniform random slowness in [0.25, 1] per 16^3
    block (seed 5), run to convergence -- what exact skipping is worth when iterations 2 ... do not come for free."""
    import cases
    import torch

    rng = np.random.default_rng(5)
    nb = (n + 15) // 16
    b = torch.from_numpy(rng.uniform(0.25, 1.0, (nb, nb, nb)).astype(np.float32)).cuda()
    s = b.repeat_interleave(16, 0).repeat_interleave(16, 1).repeat_interleave(16, 2)[:n, :n, :n]
    s = s.permute(2, 1, 0).contiguous().reshape(-1)   # (nx, ny, nz) C order -> x fastest
    torch.cuda.synchronize()
    g.set_slowness_device(s.data_ptr(), s.numel())
    n_src = g.n_threads
    src = cases.mt_sources(64)[:n_src]
    rcv = cases.rcv_lattice3d()
    sr, rr = np.repeat(src, rcv.shape[0], axis=0), np.tile(rcv, (n_src, 1))
    g.raytrace(sr, rr)
    st0 = g.stopping_stats()
    ms, ev, its = 0.0, 0, 0
    t = time.perf_counter()
    for _ in range(reps):
        g.raytrace(sr, rr)
        tm = g.timing()
        ms += tm["sweep_ms"]; ev += tm["evaluated_updates"]; its += tm["node_updates"] // 8
    wall = (time.perf_counter() - t) / reps * 1e3
    st1 = g.stopping_stats()
    full = int(sum(np.sum(~np.isnan(g.get_reference_changes(i)[0])) for i in range(n_src)))   # (of the last solve)
    # the same solve decided by the fp64 sum alone (what round 4 timed): the price of the reference's own sum (option stopping_rule = 1, the default)
    g.set_option("stopping_rule", 0)
    g.raytrace(sr, rr)
    walls0 = []
    for _ in range(reps):
        t = time.perf_counter()
        g.raytrace(sr, rr)
        walls0.append((time.perf_counter() - t) * 1e3)
    wall0 = sum(walls0) / reps
    its0 = sorted({g.get_niter(i) for i in range(n_src)})
    g.set_option("stopping_rule", 1)
    g.raytrace(sr, rr)
    achieved = BYTES_PER_NODE_ITER / 8.0 * ev / (ms * 1e-3) / 1e9
    return {"model": "uniform random slowness in [0.25, 1] per 16^3 block (numpy default_rng(5))", "sources": n_src,
            "ms_per_step_wall": round(wall, 3), "ms_of_sweep_launches_per_step": round(ms / reps, 3),
            "stopping_rule": {"reference_sums_per_step": (st1["reference_sums"] - st0["reference_sums"]) // reps,
                              "missed_per_step": (st1["reference_sums_missed"] - st0["reference_sums_missed"]) // reps,
                              "of_them_summed_in_full_per_step": full,
                              "ms_per_step_wall_with_the_fp64_sum_alone": round(wall0, 3), "ms_of_each_such_step": [round(w, 3) for w in walls0], "sweep_iterations_with_the_fp64_sum_alone": its0,
                              "note": "iterations whose fp64 change lies within [1/2, 16] x eps N are decided by the reference's sequential T1 sum (ttcr/Grid3Drnfs.h:141-152), computed exactly and in parallel over the non-zero terms of a snapshot difference -- or by rigorous bounds on that sum where they leave no doubt (option stopping_shortcuts)"},
            "Mnodes_per_s_per_sweep_iteration": round(its / (wall * reps * 1e-3) / 1e6, 1),
            "frac": round(achieved / HBM_PEAK_GBS, 4), "evaluated_fraction": round(ev / max(its * 8, 1), 4),
            "sweep_iterations": sorted({g.get_niter(i) for i in range(n_src)}), "kernel": g.last_kernel(), "steps": reps}


def weno_leg(local_rank, n=256, arith=0):
    """The two-stage solver (weno=True, the ttcrpy default: first-order sweeps, then third-order WENO sweeps,
    ttcr/Grid3Drnfs.h:104-136) on the n^3 gradient model, 1 and 8 sources.  arith = 2: both stages with the tolerance-grade
    arithmetic (outside the 1e-5 s bound on grids with the WENO stage: include/ttcr_amd.h, option "arith")."""
    import cases
    import ttcr_amd

    dx = 20.0 / (n - 1)
    x = np.arange(n, dtype=np.float64) * dx
    s = np.ascontiguousarray(np.broadcast_to(gradient_slowness_f32(n, dx), (n, n, n)))
    out = {}
    for n_src in (1, 8):
        g = ttcr_amd.Grid3d(x, x, x, n_threads=n_src, cell_slowness=0, method="FSM", tt_from_rp=0, weno=1, dtype=np.float32, device=local_rank)
        g.set_slowness(s)
        if arith:
            g.set_option("arith", arith)
        src = cases.mt_sources(64)[:n_src]
        rcv = cases.rcv_lattice3d()
        sr, rr = np.repeat(src, rcv.shape[0], axis=0), np.tile(rcv, (n_src, 1))
        g.raytrace(sr, rr)
        t = time.perf_counter()
        g.raytrace(sr, rr)
        wall = (time.perf_counter() - t) * 1e3
        tm = g.timing()
        its = tm["node_updates"] // 8
        out[f"sources_{n_src}"] = {"ms_per_solve_wall": round(wall, 2), "ms_of_sweep_launches": round(tm["sweep_ms"], 2),
                                   "sweep_iterations_first_order": sorted({g.get_niter(i) for i in range(n_src)}),
                                   "sweep_iterations_weno": sorted({g.get_niterw(i) for i in range(n_src)}),
                                   "Mnodes_per_s_per_sweep_iteration": round(its / (wall * 1e-3) / 1e6, 1), "kernel_of_the_last_stage": g.last_kernel()}
        del g
    out["grid"] = f"{n}^3 nodes, gradient model, fp32, weno=True, tt_from_rp=False" + (f", option arith = {arith}" if arith else "")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=512, help="nodes per axis")
    ap.add_argument("--sources", type=int, default=64, help="total sources of the job (sharded over the GPUs)")
    ap.add_argument("--sources-per-gpu", type=int, default=0, help="override: fixed sources per GPU (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-source", action="store_true", help="skip the 1-source leg of the N=1 run")
    ap.add_argument("--max-batch", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="key=value handed to ttcr_fsm_set_option (tuning / bisecting)")
    ap.add_argument("--per-step", action="store_true", help="print launches / evaluated updates of every timed step to stderr")
    ap.add_argument("--force-dist", action="store_true", help="N = 1 only: create the process group (backend nccl = RCCL) for the one rank and run "
                    "the broadcast of the model and the all_gather of every step through it, as the N > 1 runs do")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo only to "
                    "exercise the multi-rank path on a single-GPU box)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import cases
    import ttcr_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    ndev = torch.cuda.device_count()
    if args.backend != "nccl":
        local_rank = local_rank % max(ndev, 1)  # test mode: ranks may share a device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where collectives run
    use_dist = world > 1 or args.force_dist   # collectives are issued (a group of one rank with --force-dist)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    n = args.size
    from ttcr_amd.dist import shard_bounds
    if args.sources_per_gpu > 0:
        n_total, weak = args.sources_per_gpu * world, True
    else:
        n_total, weak = args.sources, False
    src_lo, src_hi = shard_bounds(n_total, world, rank)
    S = src_hi - src_lo
    dx = 20.0 / (n - 1)
    x = np.arange(n, dtype=np.float64) * dx
    grid = ttcr_amd.Grid3d(x, x, x, n_threads=S, cell_slowness=0, method="FSM", tt_from_rp=0, weno=0,
                           dtype=np.float32, device=local_rank)
    if args.max_batch > 0:
        grid.set_option("max_batch", args.max_batch)
    for kv in args.opt:
        k_, v_ = kv.split("=")
        grid.set_option(k_, float(v_))

    # slowness: generated on rank 0 in HBM, broadcast over RCCL/xGMI, handed over as a device pointer
    s_dev = torch.empty(n * n * n, dtype=torch.float32, device=dev)
    if rank == 0:
        sz = torch.from_numpy(gradient_slowness_f32(n, dx)).to(dev)
        s_dev.copy_(sz.repeat_interleave(n * n))
        del sz
    if use_dist:
        if args.backend == "nccl":
            dist.broadcast(s_dev, src=0)
        else:
            tmp = s_dev.cpu()
            dist.broadcast(tmp, src=0)
            s_dev.copy_(tmp)
            del tmp
    torch.cuda.synchronize()
    grid.set_slowness_device(s_dev.data_ptr(), s_dev.numel())

    all_src = cases.mt_sources(max(64, n_total))
    my_src = all_src[src_lo:src_hi]
    rcv1 = cases.rcv_lattice3d()
    # ttcrpy convention: one (source, receiver) row pair per datum
    src_rows = np.repeat(my_src, rcv1.shape[0], axis=0)
    rcv_rows = np.tile(rcv1, (S, 1))
    n_nodes = n ** 3

    # equal-sized gather buffers (ranks own at most ceil(n_total/world) sources)
    max_rows = -(-n_total // world) * rcv1.shape[0]
    # (all_gather rather than gather: a few KB, and the one collective every backend implements natively)
    gathered = [torch.empty(max_rows, dtype=torch.float32, device=cdev) for _ in range(world)]

    def step():
        tt = grid.raytrace(src_rows, rcv_rows)
        tm = grid.timing()
        if use_dist:
            t_dev = torch.zeros(max_rows, dtype=torch.float32, device=cdev)
            t_dev[:tt.shape[0]] = torch.from_numpy(tt).to(cdev)
            dist.all_gather(gathered, t_dev)
        return tt, tm

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    sweep_ms = 0.0
    launches = 0
    node_iters = 0
    evaluated = 0
    step_launches, step_niter = [], []
    host_ms = 0.0
    for _ in range(args.steps):
        tw = time.perf_counter()
        tt, tm = step()
        tw = (time.perf_counter() - tw) * 1e3
        host_ms += tw - tm["sweep_ms"]
        sweep_ms += tm["sweep_ms"]
        launches += tm["kernel_launches"]
        node_iters += tm["node_updates"] // 8
        evaluated += tm["evaluated_updates"]
        step_launches.append(int(tm["kernel_launches"]))
        step_niter.append(tuple(grid.get_niter(i) for i in range(S)))   # (a few microseconds: host-side counters)
        if args.per_step:
            sys.stderr.write(f"step {len(step_launches) - 1}: launches {step_launches[-1]} evaluated {tm['evaluated_updates'] / n ** 3:.3f} N "
                             f"sweep_ms {tm['sweep_ms']:.2f} wall_ms {tw:.2f} lib_total_ms {tm['total_ms']:.2f} niter {sorted(set(step_niter[-1]))}\n")
    fence()
    el = time.perf_counter() - t0
    kernel_name = grid.last_kernel()
    iters_per_src = list(step_niter[-1])
    # every timed step solves the same sources on the same model: launch count and iteration counts must not move
    odd_steps = [k for k in range(args.steps) if step_launches[k] != step_launches[0] or step_niter[k] != step_niter[0]]

    stats = torch.tensor([el, sweep_ms, float(node_iters), float(launches)], dtype=torch.float64, device=cdev)
    per_rank = [(S, el)]
    if world > 1:
        assert dist.get_world_size() == args.gpus
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        el_max, sweep_ms_max = float(mx[0]), float(mx[1])
        node_iters_all = float(sm[2])
        mine = torch.tensor([float(S), el], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [(int(v[0]), float(v[1])) for v in allr]
    else:
        el_max, sweep_ms_max, node_iters_all = el, sweep_ms, float(node_iters)

    # which device every rank ran on (uuid where torch exposes it): the first multi-GPU run is self-verifying
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        mine_id = f"rank {rank}: cuda:{local_rank} {pr.name} uuid {getattr(pr, 'uuid', 'n/a')}"
    except Exception as e:
        mine_id = f"rank {rank}: cuda:{local_rank} ({e})"
    if world > 1:
        dev_ids = [None] * world
        dist.all_gather_object(dev_ids, mine_id)
    else:
        dev_ids = [mine_id]
    if rank == 0:
        assert np.all(np.isfinite(tt)) and float(tt.max()) < 1e3, "non-physical traveltimes"
        value = node_iters_all / el_max / 1e6
        # roofline of the dominant kernel (fsm_sweep_tile), rank 0's launches: algorithmic bytes of
        # the nodes one launch sequence sweeps / HIP-event time of those launches (ttcr_fsm_last_timing)
        # Chunks whose inputs provably did not change since their last evaluation are skipped
        # (exact); only the node updates that were really evaluated are priced, at 104/8 B each.
        bytes_total = BYTES_PER_NODE_ITER / 8.0 * evaluated
        achieved = bytes_total / (sweep_ms * 1e-3) / 1e9
        nominal = BYTES_PER_NODE_ITER * node_iters / (sweep_ms * 1e-3) / 1e9
        traffic, traffic_src, traffic_rd, traffic_wr = profiled_traffic(n, S, world)
        out = {
            "metric": "Mnodes/s per sweep-iteration (512^3 fp32 grid, first-order FSM)",
            "value": round(value, 1),
            "unit": "Mnodes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(el_max / args.steps * 1e3, 3),
            "ms_per_step_outside_sweep_launches": round(host_ms / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "build_id": ttcr_amd._lib.build_id(),
            "build_id_note": "hash of the kernel sources + compiler flags baked into libttcr_amd.so; the loader refuses a library whose id is not the hash of the sources beside it (ttcr_amd/_lib.py)",
            "arith": "default (the reference's arithmetic: results bit-identical to the reference); the opt-in tolerance-grade mode is reported under tolerance_mode / *_tolerance",
            "sources_per_s": round(n_total * args.steps / el_max, 3),
            "config": {"workload": f"Grid3d {n}^3 nodes gradient velocity, {n_total} sources of the mt19937_64(12345) "
                                   f"set block-distributed over {world} GPU(s) ({S} on rank 0), 441 receivers, fp32, "
                                   f"weno=False, tt_from_rp=False",
                       "grid_nodes": n_nodes, "sources_total": n_total, "sources_rank0": S,
                       "sources_per_rank": [p[0] for p in per_rank],
                       "sources_per_s_per_rank": [round(p[0] * args.steps / p[1], 3) for p in per_rank],
                       "sweep_iterations_per_source": sorted(set(iters_per_src)),
                       "launches_per_step": sorted(set(step_launches)),
                       "steps_that_differ_from_the_first": odd_steps,
                       "parallelism": f"source-sharded x{world} (RCCL broadcast of slowness, all_gather of receiver traveltimes)",
                       "collective_backend": (dist.get_backend() if use_dist else None), "world_size": world,
                       "collectives_note": ("--force-dist: a process group of one rank; the broadcast of the model and the all_gather of every timed "
                                            "step went through the backend" if (use_dist and world == 1) else None),
                       "devices": dev_ids},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src, "traffic_reads": traffic_rd, "traffic_writes": traffic_wr,
                         "frac_real_traffic": (round(traffic * launches / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None),
                         "frac_real_traffic_note": ("PMC bytes of the committed profile (another box, same kernel sources) over THIS run's launch time" if traffic else None),
                         "frac_contract_all_updates": round(nominal / HBM_PEAK_GBS, 4),
                         "kernel": kernel_name + " (as reported by the library: ttcr_fsm_last_kernel; 8th parameter = sources marched per workgroup, 6th = exact skipping)",
                         "algorithmic_bytes_per_node_per_sweep_iteration": BYTES_PER_NODE_ITER,
                         "evaluated_fraction": round(evaluated / max(node_iters * 8, 1), 4),
                         "nominal_GBs_all_updates": round(nominal, 1),
                         "launches": int(launches), "avg_launch_us_hip_events": round(sweep_ms * 1e3 / max(launches, 1), 3),
                         "algorithmic_bytes_per_launch": round(bytes_total / max(launches, 1), 1),
                         "pricing_note": "achieved / frac price only the node updates the kernel EVALUATED, at the contract's 104/8 B "
                                         "each (T read, s read, T write + the snapshot share), whether or not they changed the node; "
                                         "chunks, units and sweeps that provably cannot change a node are stepped over (exact) and "
                                         "priced at nothing -- frac_contract_all_updates is SURVEY 8(d)'s 104 B x N x iterations / time, "
                                         "which counts them; frac_real_traffic is the PMC traffic of profiles/ over the same time"},
        }
        if world == 1:
            try:
                cp = measured_copy_bandwidth(dev)
                out["roofline"]["measured_copy_GBs"] = round(cp, 1)
                out["roofline"]["measured_copy_note"] = "a torch Tensor.copy_ of 1 GiB (read + write bytes over its time), not a tuned stream kernel: the guide's achievable figure is higher (about 6.3 TB/s), so frac_of_measured_copy flatters"

                out["roofline"]["frac_of_measured_copy"] = round(achieved / cp, 4)
            except Exception as e:   # (reported, never fatal: the contract fields above do not depend on it)
                out["roofline"]["measured_copy_GBs"] = None
                sys.stderr.write("copy bandwidth not measured: %s\n" % e)
        if world == 1 and not args.no_single_source:
            # the kernel with every node update evaluated (exact skipping switched off), same grid, same sources, same
            # process: the roofline of the sweep kernel itself, where algorithmic bytes and evaluated work coincide
            try:
                if os.environ.get("TTCR_FSM_SKIP") is None:
                    grid.set_option("skip", 0)
                    grid.raytrace(src_rows, rcv_rows)
                    ms0, l0, ev0 = 0.0, 0, 0
                    for _ in range(2):
                        grid.raytrace(src_rows, rcv_rows)
                        tm0 = grid.timing()
                        ms0 += tm0["sweep_ms"]; l0 += tm0["kernel_launches"]; ev0 += tm0["evaluated_updates"]
                    grid.set_option("skip", -1)
                    a0 = BYTES_PER_NODE_ITER / 8.0 * ev0 / (ms0 * 1e-3) / 1e9
                    out["roofline"]["evaluate_all_kernel"] = {
                        "kernel": "fsm_sweep_persistent<float,16,16,8,true,false,1,2,true,true>", "frac": round(a0 / HBM_PEAK_GBS, 4),
                        "achieved": round(a0, 1), "avg_launch_us_hip_events": round(ms0 * 1e3 / max(l0, 1), 3), "launches": int(l0),
                        "note": "option skip = 0: every node update of every sweep evaluated (what round 2 reported as roofline); 2 steps after the timed region"}
            except Exception as e:
                sys.stderr.write("evaluate-all leg failed: %s\n" % e)
            out["single_source"] = single_source_leg(n, dx, x, s_dev, local_rank)
            try:
                out["single_source_tolerance"] = single_source_leg(n, dx, x, s_dev, local_rank, arith=1)
            except Exception as e:
                out["single_source_tolerance"] = {"error": str(e)[:300]}
            # the headline batch with the tolerance-grade arithmetic (option arith = 1), same grid, same sources, after the timed region
            try:
                ref_rcv = grid.raytrace(src_rows, rcv_rows).astype(np.float64)
                samp = [grid._flat_tt(i)[::4099].astype(np.float64) for i in range(S)]
                grid.set_option("arith", 1)
                grid.raytrace(src_rows, rcv_rows)
                tq = time.perf_counter()
                ms1, it1, ev1 = 0.0, 0, 0
                for _ in range(args.steps):
                    tt1 = grid.raytrace(src_rows, rcv_rows)
                    tm1 = grid.timing()
                    ms1 += tm1["sweep_ms"]; it1 += tm1["node_updates"] // 8; ev1 += tm1["evaluated_updates"]
                el1 = time.perf_counter() - tq
                d2 = np.concatenate([grid._flat_tt(i)[::4099].astype(np.float64) - samp[i] for i in range(S)])
                a1 = BYTES_PER_NODE_ITER / 8.0 * ev1 / (ms1 * 1e-3) / 1e9
                out["tolerance_mode"] = {
                    "option": "arith = 1 (update3_fast: fp32 scaled-difference quadratics, one v_sqrt_f32 per update; opt-in, NOT bit-identical)",
                    "value": round(it1 / el1 / 1e6, 1), "unit": "Mnodes/s", "ms_per_step": round(el1 / args.steps * 1e3, 3), "steps": args.steps,
                    "roofline_frac_evaluated": round(a1 / HBM_PEAK_GBS, 4), "kernel": grid.last_kernel(),
                    "sweep_iterations_per_source": sorted({grid.get_niter(i) for i in range(S)}),
                    "rms_vs_default_mode_s": float(np.sqrt(np.mean(d2 * d2))), "max_abs_vs_default_mode_s": float(np.max(np.abs(d2))),
                    "rms_receivers_vs_default_mode_s": float(np.sqrt(np.mean((tt1.astype(np.float64) - ref_rcv) ** 2))),
                    "tolerance_rms_s": 1e-5,
                    "accuracy_note": "every 4099th node of all fields against the default mode on the same grid (= the reference bit for bit); whole fields against the CPU oracle: tests/test_arith_mode_gpu.py"}
                grid.set_option("arith", 0)
                del samp, d2
            except Exception as e:
                out["tolerance_mode"] = {"error": str(e)[:300]}
            # the small-batch regime (one GPU's share of the workload on an 8-GPU node), a model without free iterations, the WENO stage
            for name, fn in (("eight_sources", None), ("heterogeneous", None), ("weno", None), ("weno_arith2", None)):
                try:
                    if name == "eight_sources":
                        out[name], g8 = small_batch_leg(n, dx, x, s_dev, local_rank, 8, 3, el_max / args.steps * 1e3 if n_total == 64 else None)
                        try:
                            out["eight_sources_tolerance"], g8t = small_batch_leg(n, dx, x, s_dev, local_rank, 8, 3, None, arith=1)
                            if n_total == 64 and "tolerance_mode" in out and "ms_per_step" in out["tolerance_mode"]:
                                out["eight_sources_tolerance"]["projected_strong_scaling_efficiency_8_gpus"] = round(
                                    out["tolerance_mode"]["ms_per_step"] / (8.0 * out["eight_sources_tolerance"]["ms_per_step_wall"]), 3)
                            del g8t
                        except Exception as e:
                            out["eight_sources_tolerance"] = {"error": str(e)[:300]}
                    elif name == "heterogeneous":
                        out[name] = heterogeneous_leg(g8, n)
                        del g8
                    elif name == "weno":
                        out[name] = weno_leg(local_rank)
                    else:
                        out[name] = weno_leg(local_rank, arith=2)
                except Exception as e:   # (reported, never fatal)
                    out[name] = {"error": str(e)[:300]}
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed on rank 0 of the single-GPU run only
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if odd_steps:
        for k in odd_steps[:8]:
            w = [i for i in range(S) if step_niter[k][i] != step_niter[0][i]]
            sys.stderr.write(f"rank {rank}: step {k}: {step_launches[k]} launches (step 0: {step_launches[0]}), niter differs in slots "
                             f"{w}: {[step_niter[k][i] for i in w]} vs {[step_niter[0][i] for i in w]}\n")
        raise SystemExit(f"bench.py: {len(odd_steps)} of {args.steps} timed steps differ from the first in launch or iteration count")


if __name__ == "__main__":
    main()
