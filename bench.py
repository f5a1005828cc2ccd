#!/usr/bin/env python
"""Benchmark of the outerbase hot path on MI355X: fit + predict points/sec.

One "step" = one full pass of the hot path over one synthetic batch that is
already resident in HBM (BASELINE.md sections 2-3):

  fit      standardise y (R/fitting.R:55-57) -> basis build (outerbase::build)
           -> Gram B^T B on the FP64 matrix cores + B^T y -> [all-reduce over
           ranks] -> H = e^{-2 sigma} G + prior, Cholesky, two triangular solves
           (lpdf::optnewton, "back end A")
  predict  fused basis build at n fresh rows + B theta (predictor$update/$mean)

Default workload: BASELINE.json configs[2] = d=20, n=1e6, p=4096, Matern-5/2 in
every dimension, 40 knots per dimension, rows sharded over ranks (weak
scaling: every rank owns n rows).  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X datasheet FP64 matrix (dense); see DESIGN.md
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md "HBM3E peak BW"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", dest="n", type=int, default=1_000_000, help="rows per GPU")
    ap.add_argument("--d", "--dims", dest="d", type=int, default=20,
                    help="(--dims: torch.distributed.run swallows an abbreviated --d)")
    ap.add_argument("--p", type=int, default=4096)
    ap.add_argument("--knots", type=int, default=40)
    ap.add_argument("--backend", choices=["newton", "cg"], default="newton")
    ap.add_argument("--kinds", default="mat25", help="comma list cycled over dimensions")
    ap.add_argument("--gram-backend", type=int, default=0,
                    help="0 auto, 1 MFMA 16x16x4, 2 vector pipe, 3 fused MFMA 4x4x4, 4 materialised-B MFMA 4x4x4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-backend", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=200000)
    return ap.parse_args()


def check_against_oracle(hp, rows=2000):
    """Untimed parity of THIS run against the CPU oracle (test infrastructure):
    (a) device predictions on the first rows of the prediction shard vs the
    oracle's basis build + B theta with the device's theta; (b) Newton
    stationarity of the device theta, H theta = e^{-2 sigma} B^T y, evaluated with
    the matrix-free device kernels (independent of the Gram/Cholesky kernels)."""
    import ctypes as C
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ob_oracle as O
    from outerbase_amd._lib import call
    rows = min(rows, hp.n)
    om = O.OuterMod()
    om.setcovfs(hp.kinds)
    om.setknot(O.bench_knots(hp.kinds, hp.m))
    rot, bv, ml = hp.om.rotation()
    om.rotmat, om.basisvar, om.maxlevel = rot, bv, ml   # share the device model's rotation
    theta = hp.theta.cpu().numpy()
    xnew = hp.xnew[:, :rows].cpu().numpy().T
    want = hp.y_cent + hp.y_sca * O.predict_mean(om, hp.terms, theta, xnew)
    got = hp.mean[:rows].cpu().numpy()
    out = {"rows": rows,
           "predict_max_rel_err": float(np.max(np.abs(got - want)) / np.max(np.abs(want)))}
    if hp.backend == "newton" and hp.world == 1:
        e2 = math.exp(-2 * hp.sigma)
        tmp = torch.empty(hp.n, dtype=torch.float64, device="cuda")
        hv = torch.empty(hp.p, dtype=torch.float64, device="cuda")
        call("obhip_basis_mm_dev", hp.basis, hp.t._h, hp.theta.data_ptr(), tmp.data_ptr(), 0)
        call("obhip_basis_tmm_dev", hp.basis, hp.t._h, tmp.data_ptr(), hv.data_ptr(), 0)
        torch.cuda.synchronize()
        prec = 1.0 / (hp.om.getvar(hp.terms) * math.exp(2 * hp.rho))
        lhs = e2 * hv.cpu().numpy() + prec * theta
        rhs = e2 * hp.g.cpu().numpy()
        out["newton_residual_rel"] = float(np.linalg.norm(lhs - rhs) / np.linalg.norm(rhs))
    return out


def cpu_baseline(hp, ns, threads=16):
    """CPU restatement of the reference path timed on the host cores on a bounded row
    sample of the same workload (test infrastructure, oracle/): the reference's own
    loops (outerbase::build, getm_, prodmm_) in C++/OpenMP with the reference's chunk
    schedule (oracle/ob_cpu.cpp), BLAS/LAPACK (NumPy) for basismat.t()*basismat and
    solve() exactly where the reference hands over to Armadillo.  Row-proportional
    work is scaled to n, the p x p solve is counted once."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ob_oracle as O
    import ob_cpu
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    threads = min(threads, os.cpu_count() or 1)
    ns = min(ns, hp.n)
    use_cpp = ob_cpu.available()

    def run():
        om = O.OuterMod()
        om.setcovfs(hp.kinds)
        om.setknot(O.bench_knots(hp.kinds, hp.m))
        x, y = O.synth_xy(42, 0, ns, hp.kinds)
        xnew, _ = O.synth_xy(43, 0, ns, hp.kinds)
        y = (y - y.mean()) / y.std(ddof=1)
        t = {}
        t0 = time.perf_counter()
        if use_cpp:
            bm, bs = ob_cpu.build(om, x, threads)            # outerbase::build, all knots
        else:
            ob = O.OuterBase(om, x)
            bm, bs = ob.basemat, ob.basescale
        t["build"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        if use_cpp:
            B = ob_cpu.getmat(om, hp.terms, bm, bs, threads)  # getm_ (loglik_std ctor)
        else:
            B = O.getm(hp.terms, bm, bs, om.knotptst)
        t["getmat"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        sigma = O.default_sigma(y)
        H = math.exp(-2 * sigma) * (B.T @ B)                  # loglik_std::hess
        g = math.exp(-2 * sigma) * (B.T @ y)
        H[np.diag_indices_from(H)] += O.prior_prec(om, hp.terms, hp.rho)
        t["gram"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        theta = np.linalg.solve(H, g)                         # fit.cpp:120
        t["solve"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        if use_cpp:                                           # predictor update + mean
            bmn, bsn = ob_cpu.build(om, xnew, threads)
            ob_cpu.mm(om, hp.terms, bmn, bsn, theta, threads)
        else:
            O.predict_mean(om, hp.terms, theta, xnew)
        t["predict"] = time.perf_counter() - t0
        if use_cpp:   # one B a and one B^T r pass: what a PCG iteration is made of (fit.cpp:71-85)
            r = y - B @ theta
            t0 = time.perf_counter()
            ob_cpu.mm(om, hp.terms, bm, bs, theta, threads)
            ob_cpu.tmm(om, hp.terms, bm, bs, r, threads)
            t["mm_tmm"] = time.perf_counter() - t0
        return t

    if threadpool_limits is not None:
        with threadpool_limits(limits=threads):
            t = run()
    else:
        t = run()
    per_row = (t["build"] + t["getmat"] + t["gram"] + t["predict"]) / ns
    full = per_row * hp.n + t["solve"]
    impl = "oracle/ob_cpu.cpp (C++/OpenMP, reference chunk schedule)" if use_cpp \
        else "oracle/ob_oracle.py (NumPy)"
    pcg = None
    if "mm_tmm" in t:
        # the matrix-free path obfit itself takes -- build, then per
        # iteration update() + hessmult() = two B a and two B^T r passes (fit.cpp:71-85), at
        # the iteration count the device PCG needed, then predict
        iters = getattr(hp, "cg_iters", None) or 22
        full_pcg = (t["build"] + t["predict"] + 2 * (iters + 1) * t["mm_tmm"]) / ns * hp.n
        pcg = {"value": hp.n / full_pcg, "unit": "points/s", "iterations": iters,
               "mm_plus_tmm_s_on_sample": t["mm_tmm"]}
    return {"value": hp.n / full, "unit": "points/s", "cores": threads, "kind": "port", "pcg_path": pcg,
            "sample": "%s + NumPy BLAS/LAPACK for B^T B and solve, %d threads, %d rows of the "
                      "same workload: build %.2fs getmat %.2fs gram %.2fs solve %.2fs predict "
                      "%.2fs; row work scaled to n=%d, solve counted once"
                      % (impl, threads, ns, t["build"], t["getmat"], t["gram"], t["solve"],
                         t["predict"], hp.n)}


def which_config(d, n, p):
    """Name of the BASELINE.json configuration these sizes are (rows per GPU)."""
    known = {(10, 100000, 1024): "BASELINE.json configs[1]", (20, 1000000, 1024 * 4): "BASELINE.json configs[2]",
             (20, 1250000, 4096): "BASELINE.json configs[3] (one of 8 row shards)",
             (40, 125000, 16384): "BASELINE.json configs[4] (one of 8 row shards)"}
    return known.get((d, n, p), "custom sizes")


def n_rows_all(hp):
    return hp.n * hp.world


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0 and world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (libobhip has no CPU fallback)")
    # one rank per GPU; OBHIP_DIST_BACKEND=gloo lets several ranks rehearse the N > 1 path
    # on a single GPU (gloo stages CUDA tensors through the host)
    backend = os.environ.get("OBHIP_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev = local if backend == "nccl" else local % max(1, ndev)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    import outerbase_amd as ob
    from outerbase_amd import _lib
    from outerbase_amd.driver import HotPath

    kinds = [k.strip() for k in args.kinds.split(",")]
    kinds = [kinds[i % len(kinds)] for i in range(args.d)]
    hp = HotPath(kinds, args.knots, args.p, args.n, rank=rank, world=world,
                 backend=args.backend)
    hp.setup()
    _lib.call("obhip_set_gram_backend", args.gram_backend)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        hp.step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hp.step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel timing (hipEvents on the launch stream), outside the timed region
    _lib.call("obhip_profile_reset")
    _lib.call("obhip_profile_enable", 1)
    nprof = 2
    for _ in range(nprof):
        hp.step()
    torch.cuda.synchronize()
    prof = {}
    for name in ["build_basis", "materialize_B", "gram", "gram_reduce", "tmm", "mm", "sqtmm", "form_hessian",
                 "cholesky", "backsolve", "predict"]:
        cnt, ms = C.c_uint64(0), C.c_double(0)
        _lib.call("obhip_profile_get", name.encode(), C.byref(cnt), C.byref(ms))
        if cnt.value:
            prof[name] = dict(launches=cnt.value, avg_ms=ms.value / cnt.value,
                              ms_per_step=ms.value / nprof)
    _lib.call("obhip_profile_enable", 0)

    # fit-only and predict-only wall times, outside the timed region
    split = {}
    for name, fn in (("fit", hp.fit), ("predict", hp.predict)):
        sync()
        t0 = time.perf_counter()
        for _ in range(2):
            fn()
        sync()
        dt = (time.perf_counter() - t0) / 2
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        split[name + "_ms"] = dt * 1e3
        split[name + "_only_points_per_s"] = float(n_rows_all(hp)) / dt

    # what a host-buffer caller pays on top: x, y, xnew in, mean out over PCIe
    # (pinned buffers); reported beside `value`, never part of it
    pcie = None
    if rank == 0:
        hx = torch.empty(hp.x.shape, dtype=torch.float64).pin_memory()
        hy = torch.empty(hp.n, dtype=torch.float64).pin_memory()
        sync_local = torch.cuda.synchronize
        sync_local()
        t0 = time.perf_counter()
        hp.x.copy_(hx, non_blocking=True)
        hp.xnew.copy_(hx, non_blocking=True)
        hp.y_raw.copy_(hy, non_blocking=True)
        hy.copy_(hp.mean, non_blocking=True)
        sync_local()
        pcie = (time.perf_counter() - t0) * 1e3
        # restore the synthetic inputs the copies overwrote
        hp.setup_inputs()
        sync_local()


    # the other back end on the same inputs, for the record (untimed region; every rank
    # takes part because the fit all-reduces): B = matrix-free PCG, what obfit() runs
    alt = None
    if args.backend == "newton" and not args.no_alt_backend:
        import numpy as np
        theta_newton = hp.theta.clone()
        mean_newton = hp.mean.clone()
        hp.backend = "cg"
        hp.step()
        sync()
        t0 = time.perf_counter()
        hp.step()
        sync()
        dt = time.perf_counter() - t0
        rel = float((hp.mean - mean_newton).abs().max() / mean_newton.abs().max())
        alt = {"backend": "cg (lpdf::optcg, tol 1e-10, cap .getsteps)", "ms_per_step": dt * 1e3,
               "points_per_s": float(n_rows_all(hp)) / dt, "cg_iterations": hp.cg_iters,
               "max_rel_diff_of_predictions_vs_newton": rel}
        hp.backend = "newton"
        hp.theta.copy_(theta_newton)
        hp.mean.copy_(mean_newton)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    n, p = args.n, args.p
    pts = float(n) * world * args.steps
    ms_per_step = elapsed / args.steps * 1e3
    out = {
        "metric": "fit+predict points/sec, d=%d n=%g p=%d" % (args.d, n, p),
        "value": pts / elapsed,
        "unit": "points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "%s: d=%d n=%d rows/GPU p=%d, %s, %d knots/dim, "
                        "fit (%s) + predict on n fresh rows"
                        % (which_config(args.d, n, p), args.d, n, p, "/".join(sorted(set(kinds))),
                           args.knots, "Gram+Cholesky" if args.backend == "newton" else "PCG"),
            "backend": args.backend,
            "rows_per_gpu": n, "d": args.d, "p": p,
            "terms_nnz": hp.terms_info["nnz_total"], "basis_columns": hp.ncols,
            "parallelism": "rows sharded over %d rank(s); all-reduce of G and g" % world,
        },
        "fit_predict_split": split,
        "host_buffer_overhead": None if pcie is None else {
            "pcie_ms_per_step": pcie,
            "what": "x, xnew, y host->device and mean device->host, pinned, one GPU",
            "points_per_s_including_copies": float(n) * world / (ms_per_step * 1e-3 + pcie * 1e-3)},
        "kernels_ms": prof,
        "parity_check": None,
        "alt_backend": alt,
    }
    if "gram" in prof and args.backend == "newton":
        flops = float(n) * p * (p + 1)  # p(p+1) flop per point
        ach = flops / (prof["gram"]["avg_ms"] * 1e-3) / 1e12
        # HBM-side bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE,
        # WRITE_SIZE) on this exact workload; see the file for the command and caveats
        traffic = None
        kernel = {0: "k_gram_dma2", 1: "k_gram", 2: "k_gram_valu", 3: "k_gram_mfma4",
                  4: "k_gram_dma2"}[args.gram_backend]
        tf = os.path.join(ROOT, "profiles", "r01_gram_traffic.json")
        if os.path.exists(tf):
            tj = json.load(open(tf))
            c = tj["config"]
            if tj["kernel"] == kernel and \
                    (c["d"], c["rows"], c["p"], c["knots"]) == (args.d, n, p, args.knots):
                traffic = tj["traffic_bytes_per_launch"]
        out["roofline"] = {"bound": "mfma", "kernel": kernel, "achieved": ach,
                           "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                           "traffic_unit": "bytes per launch (PMC, separate pass)",
                           "avg_launch_ms": prof["gram"]["avg_ms"]}
    elif "mm" in prof:
        byts = float(n) * 8 * (hp.ncols + 1)
        ach = byts / (prof["mm"]["avg_ms"] * 1e-3) / 1e9
        # the PCG back end's kernels are LDS-bound (80 % LdsUtil, profiles/r01_pmc_products.txt);
        # the contract's roofline object only knows hbm | mfma, so this is the HBM view of k_mm_tl
        out["roofline"] = {"bound": "hbm", "kernel": "k_mm_tl", "achieved": ach, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                           "avg_launch_ms": prof["mm"]["avg_ms"],
                           "note": "LDS-bound kernel, see profiles/r01_pmc_products.txt"}
    if not args.no_cpu_baseline and world == 1:
        # the CPU leg: the oracle as the timed baseline and as the checker of this very run
        # (device predictions and Newton stationarity on a row sample); nothing else in this
        # file touches oracle/
        if alt and hp.cg_iters is None:
            hp.cg_iters = alt.get("cg_iterations")
        out["cpu_baseline"] = cpu_baseline(hp, args.cpu_sample)
        out["parity_check"] = check_against_oracle(hp)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
