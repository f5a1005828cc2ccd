#!/usr/bin/env python
"""Benchmark of the outerbase hot path on MI355X: fit + predict points/sec.

One "step" = one full pass of the hot path over one synthetic batch that is
already resident in HBM (BASELINE.md sections 2-3):

  fit      y standardised over all rows (R/fitting.R:55-57; 24 bytes cross the ranks) ->
           basis build (outerbase::build) -> Gram B^T B on the FP64 matrix cores, B^T y ->
           [ONE exchange buffer summed over ranks: packed upper triangle of G + B^T y, written
           by the Gram reduction itself] -> H = e^{-2 sigma} G + prior formed by the unpack
           (one rank: by the reduction), Cholesky, two triangular solves (lpdf::optnewton,
           "back end A").
  predict  fused basis build at the rank's n fresh rows + B theta (predictor$update/$mean)

Default workload: BASELINE.json configs[2] = d=20, n=1e6, p=4096, Matern-5/2 in every
dimension, 40 knots per dimension.  With --gpus N the SAME n = 1e6 rows are sharded over
the N ranks ("scaling": "strong", what the metric "d=20 n=1e6 p=4096 at 1/2/4/8 MI355X"
says); the line also carries `configs`: the other BASELINE.json configurations that fit one
GPU, timed after the headline through the same loop (configs[1] per GPU, configs[3]'s 1.25e6
rows per GPU -- also under the old key `config3` --, configs[4]'s 125 000-row shard per GPU).
Before the warm-up a job with more than one rank verifies its transport
(obhip_comm_selftest_dev); after the timed region every run is checked, at any rank count:
`parity_check` = rank 0's predictions against the oracle's basis, Newton stationarity with the
matrix-free kernels summed through the communicator, and `fit_vs_oracle` -- Gram + Cholesky +
predict of the device against the oracle's own fit on the first 20 000 rows.
`obfit_eval` (one rank): one second-stage function evaluation of obfit on the same rows and
terms -- device PCG fit plus all hyper-parameter gradients -- with its phases.
`python bench.py --gpus N` launches its own N ranks (one process per GPU,
torch.distributed.run) when it is not already running under a launcher; `--sim-ranks N` times
on ONE GPU the step a rank of an N-GPU job runs.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X datasheet FP64 matrix (dense); see DESIGN.md
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md "HBM3E peak BW"
LDS_PEAK_GBS = 256 * 256 * 2.4  # 256 B/clk/CU (ds_read_b64) x 256 CUs x 2.4 GHz = 157 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", dest="n", type=int, default=1_000_000,
                    help="rows of the whole job (sharded over the ranks); with --weak: rows per GPU")
    ap.add_argument("--weak", action="store_true", help="--rows is per GPU (weak scaling)")
    ap.add_argument("--d", "--dims", dest="d", type=int, default=20,
                    help="(--dims: torch.distributed.run swallows an abbreviated --d)")
    ap.add_argument("--p", type=int, default=4096)
    ap.add_argument("--knots", type=int, default=40)
    ap.add_argument("--backend", choices=["newton", "cg"], default="newton")
    ap.add_argument("--kinds", default="mat25", help="comma list cycled over dimensions")
    ap.add_argument("--gram-backend", type=int, default=0, choices=[0, 3, 4],
                    help="0 automatic, 3 fused MFMA 4x4x4 (no staging memory), 4 staged design matrix MFMA 4x4x4")
    ap.add_argument("--dump", default=None,
                    help="rank 0 writes theta and its first 1000 de-standardised predictions here (.npz)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-backend", action="store_true")
    ap.add_argument("--no-obfit-eval", action="store_true",
                    help="skip the timing of one obfit function evaluation (PCG fit + hyper-gradients)")
    ap.add_argument("--no-config3", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the other BASELINE.json configurations (configs[1], configs[4]'s shard)")
    ap.add_argument("--no-fit-parity", action="store_true",
                    help="skip the oracle-compared fit on the first rows (parity_check.fit_vs_oracle)")
    ap.add_argument("--fit-parity-rows", type=int, default=20000)
    ap.add_argument("--sim-ranks", type=int, default=0,
                    help="one process, N VIRTUAL ranks holding this GPU's shard (obhip_comm_init_sim): "
                         "the step a rank of an N-GPU job runs, exchange replaced by one device pass")
    ap.add_argument("--cpu-panel", type=int, default=200000,
                    help="rows per panel of the CPU leg's Gram path (all n rows are processed)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell: start N fresh ranks (one process per
    GPU) BEFORE this process makes any GPU call, relay their output, return their exit code.
    Nothing is exec'ed: the ranks are children of this process."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def check_against_oracle(hp, rows=2000):
    """Untimed parity of THIS run against the CPU oracle (test infrastructure):
    (a) device predictions on the first rows of the prediction shard vs the
    oracle's basis build + B theta with the device's theta; (b) Newton
    stationarity of the device theta, H theta = e^{-2 sigma} B^T y, evaluated with
    the matrix-free device kernels (independent of the Gram/Cholesky kernels)."""
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
    # the synthetic rows themselves (first rows of the shard) against the oracle's generator
    xo, yo = O.synth_xy(hp.seed_train, hp.row0, min(rows, 256), hp.kinds)
    out["synthetic_rows_max_abs_diff"] = float(
        np.max(np.abs(hp.x[:, :len(yo)].cpu().numpy().T - xo)))
    return out


def oracle_model(kinds, m):
    """the oracle's outermod (numpy.linalg.eigh where the reference calls arma::eig_sym,
    src/modandbase.cpp:236) on the bench knots: deterministic, so every rank can build it"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ob_oracle as O
    om = O.OuterMod()
    om.setcovfs(kinds)
    om.setknot(O.bench_knots(kinds, m))
    return O, om


def fit_vs_oracle(HotPath, shard_rows, kinds, knots, p, rows, rank, world, transport, pred_rows=2000):
    """Untimed: Gram + Cholesky + predict of the DEVICE path against the oracle's fit on the
    same rows -- the first `rows` rows of the seed-42 stream, sharded over the ranks like the
    headline run, predictions on the first rows of the seed-43 stream (rank 0's shard).  Two
    device runs: (i) nothing shared -- the library's own Jacobi eigensolver against the oracle's
    numpy.linalg.eigh (src/modandbase.cpp:236-255), terms selected by the library; (ii) the
    oracle's eigen-rotation injected into the device model.  The oracle fit follows
    loglik_std::hess + lpdf::optnewton (src/lpdfs/loglik_std.cpp:170-173, src/fit.cpp:98-131).
    Every rank takes part in the device fits; rank 0 runs the oracle and compares."""
    import numpy as np
    import torch
    O, om = oracle_model(kinds, knots)
    out = {"rows": rows, "p": p}
    row0, n_local = shard_rows(rank, world, rows)
    # rank 0: the oracle's fit first (its H is what the device's full H is compared with below)
    theta_o = H_o = want = None
    if rank == 0:
        t0 = time.perf_counter()
        terms_o = om.selectterms(p)
        x, y = O.synth_xy(42, 0, rows, kinds)
        cent, sca = y.mean(), y.std(ddof=1)
        theta_o, H_o = O.fit_newton(O.OuterBase(om, x), terms_o, (y - cent) / sca)
        k = min(pred_rows, n_local)
        xnew, _ = O.synth_xy(43, 0, k, kinds)
        want = cent + sca * O.predict_mean(om, terms_o, theta_o, xnew)
        out["oracle_seconds"] = time.perf_counter() - t0
    runs = {}
    for name, rot in (("own_eigensolver", None),
                      ("shared_rotation", (om.rotmat, om.basisvar, om.maxlevel))):
        h = HotPath(kinds, knots, p, n_local, rank=rank, world=world, backend="newton", row0=row0,
                    n_total=rows, transport=transport, rotation=rot)
        h.setup()
        h.step()
        torch.cuda.synchronize()
        k = min(pred_rows, n_local)
        runs[name] = dict(terms=h.terms.copy(), mean=h.mean[:k].cpu().numpy(),
                          theta=h.theta.cpu().numpy(), cent=h.y_cent, sca=h.y_sca,
                          H_upper=None, diagH=h.diagH.cpu().numpy(), H_full_err=None)
        if rank == 0:
            # H above its diagonal 128 x 128 blocks (the Cholesky factor overwrote the lower
            # triangle and those blocks); the diagonal of H is kept aside by the fit
            runs[name]["H_upper"] = h.G.cpu().numpy()
            # ALL of H, diagonal blocks included: formed once more by obhip_gram_dev into a buffer
            # of its own (one rank: the rows of this rank are all rows)
            if world == 1 and np.array_equal(runs[name]["terms"], terms_o):
                runs[name]["H_full_err"] = h.hessian_full_rel_err(H_o)
        h.close()
        del h
        torch.cuda.empty_cache()
    if rank != 0:
        return None
    terms = runs["shared_rotation"]["terms"]
    out["terms_equal_oracle_selection"] = bool(np.array_equal(terms, terms_o))
    out["terms_equal_between_eigensolvers"] = bool(np.array_equal(terms, runs["own_eigensolver"]["terms"]))
    blk = np.arange(p) // 128
    iu = np.nonzero(blk[None, :] > blk[:, None])
    for name, r in runs.items():
        e = {"predict_max_rel_err": float(np.max(np.abs(r["mean"] - want)) / np.max(np.abs(want)))}
        if name == "shared_rotation" or out["terms_equal_between_eigensolvers"]:
            e["hessian_max_rel_err"] = float(max(
                np.max(np.abs(r["H_upper"][iu] - H_o[iu])) if len(iu[0]) else 0.0,
                np.max(np.abs(r["diagH"] - np.diag(H_o)))) / np.max(np.abs(H_o)))
            e["hessian_full_max_rel_err"] = r["H_full_err"]
            e["theta_max_rel_err"] = float(np.max(np.abs(r["theta"] - theta_o)) / np.max(np.abs(theta_o)))
        out[name] = e
    out["hessian_full_max_rel_err"] = out["shared_rotation"].get("hessian_full_max_rel_err")
    # the figure the tier asks for: device fit + predict vs the reference path's restatement on
    # the same inputs with NOTHING shared
    out["theta_vs_oracle_rows"] = out["own_eigensolver"]["predict_max_rel_err"]
    return out


def run_config(HotPath, label, kinds, knots, p, rows, rank, world, transport, steps, sync, torch, dist,
               _lib, real_world):
    """One of the other BASELINE.json configurations through the same timed loop (every rank
    takes part): rows = rows of THIS rank; world = ranks the fit sums over (virtual ones with
    --sim-ranks), real_world = processes."""
    n_total = rows * world
    h = HotPath(kinds, knots, p, rows, rank=rank, world=world, backend="newton", row0=rank * rows,
                n_total=n_total, transport=transport)
    h.setup()
    e, ps = timed_steps(h, steps, 1, sync, torch, dist, real_world)
    prof = kernel_profile(h, _lib, torch, nprof=1)
    resid = h.newton_residual_rel()
    out = {"workload": label, "scaling": "weak", "value": rows * real_world * steps / e, "unit": "points/s",
           "ms_per_step": e / steps * 1e3, "median_step_ms": statistics.median(ps),
           "steps": steps, "warmup": 1, "rows_per_gpu": rows, "d": len(kinds), "p": p,
           "basis_columns": h.ncols, "terms_nnz": h.terms_info["nnz_total"],
           "newton_residual_rel": resid,
           "kernels_ms": {k: round(v["ms_per_step"], 4) for k, v in prof.items()}}
    if "gram" in prof:
        ach = float(rows) * p * (p + 1) / (prof["gram"]["avg_ms"] * 1e-3) / 1e12
        out["gram"] = {"avg_launch_ms": prof["gram"]["avg_ms"], "launches_per_step": prof["gram"]["launches"],
                       "tflops": ach, "mfma_frac": ach / FP64_MFMA_PEAK_TFLOPS}
    if "cholesky" in prof:
        out["cholesky_ms"] = prof["cholesky"]["ms_per_step"]
        out["backsolve_ms"] = prof.get("backsolve", {}).get("ms_per_step")
    h.close()
    del h
    torch.cuda.empty_cache()
    _lib.call("obhip_trim_pool")
    return out


def host_threads():
    """Threads the CPU leg runs on: the reference takes omp_get_num_procs()
    (src/modandbase.cpp:464); inside a container that is the cores this process may use --
    its affinity mask, cut by the cgroup's CPU quota where one is set (a GPU box hands each
    lease a share of the host, not the whole host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(math.ceil(float(quota) / float(period)))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(hp, panel):
    """CPU restatement of the reference path timed on the host cores (test infrastructure,
    oracle/): the reference's own loops (outerbase::build, getm_, prodmm_, tprodmm_) in
    C++/OpenMP with the reference's chunk schedule and thread count rule
    (omp_get_num_procs(), src/modandbase.cpp:464) in oracle/ob_cpu.cpp; BLAS/LAPACK (NumPy)
    for basismat.t()*basismat and solve() exactly where the reference hands over to
    Armadillo.  Gram path: ALL n rows, in row panels (the reference's loglik_std would need
    the whole 32.8 GB design matrix at n = 1e6; BASELINE.md section 5: blocked row panels,
    B_panel^T B_panel accumulated), nothing extrapolated.  PCG path (what obfit runs): basis
    build and one B a + one B^T r pass timed at the FULL n, times the passes lpdf::optcg
    makes."""
    import numpy as np
    from threadpoolctl import threadpool_limits, threadpool_info   # hard requirement: the
    # BLAS thread count below is what the `sample` string claims
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ob_oracle as O
    import ob_cpu
    threads = host_threads()
    use_cpp = ob_cpu.available()
    n = hp.n
    panel = max(1, min(panel, n))

    om = O.OuterMod()
    om.setcovfs(hp.kinds)
    om.setknot(O.bench_knots(hp.kinds, hp.m))
    p = hp.p
    t = {"synth": 0.0, "build": 0.0, "getmat": 0.0, "gram": 0.0, "solve": 0.0, "predict": 0.0}
    blas_threads = None

    def run():
        # y standardised over all rows first (R/fitting.R:55-57); generating the synthetic rows
        # is input preparation, not part of the path
        t0 = time.perf_counter()
        ys = np.concatenate([O.synth_xy(42, r0, min(panel, n - r0), hp.kinds)[1]
                             for r0 in range(0, n, panel)])
        cent, sca = ys.mean(), ys.std(ddof=1)
        t["synth"] += time.perf_counter() - t0
        sigma = O.default_sigma((ys - cent) / sca)
        G = np.zeros((p, p))
        g = np.zeros(p)
        for r0 in range(0, n, panel):
            nr = min(panel, n - r0)
            t0 = time.perf_counter()
            x, y = O.synth_xy(42, r0, nr, hp.kinds)
            y = (y - cent) / sca
            t["synth"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            if use_cpp:
                bm, bs = ob_cpu.build(om, x, threads)            # outerbase::build, all knots
            else:
                ob = O.OuterBase(om, x)
                bm, bs = ob.basemat, ob.basescale
            t["build"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            if use_cpp:
                B = ob_cpu.getmat(om, hp.terms, bm, bs, threads)  # getm_ (loglik_std ctor)
            else:
                B = O.getm(hp.terms, bm, bs, om.knotptst)
            t["getmat"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            G += B.T @ B                                          # loglik_std::hess
            g += B.T @ y
            t["gram"] += time.perf_counter() - t0
            del B, bm, bs, x
        t0 = time.perf_counter()
        H = math.exp(-2 * sigma) * G
        H[np.diag_indices_from(H)] += O.prior_prec(om, hp.terms, hp.rho)
        theta = np.linalg.solve(H, math.exp(-2 * sigma) * g)      # fit.cpp:120
        t["solve"] += time.perf_counter() - t0
        for r0 in range(0, n, panel):
            nr = min(panel, n - r0)
            t0 = time.perf_counter()
            xnew, _ = O.synth_xy(43, r0, nr, hp.kinds)
            t["synth"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            if use_cpp:                                           # predictor update + mean
                bmn, bsn = ob_cpu.build(om, xnew, threads)
                ob_cpu.mm(om, hp.terms, bmn, bsn, theta, threads)
            else:
                O.predict_mean(om, hp.terms, theta, xnew)
            t["predict"] += time.perf_counter() - t0
        return theta

    with threadpool_limits(limits=threads):
        blas_threads = sorted({int(i["num_threads"]) for i in threadpool_info()
                               if i.get("user_api") == "blas"})
        theta = run()
    full = t["build"] + t["getmat"] + t["gram"] + t["solve"] + t["predict"]
    impl = "oracle/ob_cpu.cpp (C++/OpenMP, reference chunk schedule)" if use_cpp \
        else "oracle/ob_oracle.py (NumPy)"
    pcg = None
    if use_cpp:
        # the matrix-free path obfit itself takes, at the full n (basemat
        # 6.4 GB): build, then per iteration update() + hessmult() = two B a and two B^T r
        # passes (fit.cpp:71-85), at the iteration count the device PCG needed, then predict
        x, _ = O.synth_xy(42, 0, n, hp.kinds)
        t0 = time.perf_counter()
        bm, bs = ob_cpu.build(om, x, threads)
        tb = time.perf_counter() - t0
        r = np.ones(n)
        t0 = time.perf_counter()
        ob_cpu.mm(om, hp.terms, bm, bs, theta, threads)
        ob_cpu.tmm(om, hp.terms, bm, bs, r, threads)
        tp = time.perf_counter() - t0
        del bm, bs, x
        iters = getattr(hp, "cg_iters", None) or 22
        full_pcg = 2 * tb + (2 * (iters + 1) + 1) * tp   # fit build + predict build; +1 pass: predict
        pcg = {"value": n / full_pcg, "unit": "points/s", "iterations": iters, "rows": n,
               "build_s": tb, "mm_plus_tmm_s": tp,
               "what": "build and one B a + B^T r pass timed at the full n on %d threads; "
                       "2 (iters + 1) such passes per fit as lpdf::optcg makes them" % threads}
    return {"value": n / full, "unit": "points/s", "cores": threads, "blas_threads": blas_threads,
            "kind": "port", "host_cpu_count": os.cpu_count(), "pcg_path": pcg,
            "seconds": dict(t, total=full),
            "sample": "%s + NumPy BLAS/LAPACK (%s threads) for B^T B and solve, %d OpenMP threads "
                      "(affinity / cgroup share of the host), ALL %d rows of the same workload in "
                      "row panels of %d (B_panel^T B_panel accumulated): build %.2fs getmat %.2fs "
                      "gram %.2fs solve %.2fs predict %.2fs; nothing scaled"
                      % (impl, "/".join(map(str, blas_threads)) or "?", threads, n, panel, t["build"],
                         t["getmat"], t["gram"], t["solve"], t["predict"])}


def which_config(d, n_total, rows_per_gpu, p):
    """Name of the BASELINE.json configuration these sizes are."""
    if (d, n_total, p) == (10, 100000, 1024):
        return "BASELINE.json configs[1]"
    if (d, n_total, p) == (20, 1000000, 4096):
        return "BASELINE.json configs[2]"
    if (d, rows_per_gpu, p) == (20, 1250000, 4096):
        return "BASELINE.json configs[3] (1.25e6 rows per GPU)"
    if (d, p) == (40, 16384):
        return "BASELINE.json configs[4] shape (d=40, p=16384, mixed covariances)"
    return "custom sizes"


def timed_steps(hp, steps, warmup, sync, torch, dist, world):
    """W untimed steps, then K timed steps between barrier + synchronize; -> (elapsed of the
    K steps, max over ranks; per-step times from events on the launch stream)."""
    for _ in range(warmup):
        hp.step()
    sync()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        hp.step()
        evs[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    per_step = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    return elapsed, per_step


def kernel_profile(hp, _lib, torch, nprof=2):
    """per-kernel hipEvent timing on the launch stream (obhip_profile_*), untimed region"""
    _lib.call("obhip_profile_reset")
    _lib.call("obhip_profile_enable", 1)
    for _ in range(nprof):
        hp.step()
    torch.cuda.synchronize()
    prof = {}
    for name in ["build_basis", "materialize_B", "gram", "gram_reduce", "tmm", "mm", "sqtmm", "tmm_dual", "hessmult",
                 "exchange", "unpack_form", "form_hessian", "cholesky", "backsolve", "predict"]:
        cnt, ms = C.c_uint64(0), C.c_double(0)
        _lib.call("obhip_profile_get", name.encode(), C.byref(cnt), C.byref(ms))
        if cnt.value:
            prof[name] = dict(launches=cnt.value, avg_ms=ms.value / cnt.value,
                              ms_per_step=ms.value / nprof)
    _lib.call("obhip_profile_enable", 0)
    return prof


def secondary_rooflines(hp, prof, _lib, np):
    """Achieved fractions of the kernels beside the Gram, from the live hipEvent averages and
    the algorithmic bytes / flops of DESIGN.md section 4 (per point: build reads 8 d and
    writes 8 (Mc + 1); a product pass reads 8 (Mc + 1) + 8; the design-matrix copy writes 8 p;
    a term-per-lane pass makes nnz_total 8-byte LDS reads per row)."""
    n, p, d, Mc = float(hp.n), float(hp.p), float(hp.d), float(hp.ncols)
    nnz = float(hp.terms_info["nnz_total"])
    out = {}
    # shared sub-products (csrc/share.cpp): what the star kernels read per row -- wave instructions
    # of 64 lanes x 8 bytes -- when the term set is in their domain (9 .. 16 family star-waves for
    # the one-workgroup kernels, few left-over terms)
    info = np.zeros(11, dtype=np.uint64)
    _lib.call("obhip_terms_share_tables", hp.t._h, info.ctypes.data, None, None, None, None)
    star = 9 <= int(info[7]) and int(info[1]) <= 192 and \
        os.environ.get("OBHIP_SHARE", "1") != "0" and os.environ.get("OBHIP_HM3", "1") != "0"
    star_reads = float(info[2] + info[9])      # family star-waves + the plain star-wave's worth of left-over terms

    def hbm(name, kernel, byts, extra=None):
        if name not in prof:
            return
        ms = prof[name]["avg_ms"]
        e = {"kernel": kernel, "avg_launch_ms": ms, "hbm_GBs": byts / ms / 1e6,
             "hbm_frac": byts / ms / 1e6 / HBM_PEAK_GBS}
        if extra:
            e.update(extra(ms))
        out[name] = e

    def lds(ms, passes=1.0):
        if star:
            b = 512.0 * n * star_reads * passes
            return {"lds_GBs": b / ms / 1e6, "lds_frac": b / ms / 1e6 / LDS_PEAK_GBS,
                    "lds_reads_per_row": star_reads * passes, "lds_reads_per_row_unshared": float(info[3]),
                    "bound": "LDS read latency / VALU issue (neither pipe saturated: profiles/r05_pmc_products.txt)"}
        b = 8.0 * n * nnz
        return {"lds_GBs": b / ms / 1e6, "lds_frac": b / ms / 1e6 / LDS_PEAK_GBS,
                "bound": "lds (term-per-lane column reads)"}
    hbm("build_basis", "k_build_basis", n * 8 * (d + Mc + 1),
        lambda ms: {"bound": "latency (bisection + table reads per dimension; interval tables in LDS)"})
    hbm("materialize_B", "k_materialize_tl", n * 8 * (Mc + 1 + p), lambda ms: {"bound": "hbm write"})
    one_wg = star and int(info[7]) <= 16
    hbm("tmm", "k_star<OP_TMM>" if star else "k_tmm_tl", n * 8 * (Mc + 2), lds)
    hbm("mm", "k_star<OP_MM>" if star else "k_mm_tl", n * 8 * (Mc + 2), lds)
    # the PCG's fused Hessian product / update() pass: one read of the basis; k_star forms every
    # product twice (phase A, phase B), k_hm2 once
    hbm("hessmult", "k_star<OP_HESS / OP_UPDATE>" if one_wg else "k_hm2 (k_hm_tl for terms it does not take)",
        n * 8 * (Mc + 2), (lambda ms: lds(ms, 2.0)) if one_wg else lds)
    hbm("tmm_dual", "k_star<OP_TMM, DUAL>" if star else "k_tmm_tl<DUAL>", n * 8 * (Mc + 2), lds)
    hbm("predict", "k_star_predict" if one_wg else "k_predict_tl", n * 8 * (d + 1), lds)
    if "cholesky" in prof:
        ms = prof["cholesky"]["avg_ms"]
        fl = p ** 3 / 3.0
        out["cholesky"] = {"kernel": "k_chol_panel2 + k_chol_update", "avg_launch_ms": ms,
                           "tflops": fl / ms / 1e9, "mfma_frac": fl / ms / 1e9 / FP64_MFMA_PEAK_TFLOPS,
                           "bound": "latency (p / 64 dependent panel steps)"}
    return out


def config0_obfit(torch, _lib):
    """BASELINE.json configs[0]: Borehole d=8, n=1000, p=256 through obfit + obpred -- the one
    configuration at the scale of the reference's own vignette (vignettes/gettingstarted.Rmd:59-68,
    R/fitting.R:27-155).  Device leg: outerbase_amd.fitting.obfit / obpred (every data-sized step a
    device call), wall time, profiled launches and host round trips.  CPU leg (BASELINE.md section 3,
    C1): oracle/ob_harness.py, the NumPy restatement of the same two-stage flow, on the host cores --
    the stated baseline of this entry, like `cpu_baseline` a checker timed, never the product."""
    import numpy as np
    import outerbase_amd as ob
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ob_oracle as O
    import ob_harness as H
    rng = np.random.default_rng(0)
    n, d, numb = 1000, 8, 256
    x = rng.random((n, d))
    y = O.borehole8d(x)
    xt = rng.random((500, d))
    yt = O.borehole8d(xt)

    def counters():
        a, b, ms = C.c_uint64(0), C.c_uint64(0), C.c_double(0)
        _lib.call("obhip_profile_get", b"*", C.byref(a), C.byref(ms))
        _lib.call("obhip_profile_get", b"host_syncs", C.byref(b), None)
        return a.value, b.value, ms.value
    ob.obfit(x[:200], y[:200], numb=50, seed=0)          # first use: instantiations, pool
    torch.cuda.synchronize()
    _lib.call("obhip_profile_reset")
    _lib.call("obhip_profile_enable", 1)
    l0, s0, _ = counters()
    t0 = time.perf_counter()
    m = ob.obfit(x, y, numb=numb, seed=0)
    t1 = time.perf_counter()
    pred = ob.obpred(m, xt)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    l1, s1, kms = counters()
    _lib.call("obhip_profile_enable", 0)
    rmse = float(np.sqrt(np.mean((pred["mean"] - yt) ** 2)) / np.std(yt))
    out = {"workload": "BASELINE.json configs[0]: Borehole d=8 n=1000 p=256 mat25pow, obfit (two stages, "
                       "BFGS over the hyper-parameters, PCG inside) + obpred (mean and var) at 500 new points",
           "device": {"obfit_s": t1 - t0, "obpred_ms": (t2 - t1) * 1e3, "profiled_launches": l1 - l0,
                      "profiled_kernel_ms": kms, "host_round_trips": s1 - s0,
                      "test_rmse_over_sd": rmse, "var_min": float(np.min(pred["var"]))}}
    del m
    numbr = min(n // 2, numb, 80 * d)
    sub = np.random.default_rng(0).choice(n, size=min(n, 3 * numbr), replace=False)
    t0 = time.perf_counter()
    mo = H.obfit(x, y, numb, ["mat25pow"] * d, sub)
    t1 = time.perf_counter()
    pm = H.obpred_mean(mo, xt)
    out["cpu"] = {"kind": "port", "what": "oracle/ob_harness.py (NumPy restatement of R/fitting.R:27-137 and the "
                  "lpdf classes under it)", "obfit_s": t1 - t0, "cores": host_threads(),
                  "test_rmse_over_sd": float(np.sqrt(np.mean((pm - yt) ** 2)) / np.std(yt)),
                  "bfgs_iterations": [len(t) for t in mo["traces"]]}
    return out


def obfit_evaluation(kinds, knots, p, n, torch, _lib, reps=3, maxlev=None, what="the headline rows and terms"):
    """One second-stage function evaluation of obfit (R/fitting.R:123-136 through BFGS_lpdf ->
    .lpdfwrapper, R/optimization.R: updatehyp, updateom, updatepara, lpdf$optcg, then value and
    gradients) on the bench's rows and terms: lpdfvec(loglik_gauss, logpr_gauss) with the marginal
    adjustment on -- the device PCG fit followed by ONE update with the hyper-parameter and
    parameter gradients.  Untimed region of the bench; hipEvent scopes of the library for the
    phases.  The model layer takes host arrays at creation, so the seeded rows go through the
    host once (not timed)."""
    import numpy as np
    import outerbase_amd as ob
    from outerbase_amd.driver import bench_knots, KIND_ID
    d = len(kinds)
    kid = (C.c_int * d)(*[KIND_ID[k] for k in kinds])
    x = torch.empty((d, n), dtype=torch.float64, device="cuda")
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    _lib.call("obhip_synth_xy_dev", 42, 0, n, d, C.cast(kid, C.c_void_p), x.data_ptr(), y.data_ptr())
    torch.cuda.synchronize()
    xh = np.ascontiguousarray(x.cpu().numpy().T)
    yh = y.cpu().numpy()
    del x, y
    yh = (yh - yh.mean()) / yh.std(ddof=1)
    om = ob.outermod()
    ob.setcovfs(om, kinds)
    ob.setknot(om, bench_knots(kinds, knots))
    terms = om.selectterms(p)
    if maxlev is not None:
        # (obfit's term sets at this shape keep fewer levels than selectterms at the default
        # hyper-parameters: its length scales grow and the eigenvalues fall faster)
        t4 = om.selectterms(4 * p)
        terms = t4[t4.max(1) <= maxlev][:p]
        assert len(terms) == p
    lik = ob.loglik_gauss(om, terms, yh, xh)
    pr = ob.logpr_gauss(om, terms)
    vec = ob.lpdfvec(lik, pr)
    hyp = ob.gethyp(om)
    para = np.asarray(ob.getpara(vec), dtype=np.float64)

    def evaluate(i):
        om.updatehyp(hyp + 1e-3 * (i + 1))
        vec.updateom()
        vec.updatepara(para)
        vec.optcg(1e-3, 100)             # BFGS_lpdf's tolerance and step cap (R/optimization.R:160-175)
    evaluate(-1)                         # first use: views, tables, instantiations
    torch.cuda.synchronize()
    _lib.call("obhip_profile_reset")
    _lib.call("obhip_profile_enable", 1)
    t0 = time.perf_counter()
    for i in range(reps):
        evaluate(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    phases = {}
    for name in ("hessmult", "tmm_d3", "sqtmm_gradhyp_dense", "build_basis", "build_basis_grad", "mm", "tmm",
                 "sqtmm", "tmm_dual"):
        cnt, tot = C.c_uint64(0), C.c_double(0)
        _lib.call("obhip_profile_get", name.encode(), C.byref(cnt), C.byref(tot))
        if cnt.value:
            phases[name] = {"launches_per_evaluation": cnt.value / reps, "ms_per_evaluation": round(tot.value / reps, 3)}
    cnt, tot = C.c_uint64(0), C.c_double(0)
    _lib.call("obhip_profile_get", b"*", C.byref(cnt), C.byref(tot))
    profiled = {"launches_per_evaluation": cnt.value / reps, "ms_per_evaluation": round(tot.value / reps, 3)}
    _lib.call("obhip_profile_enable", 0)
    out = {"workload": "one second-stage obfit function evaluation on " + what + ": updatehyp, "
                       "updateom, updatepara, lpdf$optcg (device PCG, tol 1e-3, <= 100 steps), then value, "
                       "%d hyper-parameter and 2 parameter gradients with the marginal adjustment "
                       "(lpdfvec of loglik_gauss and logpr_gauss)" % len(hyp),
           "ms_per_evaluation": ms, "cg_iterations": int(vec.cgiters) if hasattr(vec, "cgiters") else None,
           "phases": phases,
           # every profiled scope of the library together: what is left of ms_per_evaluation is the host
           # (the eigen-model of updatehyp, the interval tables, result copies) and unprofiled vector kernels
           "all_profiled_scopes": profiled,
           "gradhyp_norm": float(np.linalg.norm(np.asarray(vec.gradhyp))),
           "d": d, "p": p, "n": n, "covariance": "/".join(sorted(set(kinds))),
           "factors_per_term_max": int((terms > 0).sum(1).max()), "levels_max": int(terms.max())}
    del vec, lik, pr
    return out


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.sim_ranks and world != 1:
        sys.exit("bench.py: --sim-ranks is a one-process rehearsal (use it with --gpus 1)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (libobhip has no CPU fallback)")
    # one rank per GPU; OBHIP_DIST_BACKEND=gloo lets several ranks rehearse the N > 1 path
    # on a single GPU (libobhip's host transport stages the exchange buffer through gloo)
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

    import outerbase_amd as ob  # noqa: F401
    from outerbase_amd import _lib
    from outerbase_amd.driver import HotPath, shard_rows

    kinds = [k.strip() for k in args.kinds.split(",")]
    kinds = [kinds[i % len(kinds)] for i in range(args.d)]
    n_total = args.n * world if args.weak else args.n
    row0, n_local = shard_rows(rank, world, n_total)
    # --sim-ranks N: this process is rank 0 of N virtual ranks that all hold its n_local rows
    vworld, transport = (args.sim_ranks, "sim") if args.sim_ranks > 1 else (world, None)
    hp = HotPath(kinds, args.knots, args.p, n_local, rank=rank, world=vworld, backend=args.backend,
                 row0=row0, n_total=n_total * (vworld // world), transport=transport)
    hp.setup()
    _lib.call("obhip_set_gram_backend", args.gram_backend)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # the first contact with the transport verifies itself: closed-form sums of a buffer of the
    # real exchange size through both RCCL paths, compared on the device; a wrong
    # reduce-scatter / all-gather pair switches this communicator to ncclAllReduce in-process,
    # a wrong all-reduce ends the run with a message (obhip_comm_selftest_dev)
    selftest = hp.comm_selftest()
    if world > 1 or transport:
        # every rank, on stderr: a failed first RCCL contact on a multi-GPU node must be diagnosable
        # from that rank's log alone (n8.err of the driver's scaling run)
        ci = hp.comm_info()
        sys.stderr.write("[bench rank %d/%d] exchange: transport=%s path=%s ranks=%s rccl_ranks=%s "
                         "rccl_version=%s bytes_per_fit=%s selftest=%s\n"
                         % (rank, world, ci.get("transport"), ci.get("path"), ci.get("ranks"),
                            ci.get("rccl_ranks"), ci.get("rccl_version"), ci.get("bytes_per_fit"),
                            json.dumps(selftest)))
        sys.stderr.flush()

    elapsed, per_step = timed_steps(hp, args.steps, args.warmup, sync, torch, dist, world)
    prof = kernel_profile(hp, _lib, torch)

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
        split[name + "_only_points_per_s"] = float(n_total) / dt

    # obpred returns mean AND var (R/fitting.R:149-155; pred_gauss: var = B^2 (1 / diag H) + e^{2 sigma},
    # loglik_gauss.cpp:223-227): the fused predictor with the variance on, same rows
    if args.backend == "newton":
        cv = 1.0 / hp.diagH
        var = torch.empty_like(hp.mean)

        def predict_with_var():
            _lib.call("obhip_predict_dev", hp.om._h, hp.t._h, hp.theta.data_ptr(), hp.xnew.data_ptr(), hp.n,
                 hp.mean.data_ptr(), cv.data_ptr(), hp.sigma, var.data_ptr())
        predict_with_var()
        sync()
        t0 = time.perf_counter()
        for _ in range(2):
            predict_with_var()
        sync()
        tt = torch.tensor([(time.perf_counter() - t0) / 2], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        split["predict_with_var_ms"] = float(tt.item()) * 1e3
        split["predict_var_min"] = float(var.min().item())
        del cv, var

    # what a host-buffer caller pays on top: x, y, xnew in, mean out over PCIe
    # (pinned buffers); reported beside `value`, never part of it
    pcie = None
    if rank == 0:
        hx = torch.empty(hp.x.shape, dtype=torch.float64).pin_memory()
        hy = torch.empty(hp.n, dtype=torch.float64).pin_memory()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hp.x.copy_(hx, non_blocking=True)
        hp.xnew.copy_(hx, non_blocking=True)
        hp.y_raw.copy_(hy, non_blocking=True)
        hy.copy_(hp.mean, non_blocking=True)
        torch.cuda.synchronize()
        pcie = (time.perf_counter() - t0) * 1e3
        hp.setup_inputs()          # restore the synthetic inputs the copies overwrote
        del hx, hy
    hp.step()                      # every rank: the fit sums over ranks
    sync()

    # parity of THIS run (untimed).  Every rank: Newton stationarity of the device theta with H
    # applied matrix-free and summed through the communicator (independent of Gram / Cholesky).
    # Rank 0: its predictions against the oracle's basis (test infrastructure, oracle/).
    parity = None
    resid = hp.newton_residual_rel() if args.backend == "newton" else None
    if rank == 0:
        parity = check_against_oracle(hp)
        if resid is not None:
            parity["newton_residual_rel"] = resid

    # the other back end on the same inputs, for the record (untimed region; every rank
    # takes part because the fit sums over ranks): B = matrix-free PCG, what obfit() runs
    alt = None
    if args.backend == "newton" and not args.no_alt_backend:
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
               "points_per_s": float(n_total) / dt, "cg_iterations": hp.cg_iters,
               "max_rel_diff_of_predictions_vs_newton": rel}
        hp.backend = "newton"
        hp.theta.copy_(theta_newton)
        hp.mean.copy_(mean_newton)

    comm_info = hp.comm_info()
    if args.dump and rank == 0:
        np.savez(args.dump, theta=hp.theta.cpu().numpy(), mean=hp.mean[:1000].cpu().numpy(),
                 meansd=hp.meansd.cpu().numpy(), n_total=n_total, world=world)

    # the headline run's buffers are released before the other workloads
    hp.close()
    for name in ("x", "xnew", "y_raw", "y", "mean", "G", "exbuf", "ws"):
        setattr(hp, name, None)
    torch.cuda.empty_cache()
    _lib.call("obhip_trim_pool")

    # Gram + Cholesky + predict against the oracle's own fit on the first rows (every rank takes
    # part in the device fits, rank 0 runs the oracle)
    fitpar = None
    if not args.no_fit_parity and args.backend == "newton" and not transport:
        fitpar = fit_vs_oracle(HotPath, shard_rows, kinds, args.knots, args.p,
                               min(args.fit_parity_rows, n_total), rank, world, None)
        if rank == 0:
            parity["fit_vs_oracle"] = fitpar
            parity["theta_vs_oracle_rows"] = fitpar["theta_vs_oracle_rows"]

    # The other BASELINE.json configurations that fit one GPU, driver-timed in the same line
    # (measured after the headline so that they cannot disturb it; every rank takes part):
    # configs[1] at full size per GPU, configs[3]'s 1.25e6 rows per GPU, configs[4]'s 125 000-row
    # shard per GPU (= configs[4] itself at 8 GPUs).
    others = []
    config3 = None
    headline = (args.d, args.p, args.knots) == (20, 4096, 40) and args.backend == "newton"
    if headline and not args.no_configs:
        others.append(run_config(
            HotPath, "BASELINE.json configs[1]: d=10 n=1e5 p=1024 mat25, 40 knots/dim (per GPU)",
            ["mat25"] * 10, 40, 1024, 100_000, rank, vworld, transport, 5, sync, torch, dist, _lib, world))
    if headline and not args.no_config3:
        rows3 = 1_250_000
        config3 = run_config(
            HotPath, "BASELINE.json configs[3]: d=20 n=%d (1.25e6 rows per GPU x %d) p=4096"
            % (rows3 * vworld, vworld), kinds, args.knots, args.p, rows3, rank, vworld, transport, 3,
            sync, torch, dist, _lib, world)
        others.append(config3)
    if headline and not args.no_configs:
        mixed = [("mat25", "mat25pow", "mat25ang")[i % 3] for i in range(40)]
        others.append(run_config(
            HotPath, "BASELINE.json configs[4] shard: d=40 p=16384 mat25/mat25pow/mat25ang cyclic, "
                     "125 000 rows per GPU (n=1e6 at 8 GPUs)",
            mixed, 40, 16384, 125_000, rank, vworld, transport, 3, sync, torch, dist, _lib, world))

    # obfit's function evaluation on the same problem (north_star's entry point runs these by the
    # hundred): one process, after everything that is timed
    obfit_eval = obfit_eval_pow = config0 = None
    if headline and world == 1 and not transport and not args.no_obfit_eval:
        torch.cuda.empty_cache()
        _lib.call("obhip_trim_pool")
        obfit_eval = obfit_evaluation(kinds, args.knots, args.p, n_total, torch, _lib)
        torch.cuda.empty_cache()
        _lib.call("obhip_trim_pool")
        # obfit's DEFAULT covariance (R/fitting.R:66, listcov R/outersupport.R:195-226): mat25pow, two
        # hyper-parameters per dimension, terms of up to six factors at d = 8
        obfit_eval_pow = obfit_evaluation(
            ["mat25pow"] * 8, 40, 4096, n_total, torch, _lib, maxlev=12,
            what="%d seed-42 rows of the d = 8 Borehole surface with obfit's default covariance "
                 "(mat25pow), selectterms' 4096 terms of at most 12 levels" % n_total)
        torch.cuda.empty_cache()
        _lib.call("obhip_trim_pool")
        if not args.no_cpu_baseline:
            config0 = config0_obfit(torch, _lib)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    p = args.p
    pts = float(n_total) * args.steps
    ms_per_step = elapsed / args.steps * 1e3
    out = {
        "metric": "fit+predict points/sec, d=%d n=%g p=%d" % (args.d, n_total, p),
        "value": pts / elapsed,
        "unit": "points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "median_step_ms": statistics.median(per_step),
        "value_at_median_step": float(n_total) / (statistics.median(per_step) * 1e-3),
        "step_ms": per_step,
        "higher_is_better": True,
        "scaling": "weak" if args.weak else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "%s: d=%d n=%d (%d rows on each of %d GPU(s)) p=%d, %s, %d knots/dim, "
                        "fit (%s) + predict on n fresh rows"
                        % (which_config(args.d, n_total, n_local, p), args.d, n_total, n_local,
                           world, p, "/".join(sorted(set(kinds))), args.knots,
                           "Gram+Cholesky" if args.backend == "newton" else "PCG"),
            "backend": args.backend,
            "rows_total": n_total, "rows_per_gpu": n_local, "d": args.d, "p": p,
            "terms_nnz": hp.terms_info["nnz_total"], "basis_columns": hp.ncols,
            "parallelism": ("one rank: nothing to exchange (y standardised first, one B^T y pass)"
                            if world == 1 else
                            "rows sharded over %d rank(s); per fit 24 bytes for mean / sd of y and one "
                            "exchange buffer (packed triangle of G, B^T y)" % world),
        },
        "exchange": dict(comm_info, allreduce_ms=prof.get("exchange", {}).get("avg_ms"),
                         selftest_result=selftest),
        "fit_predict_split": split,
        "host_buffer_overhead": None if pcie is None else {
            "pcie_ms_per_step": pcie,
            "what": "x, xnew, y host->device and mean device->host, pinned, rank 0's shard",
            "points_per_s_including_copies": float(n_total) / (ms_per_step * 1e-3 + pcie * 1e-3)},
        "kernels_ms": prof,
        "parity_check": parity,
        "alt_backend": alt,
        "config3": config3,
        "configs": others,
        "obfit_eval": obfit_eval,
        "obfit_eval_mat25pow_d8": obfit_eval_pow,
    }
    if config0 is not None:
        out["configs"] = [config0] + out["configs"]
    if transport:
        out["sim_ranks"] = vworld
        out["config"]["parallelism"] = (
            "REHEARSAL on one GPU: %d virtual ranks that all hold this GPU's %d rows "
            "(obhip_comm_init_sim: every sum is one device pass buf *= ranks); the step is what one "
            "rank of a %d-GPU job runs without the wire, `value` counts this GPU's rows only"
            % (vworld, n_local, vworld))
    if "gram" in prof and args.backend == "newton":
        flops = float(n_local) * p * (p + 1)  # p(p+1) flop per point
        ach = flops / (prof["gram"]["avg_ms"] * 1e-3) / 1e12
        # HBM-side bytes per launch and the matrix-pipe utilisation come from separate
        # rocprofv3 --pmc passes on this exact workload (profiles/); see the files for the
        # commands and the gfx950 corrections.  They are printed only when the profile was
        # taken from the Gram kernel that is running (content hash of its sources, compiled
        # into the library); otherwise null and "traffic_stale": true.
        traffic = mfma_util = l2_hit = profile_sha = None
        stale = False
        kernel = {0: "k_atb_dma2", 3: "k_gram_mfma4", 4: "k_atb_dma2"}.get(
            args.gram_backend, "k_atb_dma2")
        lib_sha = _lib.lib.obhip_source_hash(1).decode()
        import glob
        tfs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_gram_traffic.json")))
        if tfs:
            tj = json.load(open(tfs[-1]))   # the latest round's PMC passes
            c = tj["config"]
            if tj["kernel"] == kernel and \
                    (c["d"], c["rows"], c["p"], c["knots"]) == (args.d, n_local, p, args.knots):
                profile_sha = tj.get("source_hash_gram")
                if profile_sha == lib_sha:
                    traffic = tj["traffic_bytes_per_launch"]
                    mfma_util = tj.get("mfma_util")
                    l2_hit = tj.get("l2_hit_rate")
                else:
                    stale = True
        out["roofline"] = {"bound": "mfma", "kernel": kernel, "achieved": ach,
                           "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                           "traffic_unit": "bytes per launch (PMC, separate pass)",
                           "traffic_stale": stale, "profile_sha": profile_sha, "library_sha": lib_sha,
                           "mfma_util": mfma_util, "l2_hit_rate": l2_hit,
                           "avg_launch_ms": prof["gram"]["avg_ms"]}
    elif "mm" in prof:
        byts = float(n_local) * 8 * (hp.ncols + 1)
        ach = byts / (prof["mm"]["avg_ms"] * 1e-3) / 1e9
        # the PCG back end's kernels are LDS-bound (profiles/r01_pmc_products.txt); the
        # contract's roofline object only knows hbm | mfma, so this is the HBM view of k_mm_tl
        out["roofline"] = {"bound": "hbm", "kernel": "k_mm_tl", "achieved": ach, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                           "avg_launch_ms": prof["mm"]["avg_ms"],
                           "note": "LDS-bound kernel, see roofline_secondary"}
    out["roofline_secondary"] = secondary_rooflines(hp, prof, _lib, np)
    if not args.no_cpu_baseline and world == 1:
        # the CPU leg: the oracle as the timed baseline (parity_check above is the other place
        # this file touches oracle/, as the checker of this very run)
        if alt and hp.cg_iters is None:
            hp.cg_iters = alt.get("cg_iterations")
        out["cpu_baseline"] = cpu_baseline(hp, args.cpu_panel)
    print(json.dumps(out))
    sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
