"""Device-resident fit + predict pipeline (the hot path of BASELINE.json's
north_star) for one rank of a row-sharded job.

Everything stays in HBM: the synthetic rows are generated on the device, the
basis is built there, the Gram and the right-hand side are all-reduced over
ranks with torch.distributed (RCCL over xGMI on a GPU node, gloo in the CPU
tests of the orchestration), the p x p solve is replicated on every rank and
prediction needs no communication.

torch is used for device memory, the current stream and the collective only;
all arithmetic is in libobhip.
"""
import ctypes as C
import math

import numpy as np

from . import _lib, obmod
from ._lib import call

KIND_ID = {"mat25": 0, "mat25pow": 1, "mat25ang": 2}
EXCHANGE_PATH = {0: "none", 1: "ncclReduceScatter + ncclAllGather (in place)", 2: "ncclAllReduce",
                 3: "host callback", 4: "sim (buf *= ranks on the device)"}
DEFAULT_RHO = 6.0  # logpr_gauss.cpp:48


def bench_knots(kinds, m=40):
    """the reference tests' grid 0.001 + 0.025 k
    (tests/testthat/test-obombasic.R:34), scaled to the kernel's domain."""
    out = []
    for kd in kinds:
        g = 0.001 + 0.025 * np.arange(m) if m <= 40 else np.linspace(0.001, 0.976, m)
        if kd == "mat25ang":
            g = g * 6.283185
        out.append(g)
    return out


def shard_rows(rank, world, n_total):
    """(row0, nrows) of the contiguous row block of the counter-based synthetic stream
    owned by `rank` when n_total rows are split over `world` ranks (SURVEY.md section 8e:
    contiguous row blocks, no data exchange)."""
    lo = rank * n_total // world
    hi = (rank + 1) * n_total // world
    return lo, hi - lo


def make_comm(rank, world, transport=None):
    """obhip_comm of this process.  transport "rccl": an RCCL communicator of libobhip's own
    (the id drawn by rank 0 travels over torch.distributed, which is the launcher's control
    plane only); "host": sums go through torch.distributed on host memory (gloo) -- the
    one-GPU rehearsal of the tests; "sim": `world` virtual ranks that all hold this process's
    shard, every sum one device pass (timing of a rank's step without the wire).  Default:
    rccl when torch.distributed runs on nccl."""
    if world == 1:
        return None, None
    if transport == "sim":
        h = C.c_void_p()
        call("obhip_comm_init_sim", C.byref(h), world)
        return h, None
    import torch
    import torch.distributed as dist
    if transport is None:
        transport = "rccl" if dist.get_backend() == "nccl" else "host"
    h = C.c_void_p()
    if transport == "rccl":
        uid = np.zeros(128, dtype=np.uint8)
        if rank == 0:
            call("obhip_comm_unique_id", uid.ctypes.data)
        box = [uid.tobytes()]
        dist.broadcast_object_list(box, src=0)
        uid = np.frombuffer(box[0], dtype=np.uint8).copy()
        call("obhip_comm_init", C.byref(h), world, rank, uid.ctypes.data)
        return h, None

    def _sum(user, host_ptr, count):
        try:
            buf = np.ctypeslib.as_array(C.cast(host_ptr, C.POINTER(C.c_double)), shape=(count,))
            dist.all_reduce(torch.from_numpy(buf))
            return 0
        except Exception:
            return 1
    cb = _lib.HOST_ALLREDUCE_FN(_sum)
    call("obhip_comm_init_host", C.byref(h), world, rank, C.cast(cb, C.c_void_p), None)
    return h, cb   # the caller keeps cb alive as long as the communicator


class HotPath:
    """One rank of the row-sharded fit + predict job.  n = rows of THIS rank, row0 = its first
    row in the synthetic stream, n_total = rows of all ranks."""

    def __init__(self, kinds, knots_per_dim, p, n, rank=0, world=1, backend="newton",
                 seed_train=42, seed_pred=43, rho=DEFAULT_RHO, cg_tol=1e-10, cg_maxit=None,
                 row0=None, n_total=None, transport=None, rotation=None, terms=None):
        self.kinds = list(kinds)
        self.d = len(kinds)
        self.m = knots_per_dim
        self.p = p
        self.n = n
        self.rank, self.world = rank, world
        self.row0 = rank * n if row0 is None else row0
        self.n_total = n * world if n_total is None else n_total
        self.backend = backend
        self.seed_train, self.seed_pred = seed_train, seed_pred
        self.rho = rho
        self.cg_tol = cg_tol
        self.cg_maxit = cg_maxit
        self.transport = transport
        # parity runs: (rotmat, basisvar, maxlevel) of another eigensolver to inject into the
        # model (None: the library's own Jacobi solver), and a term set to use instead of the
        # model's own selectterms(p)
        self.rotation = rotation
        self.terms_in = terms
        self.basis = None
        self.comm = None
        self.cg_iters = None

    # -- one-time setup (not timed): model, terms, synthetic inputs in HBM ----------
    def setup(self):
        import torch
        self.torch = torch
        dev = torch.device("cuda", torch.cuda.current_device())
        call("obhip_set_device", torch.cuda.current_device())
        call("obhip_set_stream", C.c_void_p(torch.cuda.current_stream().cuda_stream))
        om = obmod.outermod()
        obmod.setcovfs(om, self.kinds)
        obmod.setknot(om, bench_knots(self.kinds, self.m))
        self.om = om
        if self.rotation is not None:
            om.set_rotation(*self.rotation)
        # deterministic: identical on every rank
        self.terms = om.selectterms(self.p) if self.terms_in is None else np.asarray(self.terms_in)
        self.t = obmod._Terms(om, self.terms)
        self.terms_info = self.t.info()
        self.caps = self.t.maxlevels()
        self.ncols = int(1 + self.caps.sum())
        n, d, p = self.n, self.d, self.p
        f64 = torch.float64
        self.x = torch.empty((d, n), dtype=f64, device=dev)      # column-major n x d
        self.xnew = torch.empty((d, n), dtype=f64, device=dev)
        self.y_raw = torch.empty(n, dtype=f64, device=dev)
        self.y = torch.empty(n, dtype=f64, device=dev)
        self.mean = torch.empty(n, dtype=f64, device=dev)
        self.G = torch.empty((p, p), dtype=f64, device=dev)
        self.g = torch.empty(p, dtype=f64, device=dev)
        self.theta = torch.zeros(p, dtype=f64, device=dev)
        self.diagH = torch.empty(p, dtype=f64, device=dev)
        self.meansd = torch.zeros(3, dtype=f64, device=dev)
        wsb = C.c_uint64(0)
        call("obhip_newton_workspace_bytes", p, C.byref(wsb))
        self.ws = torch.empty(wsb.value, dtype=torch.uint8, device=dev)
        self.wsb = wsb.value
        cnt = C.c_uint64(0)
        call("obhip_fit_newton_count", p, self.world, C.byref(cnt))
        self.ex_count = cnt.value
        # the exchange buffer of a row-sharded fit [packed triangle of G | B^T y | zero padding];
        # one rank exchanges nothing (the Gram reduction writes H itself)
        self.exbuf = torch.zeros(self.ex_count, dtype=f64, device=dev) if self.world > 1 else None
        self.setup_inputs()
        self.comm, self._comm_cb = make_comm(self.rank, self.world, self.transport)

    def comm_info(self):
        if self.comm is None:
            return {"transport": "none", "ranks": 1}
        nr, rk, tr, rr, rv = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        call("obhip_comm_info", self.comm, C.byref(nr), C.byref(rk), C.byref(tr), C.byref(rr),
             C.byref(rv))
        path, st = C.c_int(), C.c_int()
        call("obhip_comm_exchange_path", self.comm, self.ex_count, C.byref(path), C.byref(st))
        return {"transport": {1: "rccl", 2: "host", 3: "sim (virtual ranks, device pass)"}[tr.value],
                "path": EXCHANGE_PATH[path.value],
                "selftest": {0: "not run", 1: "passed", 2: "passed after switching to ncclAllReduce"}[st.value],
                "ranks": nr.value, "rccl_ranks": rr.value, "rccl_version": rv.value,
                "bytes_per_fit": 8 * self.ex_count + 24}

    def comm_selftest(self):
        """obhip_comm_selftest_dev on a buffer of the real exchange size (collective: every rank
        calls it).  Raises when the transport returns wrong sums; switches the communicator to
        the plain all-reduce in-process when only the reduce-scatter / all-gather pair does."""
        if self.comm is None:
            return None
        res = (C.c_int64 * 4)()
        call("obhip_comm_selftest_dev", self.comm, self.ex_count, C.cast(res, C.c_void_p))
        return {"elements": int(self.ex_count), "path_in_use": EXCHANGE_PATH[int(res[0])],
                "pair_mismatches": int(res[1]), "allreduce_mismatches": int(res[2]),
                "switched_to_allreduce": bool(res[3])}

    def newton_residual_rel(self):
        """|| H theta - e^{-2 sigma} B^T y || / || e^{-2 sigma} B^T y || with H applied
        MATRIX-FREE (k_mm_tl, k_tmm_tl summed over the ranks through the communicator):
        independent of the Gram and Cholesky kernels.  Collective: every rank calls it."""
        torch = self.torch
        e2 = math.exp(-2 * self.sigma)
        tmp = torch.empty(self.n, dtype=torch.float64, device=self.x.device)
        hv = torch.empty(self.p, dtype=torch.float64, device=self.x.device)
        call("obhip_basis_mm_dev", self.basis, self.t._h, self.theta.data_ptr(), tmp.data_ptr(), 0)
        call("obhip_basis_tmm_dev", self.basis, self.t._h, tmp.data_ptr(), hv.data_ptr(), 0)
        if self.comm is not None:
            call("obhip_comm_allreduce_dev", self.comm, hv.data_ptr(), self.p)
        torch.cuda.synchronize()
        theta = self.theta.cpu().numpy()
        prec = 1.0 / (self.om.getvar(self.terms) * math.exp(2 * self.rho))
        lhs = e2 * hv.cpu().numpy() + prec * theta
        rhs = e2 * self.g.cpu().numpy()
        return float(np.linalg.norm(lhs - rhs) / np.linalg.norm(rhs))

    def hessian_full_rel_err(self, H_want):
        """max |H - H_want| / max |H_want| over ALL entries of H = e^{-2 sigma} B^T B + prior
        (loglik_std.cpp:170-173, logpr_gauss.cpp:153-158) of this rank's rows, formed once more by
        obhip_gram_dev into a buffer of its own: the fit's buffer holds the Cholesky factor in
        its lower triangle and diagonal 128 x 128 blocks.  One rank (parity checks)."""
        torch = self.torch
        G2 = torch.empty((self.p, self.p), dtype=torch.float64, device=self.x.device)
        call("obhip_gram_dev", self.basis, self.t._h, None, G2.data_ptr(), None)
        torch.cuda.synchronize()
        H = math.exp(-2 * self.sigma) * G2.cpu().numpy()
        del G2
        prec = 1.0 / (self.om.getvar(self.terms) * math.exp(2 * self.rho))
        H[np.diag_indices(self.p)] += prec
        return float(np.max(np.abs(H - H_want)) / np.max(np.abs(H_want)))

    def setup_inputs(self):
        """(Re)generate this rank's rows of the synthetic stream in HBM."""
        torch = self.torch
        n, d = self.n, self.d
        kid = (C.c_int * d)(*[KIND_ID[k] for k in self.kinds])
        scratch = torch.empty(n, dtype=torch.float64, device=self.x.device)
        call("obhip_synth_xy_dev", self.seed_train, self.row0, n, d, C.cast(kid, C.c_void_p),
             self.x.data_ptr(), self.y_raw.data_ptr())
        call("obhip_synth_xy_dev", self.seed_pred, self.row0, n, d, C.cast(kid, C.c_void_p),
             self.xnew.data_ptr(), scratch.data_ptr())
        torch.cuda.synchronize()
        del scratch

    # mean and sd of y over all rows: on the device (meansd); the host reads them only where it
    # needs the numbers (checks, reports), never inside a step
    def _pull_standardisation(self):
        cent, sca, ntot = self.meansd.tolist()     # one small D2H (synchronises)
        if int(round(ntot)) != self.n_total:
            raise RuntimeError("ranks disagree on the row count: %r vs %r" % (ntot, self.n_total))
        return cent, sca

    @property
    def y_cent(self):
        return self._pull_standardisation()[0]

    @property
    def y_sca(self):
        return self._pull_standardisation()[1]

    def standardise(self):
        """y = (y - mean) / sd over ALL ranks (R/fitting.R:55-57), two-pass like R's sd(): 24
        bytes cross the ranks, nothing returns to the host."""
        call("obhip_standardise_dev", self.comm, self.y_raw.data_ptr(), self.n, self.y.data_ptr(),
             self.meansd.data_ptr())
        # loglik_std.cpp:51: para0 = log(0.01 * var(y)); var of the standardised y is 1
        self.sigma = math.log(0.01)

    def fit(self):
        torch = self.torch
        call("obhip_set_stream", C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if self.basis is None:
            h = C.c_void_p()
            call("obhip_basis_create_dev", C.byref(h), self.om._h, self.x.data_ptr(), self.n,
                 self.caps.ctypes.data)
            self.basis = h
        else:
            call("obhip_basis_rebuild", self.basis)
        self.standardise()
        if self.backend == "newton":
            # Gram on the matrix cores -> [one sum of the packed triangle + B^T y over the ranks]
            # -> H = e^{-2 sigma} G + prior -> Cholesky, two triangular solves (replicated)
            call("obhip_fit_newton_sharded_dev", self.comm, self.basis, self.t._h, self.om._h,
                 self.y.data_ptr(), self.sigma, self.rho, self.G.data_ptr(), self.g.data_ptr(),
                 self.theta.data_ptr(), self.diagH.data_ptr(),
                 None if self.exbuf is None else self.exbuf.data_ptr(), self.ex_count,
                 self.ws.data_ptr(), self.wsb)
        else:
            self.theta.zero_()
            iters = C.c_uint64(0)
            maxit = self.cg_maxit
            if maxit is None:
                maxit = getsteps(self.p, self.n_total, 1.0 / math.exp(2 * self.sigma))
            call("obhip_fit_cg_dev", self.basis, self.t._h, self.om._h, self.y.data_ptr(),
                 self.sigma, self.rho, self.cg_tol, int(maxit), self.theta.data_ptr(),
                 C.byref(iters), self.diagH.data_ptr(), None, self.comm)
            self.cg_iters = iters.value

    def standardised_targets(self):
        """(y - cent) / sca of this rank's rows (tests, parity checks)."""
        return self.y.clone()

    def predict(self):
        call("obhip_predict_dev", self.om._h, self.t._h, self.theta.data_ptr(),
             self.xnew.data_ptr(), self.n, self.mean.data_ptr(), None, self.sigma, None)
        # obpred: y_cent + y_sca * mean (R/fitting.R:152)
        call("obhip_destandardise_dev", self.mean.data_ptr(), self.n, self.meansd.data_ptr())

    def step(self):
        self.fit()
        self.predict()

    def close(self):
        if self.basis is not None:
            _lib.lib.obhip_basis_destroy(self.basis)
            self.basis = None
        if self.comm is not None:
            _lib.lib.obhip_comm_destroy(self.comm)
            self.comm = None


def getsteps(numb, sampsize, sigtonoiseratio=1e-3, tol=0.001):
    """.getsteps (R/fitting.R:188-195): CG iteration cap used by obfit."""
    r = math.sqrt(numb / sampsize)
    # R: (1 + r)^2 / (1 - r)^2 is Inf at r = 1 and min(1000, Inf) = 1000
    kapp = 1000.0 if r == 1.0 else min(1000.0, (1 + r) ** 2 / (1 - r) ** 2)
    iterest = 0.5 * math.sqrt(kapp) * math.log(2 * sampsize * sigtonoiseratio / tol)
    return int(math.ceil(2 * iterest))
