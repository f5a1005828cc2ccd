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


def shard_rows(rank, rows_per_rank):
    """Row block of the counter-based synthetic stream owned by `rank`
    (contiguous row blocks, no data exchange)."""
    return rank * rows_per_rank, rows_per_rank


def global_standardise(local_sum, centre_and_sumsq, n_total, reduce_floats):
    """Mean and sd (n-1 denominator, R/fitting.R:55-57) of a vector sharded over
    ranks, two-pass for accuracy.  local_sum() -> sum of the local shard;
    centre_and_sumsq(cent) subtracts cent from the local shard and returns its sum
    of squares; reduce_floats(list) -> element-wise sums over all ranks."""
    cent = reduce_floats([local_sum()])[0] / n_total
    ss = reduce_floats([centre_and_sumsq(cent)])[0]
    return cent, math.sqrt(ss / (n_total - 1.0))


def merge_normal_equations(G, g, all_reduce):
    """Back end A's only exchange: sum the per-rank Gram and right-hand side
    (SURVEY.md section 8e).  all_reduce(tensor) sums in place over ranks."""
    all_reduce(G)
    all_reduce(g)
    return G, g


class HotPath:
    def __init__(self, kinds, knots_per_dim, p, n, rank=0, world=1, backend="newton",
                 seed_train=42, seed_pred=43, rho=DEFAULT_RHO, cg_tol=1e-10, cg_maxit=None):
        self.kinds = list(kinds)
        self.d = len(kinds)
        self.m = knots_per_dim
        self.p = p
        self.n = n
        self.rank, self.world = rank, world
        self.backend = backend
        self.seed_train, self.seed_pred = seed_train, seed_pred
        self.rho = rho
        self.cg_tol = cg_tol
        self.cg_maxit = cg_maxit
        self.basis = None
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
        self.terms = om.selectterms(self.p)   # deterministic: identical on every rank
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
        self.stats = torch.zeros(2, dtype=f64, device=dev)
        wsb = C.c_uint64(0)
        call("obhip_newton_workspace_bytes", p, C.byref(wsb))
        self.ws = torch.empty(wsb.value, dtype=torch.uint8, device=dev)
        self.wsb = wsb.value
        self.setup_inputs()
        if self.world > 1:
            self._cgbuf = torch.empty(p + 2, dtype=f64, device=dev)

            def _cb(user, d_buf, count):
                import torch.distributed as dist
                try:
                    buf = self._cgbuf[:count]
                    call("obhip_memcpy_d2d", buf.data_ptr(), d_buf, 8 * count)
                    dist.all_reduce(buf)
                    call("obhip_memcpy_d2d", d_buf, buf.data_ptr(), 8 * count)
                    torch.cuda.synchronize()
                    return 0
                except Exception:
                    return 1
            self._cb = _lib.ALLREDUCE_FN(_cb)
        else:
            self._cb = None

    def setup_inputs(self):
        """(Re)generate this rank's rows of the synthetic stream in HBM."""
        torch = self.torch
        n, d = self.n, self.d
        kid = (C.c_int * d)(*[KIND_ID[k] for k in self.kinds])
        row0, _ = shard_rows(self.rank, n)
        scratch = torch.empty(n, dtype=torch.float64, device=self.x.device)
        call("obhip_synth_xy_dev", self.seed_train, row0, n, d, C.cast(kid, C.c_void_p),
             self.x.data_ptr(), self.y_raw.data_ptr())
        call("obhip_synth_xy_dev", self.seed_pred, row0, n, d, C.cast(kid, C.c_void_p),
             self.xnew.data_ptr(), scratch.data_ptr())
        torch.cuda.synchronize()
        del scratch

    def _allreduce(self, t):
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(t)

    # -- y = (y - mean) / sd over ALL ranks (R/fitting.R:55-57) ------------------------
    def standardise(self):
        n_tot = float(self.n * self.world)
        self.y.copy_(self.y_raw)

        def local_sum():
            call("obhip_sum_sumsq_dev", self.y.data_ptr(), self.n, self.stats.data_ptr())
            return float(self.stats[0].item())

        def centre_and_sumsq(cent):
            call("obhip_affine_dev", self.y.data_ptr(), self.n, cent, 1.0)
            call("obhip_sum_sumsq_dev", self.y.data_ptr(), self.n, self.stats.data_ptr())
            return float(self.stats[1].item())

        def reduce_floats(vals):
            if self.world == 1:
                return vals
            t = self.torch.tensor(vals, dtype=self.torch.float64, device=self.stats.device)
            self._allreduce(t)
            return [float(v) for v in t.tolist()]

        cent, sd = global_standardise(local_sum, centre_and_sumsq, n_tot, reduce_floats)
        call("obhip_affine_dev", self.y.data_ptr(), self.n, 0.0, sd)
        self.y_cent, self.y_sca = cent, sd
        # loglik_std.cpp:51: para0 = log(0.01 * var(y)); var of the standardised y is 1
        self.sigma = math.log(0.01)

    def fit(self):
        torch = self.torch
        call("obhip_set_stream", C.c_void_p(torch.cuda.current_stream().cuda_stream))
        self.standardise()
        if self.basis is None:
            h = C.c_void_p()
            call("obhip_basis_create_dev", C.byref(h), self.om._h, self.x.data_ptr(), self.n,
                 self.caps.ctypes.data)
            self.basis = h
        else:
            call("obhip_basis_rebuild", self.basis)
        if self.backend == "newton":
            call("obhip_gram_dev", self.basis, self.t._h, self.y.data_ptr(), self.G.data_ptr(),
                 self.g.data_ptr())
            merge_normal_equations(self.G, self.g, self._allreduce)
            call("obhip_newton_solve_dev", self.om._h, self.t._h, self.G.data_ptr(),
                 self.g.data_ptr(), self.sigma, self.rho, self.theta.data_ptr(),
                 self.diagH.data_ptr(), self.ws.data_ptr(), self.wsb)
        else:
            self.theta.zero_()
            iters, val = C.c_uint64(0), C.c_double(0)
            maxit = self.cg_maxit
            if maxit is None:
                maxit = getsteps(self.p, self.n * self.world, 1.0 / math.exp(2 * self.sigma))
            call("obhip_fit_cg_dev", self.basis, self.t._h, self.om._h, self.y.data_ptr(),
                 self.sigma, self.rho, self.cg_tol, int(maxit), self.theta.data_ptr(),
                 C.byref(iters), self.diagH.data_ptr(), C.byref(val),
                 C.cast(self._cb, C.c_void_p) if self._cb is not None else None, None)
            self.cg_iters = iters.value

    def predict(self):
        call("obhip_predict_dev", self.om._h, self.t._h, self.theta.data_ptr(),
             self.xnew.data_ptr(), self.n, self.mean.data_ptr(), None, self.sigma, None)
        # obpred: y_cent + y_sca * mean (R/fitting.R:152)
        call("obhip_affine_dev", self.mean.data_ptr(), self.n, -self.y_cent / self.y_sca,
             1.0 / self.y_sca)

    def step(self):
        self.fit()
        self.predict()

    def close(self):
        if self.basis is not None:
            _lib.lib.obhip_basis_destroy(self.basis)
            self.basis = None


def getsteps(numb, sampsize, sigtonoiseratio=1e-3, tol=0.001):
    """.getsteps (R/fitting.R:188-195): CG iteration cap used by obfit."""
    r = math.sqrt(numb / sampsize)
    kapp = min(1000.0, (1 + r) ** 2 / (1 - r) ** 2)
    iterest = 0.5 * math.sqrt(kapp) * math.log(2 * sampsize * sigtonoiseratio / tol)
    return int(math.ceil(2 * iterest))
