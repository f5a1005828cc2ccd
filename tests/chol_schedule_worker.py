"""Worker of test_cholesky_schedules_with_one_to_eight_panels_per_pass: the panels-per-pass setting
(OBHIP_CHOL_PANELS) is read once per process, so every schedule runs in a process of its own.
Usage: chol_schedule_worker.py <p> [<p> ...]; prints "ok <p> <rel err>" per size."""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

import outerbase_amd as ob  # noqa: E402
from conftest import knots_for  # noqa: E402
from outerbase_amd._lib import call  # noqa: E402

kinds = ["mat25"] * 6
om = ob.outermod()
ob.setcovfs(om, kinds)
ob.setknot(om, knots_for(kinds, 40))
for p in [int(v) for v in sys.argv[1:]]:
    terms = om.selectterms(p)
    t = ob.obmod._Terms(om, terms)
    torch.manual_seed(p)
    A = torch.randn((p, p + 3), dtype=torch.float64, device="cuda")
    G = A @ A.T + 0.5 * torch.eye(p, dtype=torch.float64, device="cuda")
    g = torch.randn(p, dtype=torch.float64, device="cuda")
    sigma, rho = 0.3, 2.0
    e2 = math.exp(-2 * sigma)
    prec = torch.from_numpy(1.0 / (om.getvar(terms) * math.exp(2 * rho))).cuda()
    H = e2 * G + torch.diag(prec)
    want = torch.linalg.solve(H, e2 * g)
    wsb = C.c_uint64(0)
    call("obhip_newton_workspace_bytes", p, C.byref(wsb))
    ws = torch.empty(wsb.value, dtype=torch.uint8, device="cuda")
    th = torch.empty(p, dtype=torch.float64, device="cuda")
    dH = torch.empty(p, dtype=torch.float64, device="cuda")
    Gc = G.clone()
    call("obhip_newton_solve_dev", om._h, t._h, Gc.data_ptr(), g.data_ptr(), sigma, rho, th.data_ptr(),
         dH.data_ptr(), ws.data_ptr(), wsb.value)
    torch.cuda.synchronize()
    cond = float(torch.linalg.cond(H))
    err = float((th - want).norm() / want.norm())
    L = torch.tril(Gc)
    lerr = float((L - torch.linalg.cholesky(H)).norm() / torch.linalg.cholesky(H).norm())
    ok = err < 1e-13 * max(cond, 10.0) and lerr < 1e-13 * max(cond, 10.0)
    print(("ok" if ok else "BAD"), p, err, lerr, flush=True)
