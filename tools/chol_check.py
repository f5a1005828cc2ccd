"""Cholesky solve of libobhip against torch.linalg.solve on a random SPD system (debug aid)."""
import sys, os, math, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from outerbase_amd.driver import HotPath
from outerbase_amd._lib import call
for p in [int(v) for v in sys.argv[1:]]:
    kinds = (["mat25", "mat25pow", "mat25ang"] * 14)[:40]
    hp = HotPath(kinds, 40, p, 6400); hp.setup()
    torch.manual_seed(0)
    A = torch.randn((p, 512), dtype=torch.float64, device="cuda")
    G = A @ A.T + 10.0 * torch.eye(p, dtype=torch.float64, device="cuda")
    g = torch.randn(p, dtype=torch.float64, device="cuda")
    sigma, rho = 0.0, 20.0
    prec = torch.from_numpy(1.0 / (hp.om.getvar(hp.terms) * math.exp(2 * rho))).cuda()
    H = G + torch.diag(prec)
    want = torch.linalg.solve(H, g)
    Gc = G.clone(); th = torch.empty(p, dtype=torch.float64, device="cuda"); dH = torch.empty(p, dtype=torch.float64, device="cuda")
    call("obhip_newton_solve_dev", hp.om._h, hp.t._h, Gc.data_ptr(), g.data_ptr(), sigma, rho, th.data_ptr(), dH.data_ptr(), hp.ws.data_ptr(), hp.wsb)
    torch.cuda.synchronize()
    # L from the lower triangle of Gc
    L = torch.tril(Gc)
    Lt = torch.linalg.cholesky(H)
    bad = ((L - Lt).abs() > 1e-8 * Lt.abs().max()).nonzero()
    print("p", p, "theta rel err", float((th - want).norm() / want.norm()), "L rel err", float((L - Lt).norm() / Lt.norm()),
          "first bad", bad[0].tolist() if bad.numel() else None, "n bad", bad.shape[0])
    d = (th - want).abs()
    badi = (d > 1e-8 * want.abs().max()).nonzero().flatten()
    print("   theta bad count", badi.numel(), "first", badi[:5].tolist(), "last", badi[-5:].tolist())
    hp.close()
