import sys, os, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from outerbase_amd.driver import HotPath
from test_gpu_fullsize import _fit_is_stationary
n, p = int(sys.argv[1]), int(sys.argv[2])
hp = HotPath(["mat25"] * 20, 40, p, n); hp.setup(); hp.step(); torch.cuda.synchronize()
_fit_is_stationary(hp, 1e-9)
print("ok n", n, "p", p, "terms", hp.terms_info)
