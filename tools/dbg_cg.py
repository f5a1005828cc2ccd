import os, sys, tempfile, pathlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for s in ("tests", "", "oracle"):
    sys.path.insert(0, os.path.join(R, s))
import numpy as np
import test_00_two_rank_device as T
d = pathlib.Path(tempfile.mkdtemp())
two = T._run_ranks(d / "w2", 2, "cg", 6001, 300)
one = T._run_ranks(d / "w1", 1, "cg", 6001, 300)[0]
print("iters", [int(r["iters"]) for r in two], int(one["iters"]))
print("theta diff ranks", np.max(np.abs(two[0]["theta"] - two[1]["theta"])))
print("theta diff 2 vs 1", np.max(np.abs(two[0]["theta"] - one["theta"])))
cent, sd, theta_o, want = T._oracle(6001, 300, 500)
print("vs newton: two", np.max(np.abs(two[0]["theta"] - theta_o)), "one", np.max(np.abs(one["theta"] - theta_o)))
m2 = np.concatenate([two[0]["mean"], two[1]["mean"]])
print("pred vs newton: two", np.max(np.abs(m2[:500] - want)), "one", np.max(np.abs(one["mean"][:500] - want)))
