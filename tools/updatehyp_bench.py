"""Host time of om$updatehyp (outermod::build: one eigen-problem per dimension, dealt to host
threads) against the number of dimensions -- does the thread pool of csrc/model.cpp help on this box?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import outerbase_amd as ob
from outerbase_amd.driver import bench_knots

print("cpus of this process: %d" % len(os.sched_getaffinity(0)))
for d, kind, m in ((1, "mat25pow", 40), (2, "mat25pow", 40), (8, "mat25pow", 40), (20, "mat25", 40),
                   (40, "mat25", 40), (8, "mat25pow", 70)):
    kinds = [kind] * d
    om = ob.outermod()
    ob.setcovfs(om, kinds)
    ob.setknot(om, bench_knots(kinds, m))
    h = ob.gethyp(om)
    om.updatehyp(h)
    t0, c0 = time.perf_counter(), time.process_time()
    for i in range(20):
        om.updatehyp(h + 0.001 * (i + 1))
    print("d = %2d %-9s %d knots: updatehyp %6.2f ms wall, %6.2f ms cpu" %
          (d, kind, m, (time.perf_counter() - t0) / 20 * 1e3, (time.process_time() - c0) / 20 * 1e3))
