#!/bin/bash
# Cholesky tuning aid: the solver tests, then the kernel split of a 125 000-row step at
# p = 4096 and of a 60 000-row step at p = 16384 for the settings given as arguments
# ("VAR=value" words, one run each; one gpurun call).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels_ms"]; print("step %.2f ms  cholesky %.3f  backsolve %.3f" % (d["ms_per_step"], k["cholesky"]["ms_per_step"], k["backsolve"]["ms_per_step"]))'
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "newton or cholesky" 2>&1 | tail -2 || exit 1
for setting in "${@:-DEFAULT=1}"; do
  echo "== $setting"
  env $setting timeout -k 10 300 python3 bench.py --rows 125000 --steps 10 --warmup 2 --no-cpu-baseline --no-alt-backend --no-config3 2>/dev/null | python3 -c "$show" || exit 1
  env $setting timeout -k 10 300 python3 bench.py --rows 60000 --d 40 --p 16384 --kinds mat25,mat25pow,mat25ang --steps 3 --warmup 1 --no-cpu-baseline --no-alt-backend --no-config3 2>/dev/null | python3 -c "$show" || exit 1
done
