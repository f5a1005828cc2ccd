#!/bin/bash
# The dense part of the hyper-parameter gradients with the hyper-parameters beyond a multiple of 16
# (d = 20 mat25: 4 of 20) in a 16-block of their own (0), in groups of four in a pass of their own
# (1) or riding along with the last 16 (2): obfit_eval of bench.py on one box.  One gpurun call.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in ("obfit_eval", "obfit_eval_mat25pow_d8"):
    e=d.get(k)
    if e: print("  %s %.2f ms: " % (k, e["ms_per_evaluation"]) + ", ".join("%s %.2f" % (n, v["ms_per_evaluation"]) for n, v in e["phases"].items()))'
for f in 0 1 2; do
  echo "== OBHIP_GE0_FOURS=$f"
  OBHIP_GE0_FOURS=$f timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-backend --no-config3 --no-configs --no-fit-parity 2>/dev/null | python3 -c "$show" || exit 1
done
