cd ${GRAFT_REPO_ROOT:-/root/repo}
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels_ms"]; print("step %.3f ms  cholesky %.3f  backsolve %.3f" % (d["ms_per_step"], k["cholesky"]["avg_ms"], k["backsolve"]["avg_ms"]))'
for p in 1024 2048 3072; do
for np in 1 2; do
  echo "== p $p panels $np"
  OBHIP_CHOL_PANELS=$np python3 bench.py --dims 10 --p $p --rows 100000 --steps 20 --warmup 2 --no-cpu-baseline --no-alt-backend --no-config3 --no-configs --no-fit-parity --no-obfit-eval 2>/dev/null | python3 -c "$show"
done
done
