#!/bin/bash
# One gpurun call: GPU tests, the full bench line, and rocprofv3 kernel statistics of the same
# command (Newton path) and of the PCG path.  Outputs under gpurun_out/final/.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1 || { tail -5 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
timeout -k 10 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_newton -o p -- \
  python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-backend > $OUT/bench_line_profiled.json 2> $OUT/prof_newton.err || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cg -o p -- \
  python3 $R/bench.py --backend cg --steps 3 --warmup 1 --no-cpu-baseline --no-alt-backend > $OUT/bench_line_cg_profiled.json 2> $OUT/prof_cg.err || exit 1
echo ok
