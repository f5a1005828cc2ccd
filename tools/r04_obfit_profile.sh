#!/bin/bash
# obfit at n = 1e6, p = 4096 (8-d Borehole) under rocprofv3 --kernel-trace --stats: the default path
# (fused hyper-gradient passes, k_tmm_d3) and OBHIP_GRAD_D3=0 (one restricted pass per
# hyper-parameter and store, rounds 1-3).
# Output: gpurun_out/r04_obfit/{d3,views}/obfit_kernel_stats.csv and the run logs.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_obfit
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d3 -o obfit -- python3 $ROOT/tools/obfit_demo.py 1000000 4096 > $OUT/d3.log 2>&1
export OBHIP_GRAD_D3=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/views -o obfit -- python3 $ROOT/tools/obfit_demo.py 1000000 4096 > $OUT/views.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
