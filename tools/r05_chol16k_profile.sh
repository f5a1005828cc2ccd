#!/bin/bash
# Kernel statistics of the p = 16384 factorisation (60 000-row step at d = 40, the configs[4] shape)
# for 2, 4 and 8 panels per trailing pass.  One gpurun call; summaries land in gpurun_out/r05_chol16k.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
out=$R/gpurun_out/r05_chol16k
mkdir -p $out
for np in ${@:-4 8}; do
  export OBHIP_CHOL_PANELS=$np
  rm -rf /tmp/prof_chol
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_chol -o run --output-format csv -- \
    python3 bench.py --rows 60000 --d 40 --p 16384 --kinds mat25,mat25pow,mat25ang --steps 3 --warmup 1 \
    --no-cpu-baseline --no-alt-backend --no-config3 --no-configs --no-fit-parity --no-obfit-eval \
    > $out/line_panels$np.json 2> $out/err_panels$np.txt || exit 1
  f=$(find /tmp/prof_chol -name '*kernel_stats.csv' | head -1)
  cp "$f" $out/kernel_stats_panels$np.csv
  echo "== panels $np"; head -8 $out/kernel_stats_panels$np.csv | cut -c1-200
done
