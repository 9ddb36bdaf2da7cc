#!/bin/bash
# GPU box: issue-model microbenchmark, SQ counters of the current cfg2 flip kernel, kernel stats of cfg1 / cfg5
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out; mkdir -p $out
cd $R
timeout -k 10 300 tools/microbench/issue_model > $out/r02_issue_model.txt 2>&1 || { echo "microbench failed"; tail -5 $out/r02_issue_model.txt; }
echo "microbench done"
timeout -k 10 400 tools/pmc_base.sh gpurun_out/r02_pmc_cfg2_before > $out/r02_pmc_cfg2_before.log 2>&1 || echo "pmc failed"
echo "pmc done"
cd /tmp && export TMPDIR=/tmp
for w in cfg1 cfg5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r02_prof_$w -- python3 $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-alt-engine > $out/r02_prof_$w.log 2>&1 || { echo "rocprof $w failed"; }
  cp $out/r02_prof_$w/*/*kernel_stats.csv $out/r02_kernel_stats_$w.csv
  echo "stats $w done"
done
