#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out; mkdir -p $out
cd $R
for a in 1 2 4 8 6 14 15 3; do
  RNNWF_ABLATE=$a timeout -k 10 100 python tools/stamps.py cfg2 6 2>&1 | tail -1
done
timeout -k 10 100 python tools/stamps.py cfg2 2 2>&1 | tail -3 | tr ';' '\n'
