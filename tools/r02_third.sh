#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out; mkdir -p $out
cd $R
timeout -k 10 200 python tools/stamps.py cfg2 3 > $out/r02_stamps_cfg2.txt 2>&1; echo "stamps rc=$?"; cat $out/r02_stamps_cfg2.txt | tr ';' '\n' | tail -30
