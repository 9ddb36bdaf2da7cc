#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out; mkdir -p $out
cd $R
timeout -k 10 300 tools/microbench/issue_model > $out/r02_issue_model4.txt 2>&1 || { echo "microbench failed"; tail -5 $out/r02_issue_model4.txt; }
grep -E "^(g_|q_only|k_roles_g|k_ppg)" $out/r02_issue_model4.txt | cut -c1-150
