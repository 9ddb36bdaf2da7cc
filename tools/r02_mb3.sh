#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out; mkdir -p $out
cd $R
timeout -k 10 300 tools/microbench/issue_model > $out/r02_issue_model3.txt 2>&1 || { echo "microbench failed"; tail -5 $out/r02_issue_model3.txt; }
grep -E "^(d_only|r_only|k_only|p_only|m_d2|k_roles|k_pp_bar)" $out/r02_issue_model3.txt | cut -c1-150
