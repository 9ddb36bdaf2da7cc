#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out; mkdir -p $out
cd $R
timeout -k 10 300 tools/microbench/issue_model > $out/r02_issue_model5.txt 2>&1 || { echo "microbench failed"; tail -5 $out/r02_issue_model5.txt; }
grep -E "^(n_|m_only_ref|m_lds|m_only )" $out/r02_issue_model5.txt | cut -c1-150
