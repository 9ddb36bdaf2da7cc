#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > $out/r02_t_all.log 2>&1; echo "pytest all rc=$?"; tail -15 $out/r02_t_all.log
