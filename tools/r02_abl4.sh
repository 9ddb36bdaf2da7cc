#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in 0 1 2 3 4 8 16 32 28 60 63; do
  RNNWF_ABLATE=$a timeout -k 10 200 python tools/stamps.py cfg4 3 2>&1 | tail -1
done
