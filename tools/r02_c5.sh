#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in cfg5 cfg1 2d1drnn cfg2; do
timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-alt-engine | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$w product', r['value'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['roofline']['base_pass_ms'])"
done
RNNWF_ABLATE=0 timeout -k 10 300 python tools/stamps.py cfg5 3 2>&1 | tail -1
python -m pytest tests -q -m gpu -x 2>&1 | tail -3
