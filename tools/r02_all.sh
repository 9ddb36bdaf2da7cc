#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu > $out/r02_t_all.log 2>&1; echo "pytest all rc=$?"; tail -8 $out/r02_t_all.log
python tools/dbg_crnn_multipass.py 2>&1 | tail -5
for w in cfg2 cfg3; do
for e in "" bf16x3-serial; do
  RNNWF_ENGINE=$e timeout -k 10 200 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-alt-engine > $out/r02_b_${w}_$e.json 2> $out/r02_b_${w}_$e.err; echo "bench $w '$e' rc=$?"
  python - <<PY
import json
try:
    r=json.load(open("$out/r02_b_${w}_$e.json"))
    print("$w '$e'", "value %.4g  ms/step %.3f  kernel %.3f ms  frac %.3f  mean_E %.6f" % (r["value"], r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"], r["config"]["mean_E"]))
except Exception as ex: print("failed", ex)
PY
done
done
