#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out
python tools/dbg_crnn_multipass.py 2>&1 | tail -6
python tools/dbg_crnn_multipass.py slp 2>&1 | tail -6
timeout -k 10 600 python -m pytest tests/test_gpu_crnn.py -x -q -m gpu > $out/r02_t_crnn.log 2>&1; echo "pytest crnn rc=$?"; tail -4 $out/r02_t_crnn.log
for e in "" bf16x3-serial; do
  RNNWF_ENGINE=$e timeout -k 10 200 python bench.py --workload cfg3 --steps 20 --warmup 3 --no-cpu-baseline --no-alt-engine > $out/r02_b_cfg3_$e.json 2> $out/r02_b_cfg3_$e.err; echo "bench cfg3 '$e' rc=$?"
  python - <<PY
import json
try:
    r=json.load(open("$out/r02_b_cfg3_$e.json"))
    print("cfg3 '$e'", "value %.4g  ms/step %.3f  swap %.3f ms  frac %.3f  mean_E %.6f" % (r["value"], r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"], r["config"]["mean_E"]))
except Exception as ex: print("failed", ex)
PY
done
