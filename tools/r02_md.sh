#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_mdrnn.py tests/test_gpu_training.py -q -m gpu -x > $out/r02_t_md.log 2>&1; echo "pytest md rc=$?"; tail -6 $out/r02_t_md.log
timeout -k 10 300 python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline --no-alt-engine > $out/r02_b_cfg4.json 2> $out/r02_b_cfg4.err; echo "bench rc=$?"
python - <<PY
import json
r=json.load(open("$out/r02_b_cfg4.json"))
print("cfg4 value %.4g  ms/step %.3f  flip %.3f ms  frac %.3f  base %.3f ms mean_E %.8f" % (r["value"], r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"], r["roofline"]["base_pass_ms"], r["config"]["mean_E"]))
PY
for a in 0 3 28 32 60 63; do
  RNNWF_ABLATE=$a timeout -k 10 200 python tools/stamps.py cfg4 3 2>&1 | tail -1
done
