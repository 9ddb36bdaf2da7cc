#!/bin/bash
# GPU box: pRNN GPU tests, cfg2 bench A/B (ping-pong vs serial bf16x3 kernel), in-kernel stamps of the ping-pong kernel
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out; mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_prnn.py -x -q -m gpu > $out/r02_t_prnn.log 2>&1; echo "pytest rc=$?"; tail -5 $out/r02_t_prnn.log
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt-engine > $out/r02_b_cfg2_pp.json 2> $out/r02_b_cfg2_pp.err; echo "bench pp rc=$?"
RNNWF_ENGINE=bf16x3-serial timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt-engine > $out/r02_b_cfg2_serial.json 2> $out/r02_b_cfg2_serial.err; echo "bench serial rc=$?"
python - <<'PY'
import json,os
for t in ("pp","serial"):
    try:
        r=json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out","r02_b_cfg2_%s.json"%t)))
        print(t, "value %.4g  ms/step %.3f  flip %.3f ms  frac %.3f  mean_E %.6f" % (r["value"], r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"], r["config"]["mean_E"]))
    except Exception as e:
        print(t, "failed", e)
PY
timeout -k 10 200 python tools/stamps.py cfg2 2 > $out/r02_stamps_cfg2.txt 2>&1; echo "stamps rc=$?"; tail -12 $out/r02_stamps_cfg2.txt | tr ';' '\n' | tail -12
