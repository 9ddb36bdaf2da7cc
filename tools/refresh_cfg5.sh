# usage (GPU box, repo root): bash tools/refresh_cfg5.sh  -> full GPU suite, config-5 bench lines of the three step variants (default 16x16x32 asm,
# RNNWF_ENGINE=bf16x3-asm32, bf16x3-hipcc), FETCH_SIZE / WRITE_SIZE passes and rocprofv3 stats of the default kernel, all under gpurun_out/r03_n_*
timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/r03_n_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r03_n_tests.log
for e in default bf16x3-asm32 bf16x3-hipcc; do if [ $e = default ]; then unset RNNWF_ENGINE; else export RNNWF_ENGINE=$e; fi; timeout -k 10 200 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-engine > gpurun_out/r03_n_cfg5_$e.json 2> gpurun_out/r03_n_cfg5_$e.err; python -c "
import json
d=json.load(open('gpurun_out/r03_n_cfg5_$e.json')); print('$e', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['mfma_issue_frac'], d['parity']['max_abs_dE_per_site'])
"; done; unset RNNWF_ENGINE
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r03_n_pmc_cfg5_$c -- python3 $R/bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-engine --no-parity > $R/gpurun_out/r03_n_pmc_cfg5_$c.log 2>&1; done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_n_prof_cfg5 -- python3 $R/bench.py --workload cfg5 --steps 4 --warmup 2 --train 3 --no-cpu-baseline --no-alt-engine --no-parity > $R/gpurun_out/r03_n_prof_cfg5.log 2>&1
cat $R/gpurun_out/r03_n_prof_cfg5/*/*kernel_stats.csv | cut -c1-140
python3 - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    agg=collections.defaultdict(list)
    for f in glob.glob("$R/gpurun_out/r03_n_pmc_cfg5_%s/**/*counter_collection.csv"%c, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        if "flip" in k: print(c, k, sum(v)/len(v))
PY
