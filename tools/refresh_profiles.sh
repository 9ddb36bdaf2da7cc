#!/bin/bash
# usage (on the GPU box, from the repo root):  tools/refresh_profiles.sh <tag>     e.g.  r01_e
# Writes gpurun_out/<tag>_*: bench lines of the five BASELINE configs, rocprofv3 kernel stats (cfg2, cfg3, cfg4),
# FETCH_SIZE / WRITE_SIZE passes of cfg2 and cfg4 (separate --pmc runs, as the gfx950 guide prescribes).
tag=${1:-r01_x}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out
set -o pipefail
cd $R
for w in cfg1 cfg2 cfg3 cfg4 cfg5; do
  extra="--no-cpu-baseline"; [ $w = cfg2 ] && extra=""
  steps=20; [ $w = cfg5 ] && steps=5; [ $w = cfg4 ] && steps=10
  timeout -k 10 300 python bench.py --workload $w --steps $steps --warmup 3 $extra > $out/${tag}_bench_$w.json 2> $out/${tag}_bench_$w.err || { echo "bench $w failed"; exit 1; }
  echo "bench $w done"
done
for w in cfg2 cfg3; do
  RNNWF_ENGINE=f32 timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-alt-engine > $out/${tag}_bench_${w}_engine_f32.json 2>/dev/null || exit 1
done
cd /tmp && export TMPDIR=/tmp
for w in cfg2 cfg3 cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-alt-engine > $out/${tag}_prof_$w.log 2>&1 || { echo "rocprof $w failed"; exit 1; }
  cp $out/${tag}_prof_$w/*/*kernel_stats.csv $out/${tag}_kernel_stats_$w.csv
  echo "stats $w done"
done
for w in cfg2 cfg4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_${w}_$c -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-alt-engine > $out/${tag}_pmc_${w}_$c.log 2>&1 || { echo "pmc $w $c failed"; exit 1; }
  done
done
python3 - <<PY
import csv, glob, json, collections
out = "$out"; tag = "$tag"
res = {}
for w in ("cfg2", "cfg4"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("%s/%s_pmc_%s_%s/**/*counter_collection.csv" % (out, tag, w, c), recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[w] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open("%s/%s_pmc_raw.json" % (out, tag), "w"), indent=1)
for w in res:
    for k, d in res[w].items():
        if "flip" in k or "base" in k:
            print(w, k, d)
PY
echo refresh done
