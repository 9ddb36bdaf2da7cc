#!/bin/bash
# usage (on the GPU box, from the repo root):  tools/refresh_profiles.sh <tag> [bench|prof|pmc|all]     e.g.  r04_z bench
# (two gpurun calls of <= 20 minutes each: "bench" = the bench / train lines, "prof" = rocprofv3 stats and counters)
# Writes gpurun_out/<tag>_*: bench lines of the five BASELINE configs, rocprofv3 kernel stats (cfg2, cfg3, cfg4),
# FETCH_SIZE / WRITE_SIZE passes of cfg2, cfg4 and cfg5 (separate --pmc runs, as the gfx950 guide prescribes).
tag=${1:-r01_x}
part=${2:-all}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out
set -o pipefail
cd $R
if [ $part = all ] || [ $part = bench ]; then
for w in cfg1 cfg2 cfg3 cfg4 cfg5; do
  extra="--no-cpu-baseline"; [ $w = cfg2 ] && extra=""
  steps=20; [ $w = cfg5 ] && steps=5; [ $w = cfg4 ] && steps=10
  timeout -k 10 300 python bench.py --workload $w --steps $steps --warmup 3 $extra > $out/${tag}_bench_$w.json 2> $out/${tag}_bench_$w.err || { echo "bench $w failed"; exit 1; }
  echo "bench $w done"
done
# SURVEY 8 rows f4 / f1 "at speed": the parity-symmetric model, stacked layers, and the gradient leg (--train) of every config
for w in cfg2_parity cfg2_l2 cfg2_l3 cfg3_l2; do
  timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 2 --train 5 > $out/${tag}_bench_$w.json 2> $out/${tag}_bench_$w.err || { echo "bench $w failed"; exit 1; }
done
for w in cfg2 cfg3 cfg4 cfg5; do
  timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 --train 10 --no-cpu-baseline --no-alt-engine --no-parity > $out/${tag}_train_$w.json 2> $out/${tag}_train_$w.err || { echo "train $w failed"; exit 1; }
done
for w in cfg2 cfg3; do
  RNNWF_ENGINE=f32 timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-alt-engine > $out/${tag}_bench_${w}_engine_f32.json 2>/dev/null || exit 1
done
echo "bench part done"
fi
[ $part = bench ] && exit 0
cd /tmp && export TMPDIR=/tmp
if [ $part = all ] || [ $part = prof ]; then
for w in cfg1 cfg2 cfg3 cfg4 cfg5 cfg2_parity cfg2_l2 cfg2_l3 cfg3_l2; do
  steps=10; [ $w = cfg5 ] && steps=4
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_$w -- python3 $R/bench.py --workload $w --steps $steps --warmup 2 --train 3 --no-cpu-baseline --no-alt-engine --no-parity > $out/${tag}_prof_$w.log 2>&1 || { echo "rocprof $w failed"; exit 1; }
  cp $out/${tag}_prof_$w/*/*kernel_stats.csv $out/${tag}_kernel_stats_$w.csv
  echo "stats $w done"
done
fi
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters of the dominant passes; "pmc" runs this part alone
PMCW=${PMC_WORKLOADS:-cfg2 cfg3 cfg4 cfg5 cfg2_l2 cfg3_l2}
for w in $PMCW; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_${w}_$c -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-alt-engine --no-parity > $out/${tag}_pmc_${w}_$c.log 2>&1 || { echo "pmc $w $c failed"; exit 1; }
  done
done
# SQ counters and held clocks of the dominant kernels (one pass per counter set; no trace domains besides --kernel-trace)
for w in $PMCW; do
  bash $R/tools/sq_counters.sh $w $tag > $out/${tag}_sq_$w.log 2>&1 || echo "sq counters $w failed"
done
cd /tmp
python3 - <<PY
import csv, glob, json, collections
out = "$out"; tag = "$tag"
res = {}
for w in "$PMCW".split():
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
# profiles/pmc_traffic.json: what bench.py replays as roofline.traffic (labelled with this build)
# the dominant pass of a workload: one kernel, or - the layer pipeline of stacked layers - one kernel per layer, summed
PASS = {"cfg2": ("prnn_flip_pp_kernel",), "cfg3": ("crnn_swap_pp_kernel",), "cfg4": ("mdrnn_flip",), "cfg5": ("prnn_flip_riders16",),
        "cfg2_l2": ("prnn_flip_pp_kernel", "prnn_flip_pp_upper_kernel"), "cfg3_l2": ("crnn_swap_pp_kernel", "crnn_swap_pp_upper_kernel")}
try:
    traffic = json.load(open("$R/profiles/pmc_traffic.json"))
except (OSError, ValueError):
    traffic = {}
for w in res:
    ks = [k for k in res[w] if any(sub in k for sub in PASS.get(w, ())) and "FETCH_SIZE" in res[w][k] and "WRITE_SIZE" in res[w][k]]
    if not ks:
        continue
    fetch = sum(res[w][k]["FETCH_SIZE"] for k in ks) * 1024.0              # counter unit KiB
    write = sum(res[w][k]["WRITE_SIZE"] for k in ks) * 1024.0
    traffic[w] = {"kernel": ks[-1] if len(ks) == 1 else " + ".join(sorted(ks)), "fetch_bytes_raw": fetch, "fetch_bytes_x2": 2 * fetch,
                  "write_bytes": write, "hbm_bytes_per_launch": 2 * fetch + write, "build": tag,
                  "source": "gpurun_out/%s_pmc_%s_{FETCH,WRITE}_SIZE (rocprofv3 --pmc, separate passes, tools/refresh_profiles.sh); "
                            "FETCH_SIZE doubled as the gfx950 guide prescribes" % (tag, w)}
# the clock each dominant kernel holds, from the same call's GRBM_GUI_ACTIVE pass (tools/sq_counters.sh)
import re
for w in traffic:
    try:
        txt = open("%s/%s_sq_%s_summary.txt" % (out, tag, w)).read()
    except OSError:
        continue
    if traffic[w].get("build") != tag:
        continue
    m = re.search(re.escape(traffic[w]["kernel"].split(" + ")[-1][:70]) + r".*?held clock[^:]*: ([0-9.]+) GHz", txt, re.S)
    if m:
        traffic[w]["held_clock_ghz"] = float(m.group(1))
        traffic[w]["held_clock_source"] = "profiles/%s_sq_%s_summary.txt (rocprofv3 --pmc GRBM_GUI_ACTIVE / 8 XCDs / kernel duration, tools/sq_counters.sh)" % (tag, w)
json.dump(traffic, open("%s/%s_pmc_traffic.json" % (out, tag), "w"), indent=1)
PY
echo refresh done
