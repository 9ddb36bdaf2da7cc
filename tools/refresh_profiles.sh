#!/bin/bash
# usage (on the GPU box, from the repo root):  tools/refresh_profiles.sh <tag>     e.g.  r01_e
# Writes gpurun_out/<tag>_*: bench lines of the five BASELINE configs, rocprofv3 kernel stats (cfg2, cfg3, cfg4),
# FETCH_SIZE / WRITE_SIZE passes of cfg2, cfg4 and cfg5 (separate --pmc runs, as the gfx950 guide prescribes).
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
cd /tmp && export TMPDIR=/tmp
for w in cfg1 cfg2 cfg3 cfg4 cfg5 cfg2_parity cfg2_l2 cfg2_l3 cfg3_l2; do
  steps=10; [ $w = cfg5 ] && steps=4
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_$w -- python3 $R/bench.py --workload $w --steps $steps --warmup 2 --train 3 --no-cpu-baseline --no-alt-engine --no-parity > $out/${tag}_prof_$w.log 2>&1 || { echo "rocprof $w failed"; exit 1; }
  cp $out/${tag}_prof_$w/*/*kernel_stats.csv $out/${tag}_kernel_stats_$w.csv
  echo "stats $w done"
done
for w in cfg2 cfg4 cfg5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_${w}_$c -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-alt-engine --no-parity > $out/${tag}_pmc_${w}_$c.log 2>&1 || { echo "pmc $w $c failed"; exit 1; }
  done
done
# SQ counters of the dominant kernels at config 2 (one pass per counter set; no trace domains besides --kernel-trace)
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES"; do
  t=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/${tag}_sq_cfg2/$t -- python3 $R/bench.py --workload cfg2 --steps 2 --warmup 1 --no-cpu-baseline --no-alt-engine --no-parity > $out/${tag}_sq_cfg2_$t.log 2>&1 || echo "sq set failed: $set"
done
python3 - <<PY
import csv, glob, collections
root = "$out/${tag}_sq_cfg2"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
with open("$out/${tag}_sq_cfg2_summary.txt", "w") as o:
    for k in agg:
        o.write(k + "\n")
        for c in sorted(agg[k]):
            o.write("    %-28s per launch %.6g  (launches %d)\n" % (c, agg[k][c] / cnt[k][c], cnt[k][c]))
PY
python3 - <<PY
import csv, glob, json, collections
out = "$out"; tag = "$tag"
res = {}
for w in ("cfg2", "cfg4", "cfg5"):
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
def pick(w, sub):
    for k, d in res.get(w, {}).items():
        if sub in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            return k, d
    return None, None
traffic = {}
for w, sub in (("cfg2", "flip"), ("cfg4", "mdrnn_flip"), ("cfg5", "flip")):
    k, d = pick(w, sub)
    if k:
        fetch, write = d["FETCH_SIZE"] * 1024.0, d["WRITE_SIZE"] * 1024.0       # counter unit KiB
        traffic[w] = {"kernel": k, "fetch_bytes_raw": fetch, "fetch_bytes_x2": 2 * fetch, "write_bytes": write,
                      "hbm_bytes_per_launch": 2 * fetch + write, "build": tag,
                      "source": "gpurun_out/%s_pmc_%s_{FETCH,WRITE}_SIZE (rocprofv3 --pmc, separate passes, tools/refresh_profiles.sh); "
                                "FETCH_SIZE doubled as the gfx950 guide prescribes" % (tag, w)}
json.dump(traffic, open("%s/%s_pmc_traffic.json" % (out, tag), "w"), indent=1)
PY
echo refresh done
