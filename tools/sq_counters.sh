#!/bin/bash
# usage (GPU box, repo root): tools/sq_counters.sh <workload> <tag>  -> gpurun_out/<tag>_sq_<workload>_summary.txt
# SQ counters of every kernel of the workload, one rocprofv3 --pmc pass per counter set (no trace domain besides --kernel-trace),
# and the clock the dominant kernels hold (GRBM_GUI_ACTIVE / 8 / duration).
w=${1:-cfg3}; tag=${2:-r04_x}
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  t=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/${tag}_sq_$w/$t -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-alt-engine --no-parity > $out/${tag}_sq_${w}_$t.log 2>&1 || echo "sq set failed: $set"
done
python3 - <<PY
import csv, glob, collections
root = "$out/${tag}_sq_$w"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for f in glob.glob(root + "/SQ_WAVES*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0][:70]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
with open("$out/${tag}_sq_${w}_summary.txt", "w") as o:
    for k in sorted(agg, key=lambda k: -sum(dur.get(k, [0]))):
        d = sum(dur[k]) / max(len(dur[k]), 1) if k in dur else 0.0
        o.write("%s   avg %.4f ms over %d launches (counter pass)\n" % (k, d / 1e6, len(dur.get(k, []))))
        for c in sorted(agg[k]):
            o.write("    %-28s per launch %.6g  (launches %d)\n" % (c, agg[k][c] / cnt[k][c], cnt[k][c]))
        if "GRBM_GUI_ACTIVE" in agg[k] and d > 0:
            o.write("    held clock (GRBM_GUI_ACTIVE / 8 engines / duration): %.3f GHz\n" % (agg[k]["GRBM_GUI_ACTIVE"] / cnt[k]["GRBM_GUI_ACTIVE"] / 8 / d))
print(open("$out/${tag}_sq_${w}_summary.txt").read()[:6000])
PY
rm -rf $out/${tag}_sq_$w
