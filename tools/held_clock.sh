#!/bin/bash
# usage (GPU box): tools/held_clock.sh <workload> -> GRBM_GUI_ACTIVE (shader-engine clock cycles the GPU was busy) per launch and
# kernel durations of the same run: clock held by the dominant kernel = cycles / duration
w=${1:-cfg4}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/clock_$w
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-alt-engine > $out.log 2>&1 || { echo failed; tail -5 $out.log; exit 1; }
python3 - <<PY
import csv, glob, collections
cnt = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for f in glob.glob("$out/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in cnt:
    if k in dur and sum(dur[k]) / len(dur[k]) > 2e5:
        c = sum(cnt[k]) / len(cnt[k]); d = sum(dur[k]) / len(dur[k])
        print("%-60s GRBM_GUI_ACTIVE %.4g per launch, %.4f ms -> %.3f GHz (if the counter sums %d engines: %.3f GHz)" % (k[:60], c, d / 1e6, c / d, 8, c / d / 8))
PY
