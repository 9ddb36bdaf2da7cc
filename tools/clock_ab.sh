#!/bin/bash
# usage (GPU box): tools/clock_ab.sh <workload> <tag> [ENGINE]  -> held clock (GRBM_GUI_ACTIVE / 8 / duration) and SQ cycle / instruction
# counters of the dominant kernel, one rocprofv3 --pmc pass per counter set (no trace domains besides --kernel-trace)
w=${1:-cfg5}; tag=${2:-x}; eng=${3:-}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/cab_${w}_$tag
[ -n "$eng" ] && export RNNWF_ENGINE=$eng
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/s$i -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-alt-engine --no-parity > $out.s$i.log 2>&1 || { echo "set $i failed"; tail -3 $out.s$i.log; }
done
python3 - <<PY
import csv, glob, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$out/s1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
with open("$out.summary.txt", "w") as o:
    for k in cnt:
        if k in dur and sum(dur[k]) / len(dur[k]) > 1e5:
            d = sum(dur[k]) / len(dur[k])
            line = "%s  [%s %s]  %.4f ms" % (k[:70], "$w", "$tag", d / 1e6)
            for c in sorted(cnt[k]):
                v = sum(cnt[k][c]) / len(cnt[k][c])
                line += "\n     %-28s %.6g" % (c, v)
                if c == "GRBM_GUI_ACTIVE":
                    line += "   -> %.3f GHz" % (v / d / 8)
            print(line); o.write(line + "\n")
PY
