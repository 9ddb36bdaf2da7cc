#!/bin/bash
# usage (on the GPU box): [WORKLOAD=cfg4] tools/pmc_base.sh <outdir>  - SQ counters of the base/flip kernels at config 2, one pass per counter set
out=${1:-gpurun_out/pmc_base}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$out/$tag -- python3 $R/bench.py --workload ${WORKLOAD:-cfg2} --steps 2 --warmup 1 --no-cpu-baseline --no-alt-engine > $R/$out/$tag.log 2>&1 || echo "set failed: $set"
done
python3 - <<PY
import csv, glob, collections, os
root = "$R/$out"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
with open(root + "/summary.txt", "w") as o:
    for k in agg:
        o.write(k + "\n")
        for c in sorted(agg[k]):
            o.write("    %-28s per launch %.4g  (launches %d)\n" % (c, agg[k][c] / cnt[k][c], cnt[k][c]))
print(open(root + "/summary.txt").read())
PY
