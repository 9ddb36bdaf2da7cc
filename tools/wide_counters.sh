R=$GRAFT_REPO_ROOT; out=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $out/r04_m_sq_avail.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  t=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/r04_m_ctr/$t -- python3 $R/tools/wide_widths.py $1 > $out/r04_m_ctr_$t.log 2>&1 || echo "set failed: $set"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$out/r04_m_ctr/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in agg:
    if "flip" in k or "swap" in k:
        print(k)
        for c in sorted(agg[k]): print("    %-28s per launch %.6g" % (c, agg[k][c] / cnt[k][c]))
PY
rm -rf $out/r04_m_ctr
