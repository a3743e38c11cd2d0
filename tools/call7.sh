cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAVES"; do
  d=/tmp/pmce_$(echo $pass | cut -c4-14)
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/time_emit.py 1024 sparse 1 > $d.log 2>&1
  python - <<PY
import csv, glob, collections
for f in glob.glob("$d/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        if "emit" not in k: continue
        print(k)
        for name, v in sorted(c.items()): print(f"   {name:24s} {sum(v)/len(v):.5g} (n={len(v)})")
PY
done
