#!/bin/bash
# SQ and L2 counters of the kernels that count the wide look-alike dictionaries (Unidic-like, o200k-like), one dictionary and one engine
# at a time: the micro-step walker `overlap_count_kernel` (what Auto picks there) and PFX beside it — the evidence behind "bound by the
# scattered-request rate" (DESIGN.md 4.7).   bash tools/pmc_wide.sh [mib]
R=${GRAFT_REPO_ROOT:-$(pwd)}; M=${1:-256}
cd /tmp && export TMPDIR=/tmp
export DAAC_PMC_FILTER="overlap_count_kernel,pfx_kernel" DAAC_PMC_MEAN=1
for name in unidic_like o200k_like; do
 for eng in DArray Pfx; do
  echo "===== $name, engine $eng, $M MiB, .count() (per-launch means; bytes per launch = $M MiB)"
  for pass in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
   d=/tmp/pmcw; rm -rf $d
   rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/ab_wide.py $M $name $eng count > $d.log 2>&1
   grep "GB/s" $d.log | head -1
   python $R/tools/pmc_summary.py $d | grep -v "^/tmp"
  done
 done
done
