#!/bin/bash
# GPU-box helper: SQ counters of the dominant kernel (action-expert gate/up GEMV, torch-free harness tools/pmc/skinny_pmc), one --pmc pass per counter pair with
# --kernel-trace only.   tools/pmc/collect_skinny_sq.sh <tag>  -> gpurun_out/<tag>_skinny_sq.md
set -e
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
tag="$1"
out="$REPO/gpurun_out/${tag}_skinny_sq.md"
cd /tmp && export TMPDIR=/tmp
echo "# $tag: SQ counters of skinny_kernel<NORM,SWIGLU> (gate/up GEMV, N = 17920, K = 768, M = 4; tools/pmc/skinny_pmc under rocprofv3 --pmc, one pass per pair)" > "$out"
echo >> "$out"; echo "| counters | kernel | calls | per-launch averages |" >> "$out"; echo "|---|---|---|---|" >> "$out"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVES SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA"; do
  d="/tmp/pmc_sk"; rm -rf "$d"
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$d" -- "$REPO/tools/pmc/skinny_pmc" 2 > /tmp/pmc_sk.log 2>&1 || { echo "| $grp | (failed: $(tail -1 /tmp/pmc_sk.log | cut -c1-120)) | | |" >> "$out"; continue; }
  f=$(find "$d" -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$grp" >> "$out" <<'PY'
import csv, sys, collections
f, grp = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r.get('Kernel_Name', '')
    if 'skinny_kernel' not in k: continue
    k = k.split('(')[0].replace('void ', '')
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r.get('Dispatch_Id'))
for k in sorted(acc):
    c = max(len(n[k]), 1)
    print(f"| {grp} | `{k}` | {c} | " + ', '.join(f"{name} {v / c:,.0f}" for name, v in sorted(acc[k].items())) + ' |')
PY
done
cat "$out"
