#!/bin/bash
# GPU-box helper: SQ counters of the fused attention backward (csrc/attn_bwd.hip) at the SFT shape, one --pmc pass per counter group with --kernel-trace only
# (MI355X_MICROARCH.md: separate passes), the program itself after `--`.   tools/pmc/collect_attn_bwd_pmc.sh <tag>  -> gpurun_out/<tag>_attn_bwd_pmc.md
set -e
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
tag="$1"
out="$REPO/gpurun_out/${tag}_attn_bwd_pmc.md"
cd /tmp && export TMPDIR=/tmp
echo "# $tag: SQ counters of attn_bwd_dq_kernel / attn_bwd_dkv_kernel (S = 560, 12/2 heads, hd 128; tools/micro/attn_bwd_lab.py under rocprofv3 --pmc, one pass per group)" > "$out"
echo >> "$out"; echo "| counters | kernel | calls | per-launch averages |" >> "$out"; echo "|---|---|---|---|" >> "$out"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD"; do
  d="/tmp/pmc_ab"; rm -rf "$d"
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$d" -- python3 "$REPO/tools/micro/attn_bwd_lab.py" > /tmp/pmc_ab.log 2>&1 || { echo "| $grp | (failed: $(tail -1 /tmp/pmc_ab.log | cut -c1-120)) | | |" >> "$out"; continue; }
  f=$(find "$d" -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$grp" >> "$out" <<'PY'
import csv, sys, collections
f, grp = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r.get('Kernel_Name', '')
    if 'attn_bwd' not in k: continue
    k = k.split('(')[0].replace('void ', '')
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r.get('Dispatch_Id'))
for k in sorted(acc):
    c = max(len(n[k]), 1)
    print(f"| {grp} | `{k}` | {c} | " + ', '.join(f"{name} {v / c:,.0f}" for name, v in sorted(acc[k].items())) + ' |')
PY
done
cat "$out"
