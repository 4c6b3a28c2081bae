#!/bin/bash
# GPU-box helper: SQ counters of the prefill attention kernel (csrc/attn.hip) on the path's shapes (torch-free harness tools/pmc/attn_pmc: A = ViT one tile, C = joint
# prefill S = 384), one --pmc pass per counter group with --kernel-trace only, plus an un-profiled timing.   tools/pmc/collect_attn_pmc.sh <tag>  -> gpurun_out/<tag>_pmc_attn.md
set -e
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
tag="$1"
out="$REPO/gpurun_out/${tag}_pmc_attn.md"
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
echo "# $tag: SQ counters of attn_prefill_kernel (tools/pmc/attn_pmc under rocprofv3 --pmc, one pass per group; per-dispatch means over the steady dispatches)" > "$out"
for shape in A C; do
  echo >> "$out"
  echo "## shape $shape -- un-profiled: $("$REPO/tools/pmc/attn_pmc" $shape 6 2>&1 | tail -1)" >> "$out"
  echo >> "$out"; echo "| counters | kernel | dispatches | per-launch means |" >> "$out"; echo "|---|---|---|---|" >> "$out"
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM"; do
    d="/tmp/pmc_attn"; rm -rf "$d"
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$d" -- "$REPO/tools/pmc/attn_pmc" $shape 3 > /tmp/pmc_attn.log 2>&1 || { echo "| $grp | (failed: $(tail -1 /tmp/pmc_attn.log | cut -c1-120)) | | |" >> "$out"; continue; }
    f=$(find "$d" -name '*counter_collection.csv' | head -1)
    python3 - "$f" "$grp" >> "$out" <<'PY'
import csv, sys, collections
f, grp = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(lambda: collections.defaultdict(dict))
for r in csv.DictReader(open(f)):
    k = r.get('Kernel_Name', '')
    if 'attn_prefill' not in k: continue
    k = k.split('(')[0].replace('void ', '')
    d = rows[k][r['Counter_Name']]
    d[r.get('Dispatch_Id')] = d.get(r.get('Dispatch_Id'), 0.0) + float(r['Counter_Value'])
for k in sorted(rows):
    parts = []
    n = 0
    for name, d in sorted(rows[k].items()):
        v = [d[i] for i in sorted(d, key=lambda x: int(x))]
        v = v[len(v) // 4:]                 # drop the warm-up round
        n = len(v)
        parts.append(f"{name} {sum(v) / len(v):,.0f}")
    print(f"| {grp} | `{k}` | {n} | " + ', '.join(parts) + ' |')
PY
  done
done
cat "$out"
