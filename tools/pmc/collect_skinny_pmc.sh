#!/bin/bash
# GPU-box helper: HBM traffic of the dominant kernel (action-expert gate/up GEMV) from hardware counters, as MI355X_MICROARCH.md (HBM / rocprofv3 PMC
# slots) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes with --kernel-trace only, on the torch-free harness (the program itself after `--`).
#   tools/pmc/collect_skinny_pmc.sh <tag>      -> gpurun_out/<tag>_pmc_raw/{fetch,write}_size_counter_collection.csv + <tag>_pmc_dominant_kernel.{md,json}
set -e
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
tag="$1"
raw="$REPO/gpurun_out/${tag}_pmc_raw"
mkdir -p "$raw"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d="/tmp/pmc_$c"; rm -rf "$d"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -- "$REPO/tools/pmc/skinny_pmc" 2 > "$raw/${c}_stdout.log" 2>&1 || { tail -5 "$raw/${c}_stdout.log"; exit 1; }
  f=$(find "$d" -name '*counter_collection.csv' | head -1)
  cp "$f" "$raw/$(echo $c | tr 'A-Z' 'a-z')_counter_collection.csv"
done
"$REPO/tools/pmc/skinny_pmc" 4 > "$raw/unprofiled.log" 2>&1 || true
python3 "$REPO/tools/pmc/summarize_skinny_pmc.py" "$raw" "$tag" "$REPO/gpurun_out"
