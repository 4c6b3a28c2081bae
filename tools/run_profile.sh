#!/bin/bash
# GPU-box helper: rocprofv3 kernel trace + stats of a bench.py invocation, summary written to gpurun_out/<tag>_kernel_stats.md
#   [TAIL_MS=300] tools/run_profile.sh <tag> <bench.py args...>      (TAIL_MS: also summarise the last T ms = the timed steps)
set -e
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
tag="$1"; shift
out="$REPO/gpurun_out/prof_$tag"
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$REPO/bench.py" "$@" > "$REPO/gpurun_out/${tag}_bench_line.json" 2> "$REPO/gpurun_out/${tag}_bench.err" || { tail -20 "$REPO/gpurun_out/${tag}_bench.err"; exit 1; }
python3 "$REPO/tools/prof_summary.py" "$out" "$tag: python3 bench.py $* (at::native rows = one-off synthetic weight generation)" --timeline ${TAIL_MS:+--tail-ms $TAIL_MS} > "$REPO/gpurun_out/${tag}_kernel_stats.md"
rm -rf "$out"
head -40 "$REPO/gpurun_out/${tag}_kernel_stats.md"; tail -3 "$REPO/gpurun_out/${tag}_kernel_stats.md"
