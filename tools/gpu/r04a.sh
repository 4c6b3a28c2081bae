#!/bin/bash
# r04a: baseline of the round -- chunk timeline (per-dispatch), bench line of HEAD
set -x
R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
out="$R/gpurun_out/prof_r04a"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$R/bench.py" --workload vla_chunk --steps 12 --warmup 3 --no-cpu-baseline --no-roofline > "$R/gpurun_out/r04a_bench_line.json" 2> "$R/gpurun_out/r04a_bench.err"
python3 "$R/tools/chunk_timeline.py" "$out" 4 > "$R/gpurun_out/r04a_chunk_timeline.md"
rm -rf "$out"
cd "$R"
python3 bench.py --workload vla_chunk --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r04a_chunk_line.json 2> gpurun_out/r04a_chunk.err
tail -c 600 gpurun_out/r04a_chunk_line.json
