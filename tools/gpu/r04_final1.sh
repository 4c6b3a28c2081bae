#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/r04u_bench_line.json 2> gpurun_out/r04u_bench.err; tail -c 600 gpurun_out/r04u_bench.err | tail -3
cd /tmp && export TMPDIR=/tmp
TAIL_MS=200 bash "$GRAFT_REPO_ROOT/tools/run_profile.sh" r04u_chunk --workload vla_chunk --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-8b > /dev/null
cd "$GRAFT_REPO_ROOT"
python tools/chunk_timeline.py gpurun_out/prof_r04u_chunk 2>/dev/null | tail -30 || true
