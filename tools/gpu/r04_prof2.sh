#!/bin/bash
cd /tmp && export TMPDIR=/tmp
TAIL_MS=140 bash "$GRAFT_REPO_ROOT/tools/run_profile.sh" r04t_sft --workload sft --sft-steps 10 --no-cpu-baseline --no-roofline --no-8b > /dev/null
cd "$GRAFT_REPO_ROOT"; tail -1 gpurun_out/r04t_sft_bench_line.json | cut -c1-200
