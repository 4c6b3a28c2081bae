#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_sft_gpu.py tests/test_ops_gpu.py -x -q -k "sft or attn_bwd or sumsq" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
TAIL_MS=150 bash "$GRAFT_REPO_ROOT/tools/run_profile.sh" r04p_sft --workload sft --sft-steps 10 --no-cpu-baseline --no-roofline --no-8b > /dev/null
cd "$GRAFT_REPO_ROOT"; tail -1 gpurun_out/r04p_sft_bench_line.json | cut -c1-300
timeout 600 python bench.py --workload sft --sft-steps 20 2>/dev/null | tail -1 | cut -c1-260
