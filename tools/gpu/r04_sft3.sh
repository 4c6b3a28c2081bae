#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_sft_gpu.py -x -q -k "sumsq" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
bash "$GRAFT_REPO_ROOT/tools/run_profile.sh" r04o_sft --workload sft --sft-steps 10 --no-cpu-baseline --no-roofline --no-8b
