#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1800 python -m pytest tests/test_ops_gpu.py tests/test_sft_gpu.py -x -q -k "gemm or swiglu or sft or linear" 2>&1 | tail -3
for i in 1 2; do timeout 900 python bench.py --workload sft --sft-steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sft', d['ms_per_step'], d['fwd_bwd_ms'])"; done
