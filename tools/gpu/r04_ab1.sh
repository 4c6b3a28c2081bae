#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1800 python -m pytest tests/test_ops_gpu.py tests/test_sft_gpu.py tests/test_edge_cases_gpu.py -x -q -k "attn_bwd or sft or recompute or grads" 2>&1 | tail -4
for i in 1 2; do
  for mode in one two; do
    if [ $mode = two ]; then export VLASER_ATTN_BWD_TWO_LAUNCHES=1; else unset VLASER_ATTN_BWD_TWO_LAUNCHES; fi
    timeout 900 python bench.py --workload sft --sft-steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], d['fwd_bwd_ms'])"
  done
done
