#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1800 python -m pytest tests/test_sft_gpu.py tests/test_fullsize_gpu.py -x -q -k "sft or recompute or grads or step" 2>&1 | tail -4
for i in 1 2; do
  for mode in side main; do
    if [ $mode = main ]; then export VLASER_SFT_NO_WGRAD_STREAM=1; else unset VLASER_SFT_NO_WGRAD_STREAM; fi
    timeout 900 python bench.py --workload sft --sft-steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], d['fwd_bwd_ms'])"
  done
done
