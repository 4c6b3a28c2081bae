#!/bin/bash
# r04: fused gradient norm + ViT ahead of the bucket wait -- tests, then interleaved A/B of the SFT bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sft_gpu.py -x -q -k "sumsq or norm_from or tn_lds or step_reduces or recompute" 2>&1 | tail -15
for i in 1 2; do
  for mode in fused nofused; do
    if [ $mode = nofused ]; then export VLASER_SFT_NO_FUSED_NORM=1; else unset VLASER_SFT_NO_FUSED_NORM; fi
    timeout 600 python bench.py --workload sft --sft-steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], d['last_loss'])"
  done
done
