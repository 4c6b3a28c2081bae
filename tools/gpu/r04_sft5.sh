#!/bin/bash
# SFT: AdamW grid size / stream priority sweep (one box, interleaved)
cd "$GRAFT_REPO_ROOT"
run() { timeout 600 python bench.py --workload sft --sft-steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['fwd_bwd_ms'])"; }
for rep in 1 2; do
  run default
  for b in 96 128 192 384 512; do VLASER_ADAMW_BLOCKS=$b run blocks$b; done
  VLASER_SFT_OPT_PRIORITY=-1 run opt_high_prio
done
