#!/bin/bash
# r04k: InternViT attention tail (1025th token): tests + same-box A/B of the chunk
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attn" 2>&1 | tail -8 > gpurun_out/r04k_tests.log; cat gpurun_out/r04k_tests.log
run() { VLASER_ATTN_NO_TAIL=$1 python bench.py --workload vla_chunk --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases']; print('no_tail=$1', d['ms_per_step'], 'vit', p['vit_projector_scatter_ms'], 'prefill', p['joint_prefill_ms'], 'euler', p['euler_ms'])"; }
{ run 0; run 1; run 0; run 1; } > gpurun_out/r04k_ab.log 2>&1
cat gpurun_out/r04k_ab.log
