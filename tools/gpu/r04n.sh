#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
run() { VLASER_LAB_NOSEAM=$1 python bench.py --workload vla_chunk --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases']; print('noseam=$1', d['ms_per_step'], 'vit', p['vit_projector_scatter_ms'])"; }
{ run 0; run 1; run 0; run 1; } > gpurun_out/r04n_noseam_lab.log 2>&1; cat gpurun_out/r04n_noseam_lab.log
