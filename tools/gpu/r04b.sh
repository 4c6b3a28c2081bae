#!/bin/bash
# r04b: staging launch + output ring + attention/o_proj rewrite: tests, then same-box A/B of fuse_ao, then the bench line
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attn_oproj or vla_stage or output_ring or uint8 or vla_glue or euler" 2>&1 | tail -15 > gpurun_out/r04b_tests.log
timeout 600 python -m pytest tests/test_models_gpu.py -x -q -k "infer_action or euler or graph" 2>&1 | tail -15 >> gpurun_out/r04b_tests.log
cat gpurun_out/r04b_tests.log
AB_ROUNDS=5 timeout 900 python tools/micro/ab_chunk.py qkv16,glue1 qkv16,glue1,fuse_ao > gpurun_out/r04b_ab.log 2>&1
tail -12 gpurun_out/r04b_ab.log
VLASER_EULER=qkv16,glue1,fuse_ao timeout 600 python bench.py --workload vla_chunk --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r04b_chunk_line.json 2> gpurun_out/r04b_chunk.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04b_chunk_line.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['phases'])
PY
