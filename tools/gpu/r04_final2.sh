#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/r04w_bench_line.json 2> gpurun_out/r04w_bench.err; tail -c 300 gpurun_out/r04w_bench.err | tail -2
cd /tmp && export TMPDIR=/tmp
TAIL_MS=140 bash "$GRAFT_REPO_ROOT/tools/run_profile.sh" r04w_sft --workload sft --sft-steps 10 --no-cpu-baseline --no-roofline --no-8b > /dev/null
TAIL_MS=200 bash "$GRAFT_REPO_ROOT/tools/run_profile.sh" r04w_chunk --workload vla_chunk --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-8b > /dev/null
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04w_bench_line.json').read().strip().splitlines()[-1])
print('chunk', d['ms_per_step'], d['value'], {k:v for k,v in d['phases'].items() if k.endswith('_ms')})
print('roofline', d['roofline']['frac'], d['roofline']['traffic_source'].get('measured_in_run'), 'chunk_roofline', d['chunk_roofline']['frac'])
print('sft', d['sft']['ms_per_step'], d['sft']['fwd_bwd_ms'], d['sft']['value'], d['sft']['mfma_frac'])
print('qa', d['qa']['batch1']); print('8b', d['qa_8b']['prefill_ms'], d['qa_8b']['decode_ms_per_step'])
PY
