#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1800 python -m pytest tests/test_ops_gpu.py tests/test_sft_gpu.py tests/test_models_gpu.py -x -q 2>&1 | tail -4
timeout 900 python tools/micro/gemm_timeline.py 2>&1 | grep -A 8 "^SFT\|^prefill gate" | grep -v "^--"
timeout 900 python bench.py --workload sft --sft-steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sft', d['ms_per_step'], d['fwd_bwd_ms'])"
timeout 900 python bench.py --workload vla_chunk --steps 30 --warmup 5 --no-cpu-baseline --no-8b 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk', d['ms_per_step'], {k: v for k, v in d['phases'].items() if k.endswith('_ms')})"
