#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1800 python -m pytest tests/test_models_gpu.py tests/test_fullsize_gpu.py tests/test_vla_train_gpu.py tests/test_edge_cases_gpu.py -x -q 2>&1 | tail -4
timeout 900 python bench.py --workload vla_chunk --steps 30 --warmup 5 --no-cpu-baseline --no-8b 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk', d['ms_per_step'], {k: v for k, v in d['phases'].items() if k.endswith('_ms')})"
