#!/bin/bash
# same-box comparison of several values of one environment switch on the chunk benchmark: tools/micro/ab_envs.sh VAR v1 v2 ... (two rounds)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
var="$1"; shift
run() { env $1 python bench.py --workload vla_chunk --steps 40 --warmup 3 --no-cpu-baseline --no-roofline --no-8b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for r in 1 2; do run "X_=default"; for v in "$@"; do run "$var=$v"; done; done
