#!/bin/bash
# same-box A/B of an environment switch on the chunk benchmark: tools/micro/ab_env.sh VAR=value [workload]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
kv="$1"; wl="${2:-vla_chunk}"
run() { env $1 python bench.py --workload $wl --steps 40 --warmup 3 --sft-steps 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
run "X_=0"; run "$kv"; run "X_=0"; run "$kv"
