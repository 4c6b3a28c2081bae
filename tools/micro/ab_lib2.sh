#!/bin/bash
# same-box A/B of two builds of libvlaser_hip.so: tools/micro/ab_lib2.sh <alt .so> <workload: vla_chunk|sft> ; prints ms per step, two interleaved rounds
cd "${GRAFT_REPO_ROOT:-/root/repo}"
alt="$1"; wl="${2:-vla_chunk}"
run() { VLASER_HIP_LIB="$1" python bench.py --workload $wl --steps 40 --warmup 3 --sft-steps 10 --no-cpu-baseline --no-roofline --no-8b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${1:-current}', d['ms_per_step'])"; }
run ""; run "$PWD/$alt"; run ""; run "$PWD/$alt"
