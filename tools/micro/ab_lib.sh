#!/bin/bash
# same-box A/B of two builds of libvlaser_hip.so on the chunk benchmark: tools/micro/ab_lib.sh <alt .so> [bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
alt="$1"; shift
run() { VLASER_HIP_LIB="$1" python bench.py --workload vla_chunk --steps 40 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${1:-current}', d['ms_per_step'])"; }
run ""; run "$PWD/$alt"; run ""; run "$PWD/$alt"
