#!/bin/bash
# r04j: is the one-graph chunk slower than its three phases? VLASER_GRAPH_SPLIT=1 vs 3, interleaved
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
run() { VLASER_GRAPH_SPLIT=$1 python bench.py --workload vla_chunk --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases']; print('split=$1', d['ms_per_step'], 'graph', p['chunk_graph_ms'], 'phases', p['sum_ms'], 'overhead', p['call_overhead_ms'])"; }
{ run 1; run 3; run 1; run 3; } > gpurun_out/r04j_graph_split.log 2>&1
cat gpurun_out/r04j_graph_split.log
