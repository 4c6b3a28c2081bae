#!/bin/bash
# GPU-box batch of a round's closing evidence (run through `gpurun -- bash tools/gpu/round_end.sh <tag>`; scripts must live outside gpurun_out/, which is
# not part of the snapshot): the default bench line, rocprofv3 kernel stats of the chunk and of the SFT step, the GPU suite and the smoke.
tag="${1:-rXX}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python bench.py > "gpurun_out/${tag}_bench_line.json" 2> "gpurun_out/${tag}_bench.err"; tail -c 300 "gpurun_out/${tag}_bench.err" | tail -2
cd /tmp && export TMPDIR=/tmp
TAIL_MS=140 bash "$GRAFT_REPO_ROOT/tools/run_profile.sh" "${tag}_sft" --workload sft --sft-steps 10 --no-cpu-baseline --no-roofline --no-8b > /dev/null
TAIL_MS=200 bash "$GRAFT_REPO_ROOT/tools/run_profile.sh" "${tag}_chunk" --workload vla_chunk --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-8b > /dev/null
cd "$GRAFT_REPO_ROOT"
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
