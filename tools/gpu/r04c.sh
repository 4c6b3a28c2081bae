#!/bin/bash
# r04c: re-test the staging fix; timeline of the one-launch attention + o_proj
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attn_oproj or vla_stage or output_ring or uint8 or euler" 2>&1 | tail -8 > gpurun_out/r04c_tests.log
timeout 900 python -m pytest tests/test_models_gpu.py -x -q -k "infer_action or euler or graph" 2>&1 | tail -8 >> gpurun_out/r04c_tests.log
cat gpurun_out/r04c_tests.log
timeout 600 python tools/micro/attn_o_timeline.py > gpurun_out/r04c_ao_timeline.log 2>&1; tail -16 gpurun_out/r04c_ao_timeline.log
NQ=5 timeout 600 python tools/micro/attn_o_timeline.py >> gpurun_out/r04c_ao_timeline.log 2>&1; tail -14 gpurun_out/r04c_ao_timeline.log
