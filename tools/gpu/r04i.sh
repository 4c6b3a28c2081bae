#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attn_oproj" 2>&1 | tail -8 > gpurun_out/r04i_tests.log; cat gpurun_out/r04i_tests.log
timeout 600 python tools/micro/attn_o_timeline.py > gpurun_out/r04i_ao_timeline.log 2>&1; tail -16 gpurun_out/r04i_ao_timeline.log
AB_ROUNDS=3 AB_PHASES=0 timeout 900 python tools/micro/ab_chunk.py qkv16,glue1 qkv16,glue1,fuse_ao > gpurun_out/r04i_ab.log 2>&1; tail -6 gpurun_out/r04i_ab.log
