#!/bin/bash
# r04g: attention + o_proj v4 (fixed first-pass list, wave-owned K tiles, batched P V reads): tests, timeline, A/B
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attn_oproj" 2>&1 | tail -8 > gpurun_out/r04g_tests.log
cat gpurun_out/r04g_tests.log
timeout 600 python tools/micro/attn_o_timeline.py > gpurun_out/r04g_ao_timeline.log 2>&1; tail -13 gpurun_out/r04g_ao_timeline.log
AB_ROUNDS=3 timeout 900 python tools/micro/ab_chunk.py qkv16,glue1 qkv16,glue1,fuse_ao > gpurun_out/r04g_ab.log 2>&1
tail -12 gpurun_out/r04g_ab.log
