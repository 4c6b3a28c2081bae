#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
timeout 600 python tools/micro/attn_o_timeline.py > gpurun_out/r04h_ao_timeline.log 2>&1; tail -14 gpurun_out/r04h_ao_timeline.log
