#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attn" 2>&1 | tail -6 > gpurun_out/r04m_tests.log; cat gpurun_out/r04m_tests.log
{
echo "== tail path on"; python tools/micro/vit_attn_probe.py
echo "== VLASER_ATTN_NO_TAIL=1"; VLASER_ATTN_NO_TAIL=1 python tools/micro/vit_attn_probe.py
} > gpurun_out/r04m_vit_attn.log 2>&1
grep -v amdgpu.ids gpurun_out/r04m_vit_attn.log
