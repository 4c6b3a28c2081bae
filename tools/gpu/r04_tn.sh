#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_sft_gpu.py -x -q -k "tn_lds or sumsq_slots" 2>&1 | tail -4
timeout 600 python tools/micro/tn_lab.py 2>&1 | grep -v amdgpu.ids
