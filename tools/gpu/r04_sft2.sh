#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_sft_gpu.py -x -q -k "sumsq" 2>&1 | tail -5
timeout 600 python tools/micro/sft_timeline.py 2>&1 | tail -48
