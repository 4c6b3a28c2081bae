#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "s16384 or 8b_one_tile" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_vla_train_dp_gpu.py tests/test_sft_gpu.py -x -q 2>&1 | tail -5
