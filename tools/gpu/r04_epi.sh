#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_sft_gpu.py -x -q -k "gemm or tn_lds or sumsq or linear" 2>&1 | tail -5
timeout 900 python tools/micro/gemm_epilogue_lab.py 2>&1 | grep "^\[" 
