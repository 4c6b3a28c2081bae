#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1200 python tools/micro/gemm_epilogue_lab.py 2>&1 | grep "^\["
VLASER_HIP_LIB=$PWD/tools/micro/lab_build/libvlaser_epi_lane_swap.so timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_sft_gpu.py -x -q -k "gemm or tn_lds or sumsq or linear" 2>&1 | tail -3
