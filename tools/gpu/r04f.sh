#!/bin/bash
# r04f: kernel arguments in one scalar round trip (skinny / attn_skinny / attn_oproj): same-box A/B of the chunk, HIP_FORCE_DEV_KERNARG probe, timeline
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
run() { VLASER_HIP_LIB="$1" python bench.py --workload vla_chunk --steps 40 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['ms_per_step'])"; }
{
run "" touch; run "$PWD/tools/micro/lab_build/libvlaser_notouch.so" notouch; run "" touch; run "$PWD/tools/micro/lab_build/libvlaser_notouch.so" notouch
HIP_FORCE_DEV_KERNARG=1 run "" touch_devkernarg1; HIP_FORCE_DEV_KERNARG=0 run "" touch_devkernarg0
VLASER_EULER=qkv16,glue1,fuse_ao run "" touch_fuse_ao; VLASER_EULER=qkv16,glue1,fuse_ao run "$PWD/tools/micro/lab_build/libvlaser_notouch.so" notouch_fuse_ao
} > gpurun_out/r04f_ab.log 2>&1
cat gpurun_out/r04f_ab.log
timeout 600 python tools/micro/attn_o_timeline.py > gpurun_out/r04f_ao_timeline.log 2>&1; tail -13 gpurun_out/r04f_ao_timeline.log
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "skinny or attn" 2>&1 | tail -5
