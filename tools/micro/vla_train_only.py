"""The VLA flow-matching training step alone (bench.vla_train_bench), for a kernel trace of SURVEY 8f-1:
rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/micro/vla_train_only.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
print(json.dumps(bench.vla_train_bench(0, steps=10)))
