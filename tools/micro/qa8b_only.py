"""The Vlaser-8B x 13-tile side number alone (bench.qa8b_bench), for a kernel trace of BASELINE configs[3]:
rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/micro/qa8b_only.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
print(json.dumps(bench.qa8b_bench(0)))
