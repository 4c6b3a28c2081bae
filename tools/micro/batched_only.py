"""The batch-4 chunk side number alone (bench.batched_chunks), for a kernel trace:
rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/micro/batched_only.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from vlaser_amd import config as C
torch.set_grad_enabled(False)
print(json.dumps(bench.batched_chunks(C.VLAConfig(base=C.vlaser_2b()), 'cuda:0', 20)))
