import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gslnls_amd import _lib
L = _lib.lib()
print(json.dumps(bench.multistart_bench(L, _lib, torch, None, 0, 1, 50, 3)["weak_65536_per_gpu"]))
print(json.dumps(bench.multistart_bench(L, _lib, torch, None, 0, 1, 20, 3)["weak_65536_per_gpu"]))
