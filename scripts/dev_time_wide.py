"""developer timing of the wide dense path (p = 32, ten Gaussians + line): fit + per-step time at n = 1e5 and 1e6"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from gslnls_amd import _lib
_lib.lib()
print(json.dumps(bench.wide_dense_bench(_lib), indent=1))
