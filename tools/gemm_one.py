"""One GEMM shape, one variant, N launches (for rocprofv3 --pmc passes).  usage: tools/gemm_one.py name M variant [n]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import run
from vita_amd import ops
name, M, v = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
shapes = {"qkv": (M, 3360, 1088, 0, 0, {}), "proj": (M, 1088, 1120, 0, 0, dict(out_f32=True, epi=ops.EPI_RESIDUAL)),
          "fc1": (M, 4160, 1088, 0, 0, dict(epi=ops.EPI_GELU)), "fc2": (M, 1088, 4160, 0, 0, dict(out_f32=True, epi=ops.EPI_RESIDUAL)),
          "du": (M, 4160, 1088, 0, 1, dict(epi=ops.EPI_GELU_BWD)), "dln2": (M, 1088, 4160, 0, 1, {}),
          "sq8k": (8192, 8192, 8192, 0, 0, {})}
m, n, k, la, lb, kw = shapes[name]
for _ in range(3):
    run(name, m, n, k, la, lb, [v], **kw)
