"""bench.py's llama_layer alone (for rocprofv3 --kernel-trace --stats: per-kernel durations of the decoder layer's launches):
    rocprofv3 --kernel-trace --stats -d gpurun_out/layer_trace -- python3 tools/layer_trace.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from micromix_amd import _lib, mixedgemm
dev = torch.device("cuda:0")
lib = _lib.load()
x = torch.randn((4096, 4096), device=dev, dtype=torch.float32).to(torch.bfloat16)
print(json.dumps(bench.llama_layer(dev, lib, mixedgemm, x, 20)["by_rows"]))
