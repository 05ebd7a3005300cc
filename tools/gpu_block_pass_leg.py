"""bench.py's block_pass leg alone (GPU box): python tools/gpu_block_pass_leg.py > gpurun_out/block_pass_leg.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fabric-mod_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np   # noqa: E402
import bench   # noqa: E402
import fabgpu   # noqa: E402
import coracle   # noqa: E402

print(json.dumps(bench.block_pass_leg(np, fabgpu, coracle), indent=1))
