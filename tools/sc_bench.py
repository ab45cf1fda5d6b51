"""A/B timing of the Schmidl-Cox kernel (and the full RX chain) on config-3 frames; prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import api
from tools import bench_cfg3

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
print(json.dumps(bench_cfg3.run(api, torch, frames, 5, 0)))
