"""The forward-time harness of bench.py alone: `fwd_harness.py H W [f32|bf16]` (eager and HIP graph)."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
h, w = int(sys.argv[1]), int(sys.argv[2])
dt = sys.argv[3] if len(sys.argv) > 3 else "f32"
print(json.dumps(bench.forward_time_harness(torch.device("cuda:0"), h, w, dt)))
