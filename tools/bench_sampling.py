#!/usr/bin/env python3
"""Prior sampling leg of bench.py on its own (codes/s at B = 1 and 8, top prior [32,32], d_model 512, 6+8 layers)."""
import json
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    out = bench._prior_sampling(dev)
    print(json.dumps({k: v for k, v in out.items() if k != "cpu_baseline"}))
