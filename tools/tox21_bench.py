"""Only the real-Tox21 legs of bench.py (fit / predict at batch 64 and 100): python tools/tox21_bench.py [epochs]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    print(json.dumps(bench.tox21_real(dev, int(sys.argv[1]) if len(sys.argv) > 1 else 3), indent=1))
