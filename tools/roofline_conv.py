#!/usr/bin/env python3
"""Launch the roofline kernel of bench.py (3x3 convolution at the VGG conv1_2 shape) exactly as bench.py does:
target for `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (HBM traffic per launch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

print(bench.conv_roofline(torch.device('cuda:0'), iters=20))       # the same 3 + 20 launches as bench.py's roofline leg
