#!/usr/bin/env python3
"""Launch the roofline kernel of bench.py (implicit-GEMM conv at the VGG conv1_2 shape) a few times:
target for `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (HBM traffic per launch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

print(bench.conv_roofline(torch.device('cuda:0'), iters=5))
