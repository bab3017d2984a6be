#!/usr/bin/env python3
"""Non-causal forward shapes for an A/B of the tiled kernel (select the build with AULE_LIBRARY_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from split_grid import t
print("lib:", os.environ.get("AULE_LIBRARY_PATH", "(in-tree)"))
bf, fp = torch.bfloat16, torch.float16
t(4, 32, 32, 4096, 4096, 128, bf); t(2, 32, 8, 8192, 8192, 128, bf); t(8, 16, 16, 2048, 2048, 128, bf); t(8, 32, 8, 64, 8192, 128, bf)
t(1, 32, 1, 16384, 16384, 64, fp); t(4, 32, 32, 4096, 4096, 128, fp); t(4, 32, 32, 4096, 4096, 64, bf); t(8, 32, 32, 2048, 2048, 32, bf)
