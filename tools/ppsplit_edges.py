#!/usr/bin/env python3
"""The edges of the packed-rows + KV-split rule that the main grids do not cover: short K/V and prefill-like
non-causal shapes with few heads.  Run under default and AULE_HIP_FWD_PPSPLIT=0 in one gpurun call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from split_grid import t
print("PPSPLIT=%s" % os.environ.get("AULE_HIP_FWD_PPSPLIT", "on"))
for Sk in (128, 256, 512, 1024):
    t(1, 32, 8, 1, Sk); t(8, 32, 8, 1, Sk); t(1, 32, 8, 64, Sk); t(32, 32, 8, 1, Sk)
t(1, 4, 4, 64, 256); t(1, 32, 32, 128, 1024); t(1, 8, 8, 2048, 2048); t(1, 16, 16, 1024, 1024); t(2, 8, 8, 4096, 4096)
t(1, 8, 2, 512, 512); t(1, 2, 2, 8192, 8192); t(1, 32, 8, 2048, 2048); t(4, 8, 8, 1024, 4096, 64); t(1, 8, 8, 4096, 1024, 32)
