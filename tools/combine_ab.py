#!/usr/bin/env python3
"""Route-5 shapes for an A/B of the combine kernel (AULE_HIP_FWD_COMBINE=wg = the workgroup-per-row one)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from split_grid import t
print("COMBINE=%s" % os.environ.get("AULE_HIP_FWD_COMBINE", "rows (default)"))
t(1, 32, 1, 1, 16384, 64, torch.float16); t(1, 32, 1, 64, 16384, 64, torch.float16)
t(8, 32, 8, 64, 8192); t(8, 32, 8, 32, 8192); t(8, 32, 8, 16, 8192); t(32, 32, 8, 1, 8192); t(1, 8, 8, 1024, 32768)
t(1, 32, 8, 1, 8192); t(2, 32, 8, 64, 8192); t(1, 32, 8, 1024, 8192); t(4, 32, 8, 128, 4096); t(2, 6, 3, 17, 2049, 32, torch.float16)
