#!/usr/bin/env python3
"""Many-row non-causal shapes (encoder / diffusion-like: few heads, S of a few thousand) for the `rows <= 4 Sk` bound
of route 5.  Run with the in-tree library and with build/libaule_norows.so (bound lifted) in one gpurun call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from split_grid import t
print("lib:", os.environ.get("AULE_LIBRARY_PATH", "(in-tree)"))
t(1, 16, 16, 1024, 1024); t(1, 8, 8, 2048, 2048); t(1, 8, 2, 512, 512 * 2); t(4, 8, 8, 1024, 4096, 64); t(1, 8, 8, 4096, 1024, 32)
t(1, 8, 8, 4096, 4096, 64); t(2, 10, 10, 4096, 4096, 64); t(1, 20, 20, 1024, 1024, 64); t(1, 12, 12, 1024, 2048, 64); t(1, 16, 16, 2048, 2048)
t(1, 4, 4, 4096, 4096); t(2, 8, 8, 1024, 1024, 64); t(1, 24, 24, 1536, 1536, 64); t(1, 8, 8, 1024, 1024, 128, torch.float16)
