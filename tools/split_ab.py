#!/usr/bin/env python3
"""The shapes the old split-KV rule mis-dispatched, plus ones that must not move (select the build with
AULE_LIBRARY_PATH; run both in one gpurun call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from split_grid import t
print("lib:", os.environ.get("AULE_LIBRARY_PATH", "(in-tree)"))
t(8, 32, 8, 64, 8192); t(8, 32, 8, 64, 2048); t(8, 32, 8, 32, 2048); t(8, 32, 32, 8, 2048); t(8, 32, 32, 64, 8192)
t(8, 32, 8, 16, 8192); t(8, 32, 8, 1, 8192); t(1, 32, 8, 64, 8192)          # unchanged: still split-KV
t(1, 32, 1, 64, 16384, 64, torch.float16); t(1, 32, 1, 1, 16384, 64, torch.float16)   # C5c, C5b
