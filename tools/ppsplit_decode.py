#!/usr/bin/env python3
"""Decode regime (Sq = 1 and a few query tokens) at sizes where K+V exceeds the Infinity Cache: which forward path
streams K/V fastest?  Same three settings as tools/ppsplit_grid.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from split_grid import t
print("PPSPLIT=%s SPLITKV=%s" % (os.environ.get("AULE_HIP_FWD_PPSPLIT", "on"), os.environ.get("AULE_HIP_FWD_SPLITKV", "on")))
for B in (8, 16, 32, 64):
    for Sk in (8192, 32768):
        if B * Sk <= 64 * 8192 * 2:
            t(B, 32, 8, 1, Sk)
t(8, 32, 8, 4, 32768); t(32, 32, 8, 4, 8192); t(16, 32, 32, 1, 8192); t(4, 64, 8, 1, 65536); t(1, 32, 8, 1, 131072)
t(8, 32, 8, 1, 8192, 64); t(8, 32, 8, 1, 32768, 64, torch.float16)
