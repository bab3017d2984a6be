#!/usr/bin/env python3
"""Timing grid for the packed-rows + KV-split mode of the tiled kernel.  Run in one gpurun call under
(default) / AULE_HIP_FWD_PPSPLIT=0 (previous behaviour) / AULE_HIP_FWD_SPLITKV=0 (new path instead of the
wave-per-chunk kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from split_grid import t
print("PPSPLIT=%s SPLITKV=%s" % (os.environ.get("AULE_HIP_FWD_PPSPLIT", "on"), os.environ.get("AULE_HIP_FWD_SPLITKV", "on")))
for B, Hq, Hkv in ((1, 32, 8), (8, 32, 8), (8, 32, 32), (1, 32, 1), (2, 32, 8)):
    for Sq in (1, 16, 32, 64):
        for Sk in (2048, 8192):
            t(B, Hq, Hkv, Sq, Sk)
for B, Hq, Hkv, Sq, Sk in ((1, 8, 8, 300, 8192), (1, 8, 8, 1024, 32768), (1, 32, 8, 256, 8192), (1, 32, 8, 1024, 8192),
                           (2, 16, 16, 512, 16384), (4, 32, 8, 128, 4096), (1, 32, 32, 2048, 2048), (4, 32, 32, 4096, 4096)):
    t(B, Hq, Hkv, Sq, Sk)
t(1, 32, 1, 64, 16384, 64, torch.float16); t(1, 32, 1, 1, 16384, 64, torch.float16)
