"""3x3x3 forward / data-gradient launches of C-UNETR's (fs=16) and the UNet's narrow layers: chunks of 2 / 4 k groups (round 4) against the
padded 96-byte plan.  Usage: python scripts/micro/conv_narrow_bench.py     (A/B: MISEG_CONV3_NARROW=0)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench_conv

for s in [(96, 16, 16), (96, 32, 16), (96, 16, 32), (48, 32, 32), (48, 64, 32), (48, 32, 64), (24, 64, 64), (24, 128, 64), (12, 128, 128), (12, 256, 128), (12, 128, 256),
          (96, 48, 48), (48, 96, 96)]:
    bench_conv.run(*s, torch.bfloat16, what=("fwd",))
