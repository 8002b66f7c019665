"""known-byte-count read with the KLT staging pattern (one 16-byte row load per lane), for PMC calibration"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov2slam_amd import frontend as fe
ctx = fe.Context(0)
stride, nrows = 816, 1 << 21                      # 816 B = the padded level-0 row of a 752-wide image; 1.7 GB >> 256 MiB of L3
buf = ctx.alloc(stride * nrows)
out = ctx.alloc(4 * nrows)
for _ in range(3):
    assert ctx.lib.ov2_dbg_rowload16(ctx.h, buf, stride, nrows, out) == 0
ctx.synchronize()
print("bytes_per_launch", stride * nrows)
