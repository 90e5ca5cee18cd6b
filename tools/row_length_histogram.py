#!/usr/bin/env python3
"""Chunks of 4 off-diagonal entries per row on every level of the series-B hierarchy (what the block sweeps' lanes-per-row rule sees).
usage: row_length_histogram.py <grid> <rank-grid>   (GPU)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools import series_b as sb

n, p = int(sys.argv[1]), int(sys.argv[2])
os.environ["HDA_BLOCKS"] = str(p ** 3)
import hypredrive_amd as hh
from hypredrive_amd import hypredrv as hd
h = hd.Hypredrv(sb.YAML_CPU_DEFAULTS)
s = sb.lap7_rank_blocks(n, p)
h.set_matrix_csr(0, n ** 3 - 1, s[0], s[1], s[2])
h.set_rhs_array(0, n ** 3 - 1, s[3])
h.finish_system()
h.create_and_setup()
A, amg = hh._lib.borrow(h)
for l in range(amg.num_levels - 1):
    M = amg.level_matrix(l, 0).to_scipy()
    ch = (np.diff(M.indptr) - 1 + 3) // 4
    q = np.percentile(ch, [50, 90, 99, 99.9, 100])
    hist = np.bincount(np.minimum(ch, 40))
    cum = np.cumsum(hist) / ch.size
    print(f"level {l}: rows {M.shape[0]} mean chunks {ch.mean():.2f} pct50/90/99/99.9/max {q}  share of rows with <= 4/8/12/16 chunks: "
          f"{cum[min(4, len(cum) - 1)]:.4f} {cum[min(8, len(cum) - 1)]:.4f} {cum[min(12, len(cum) - 1)]:.4f} {cum[min(16, len(cum) - 1)]:.4f}", flush=True)
