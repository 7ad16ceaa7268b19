#!/usr/bin/env python3
"""Summarise a GMRM_SWEEP_TRACE dump (diagnostic build): per-round stamps of every workgroup.
usage: trace_rounds.py trace.bin [W] [round ...]   (last launch in the file is analysed)"""
import sys
import numpy as np
path = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 245
rounds = [int(x) for x in sys.argv[3:]] or [10, 11, 12]
d = np.fromfile(path, dtype=np.uint64)
L = 256 * 64 * 8
n = len(d) // L
t = d[(n - 1) * L:n * L].reshape(256, 64, 8).astype(np.float64)[:W] / 100.0
np.set_printoptions(precision=2, suppress=True, linewidth=200)
names = ((0, 'top (restart)'), (7, 'top (promoted)'), (1, 'dots done'), (2, 'reduce done'), (3, 'totals seen'), (5, 'after sample'), (6, 'after update'))
for r in rounds:
    x = t[:, r, :]
    tops = np.where(x[:, 0] > 0, x[:, 0], x[:, 7])
    base = tops.min()
    print('round', r)
    for k, name in names:
        v = x[:, k]
        if (v == 0).all():
            continue
        v = v - base
        print(f'  {name:15s} min {v.min():6.2f} med {np.median(v):6.2f} max {v.max():6.2f}')
    if (x[:, 1] > 0).all():
        print('  own dots      p0/50/100', np.percentile(x[:, 1] - tops, [0, 50, 100]))
        print('  own reduce wait        ', np.percentile(x[:, 2] - x[:, 1], [0, 50, 100]))
    print('  own totals wait        ', np.percentile(x[:, 3] - x[:, 2], [0, 50, 100]))
    print('  own sample             ', np.percentile(x[:, 5] - x[:, 3], [0, 50, 100]))
