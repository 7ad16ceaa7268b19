#!/usr/bin/env python3
"""Which workgroups trail the others?  (GMRM_SWEEP_TRACE dump of the diagnostic build, last launch.)"""
import sys
import numpy as np
path = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 245
d = np.fromfile(path, dtype=np.uint64)
L = 256 * 64 * 8
n = len(d) // L
t = d[(n - 1) * L:n * L].reshape(256, 64, 8).astype(np.float64)[:W] / 100.0
np.set_printoptions(precision=2, suppress=True, linewidth=220)
samp, tops, dots = [], [], []
for r in range(4, 60):
    x = t[:, r, :]
    if (x[:, 3] == 0).any() or (x[:, 5] == 0).any():
        continue
    samp.append(x[:, 5] - x[:, 3])
    top = np.where(x[:, 0] > 0, x[:, 0], x[:, 7])
    tops.append(top - top.min())
    if (x[:, 1] > 0).all():
        dots.append(x[:, 1] - top)
samp, tops, dots = np.array(samp), np.array(tops), np.array(dots)
print('rounds', len(samp))
for name, a in (('sample (totals seen -> barrier)', samp), ('lateness at round top', tops), ('dots (restart rounds)', dots)):
    m = a.mean(axis=0)
    print(f'{name:32s} median {np.median(m):5.2f}  slowest', np.argsort(m)[-5:], np.sort(m)[-5:])
