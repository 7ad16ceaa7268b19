#!/usr/bin/env python3
"""Who is last?  From a GMRM_SWEEP_TRACE dump (diagnostic build): for every traced round the workgroup that reaches
each stamp last, how far behind the median it is, and the round's length.   usage: trace_who.py trace.bin [W]"""
import sys
import numpy as np
path = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 245
d = np.fromfile(path, dtype=np.uint64)
L = 256 * 64 * 8
n = len(d) // L
t = d[(n - 1) * L:n * L].reshape(256, 64, 8).astype(np.float64)[:W] / 100.0     # us
names = {0: 'top(restart)', 7: 'top(promoted)', 1: 'dots done', 2: 'reduce done', 3: 'totals seen', 5: 'after sample', 6: 'after update'}
prev_end = None
for r in range(64):
    x = t[:, r, :]
    if (x[:, 5] == 0).all():
        continue
    kind = 'U-restart' if (x[:, 0] > 0).any() else 'E-promoted'
    base = np.where(x[:, 0] > 0, x[:, 0], x[:, 7]).min()
    line = [f'round {r:2d} {kind:10s}']
    for k in (1, 2, 3, 5, 6):
        v = x[:, k]
        if (v == 0).all():
            continue
        v = v - base
        line.append(f'{names[k]}: med {np.median(v):5.2f} max {v.max():5.2f} (wg {int(v.argmax()):3d}, min wg {int(v.argmin()):3d} {v.min():5.2f})')
    end = x[:, 6].max()
    if prev_end is not None:
        line.append(f'len {end - prev_end:5.2f}')
    prev_end = end
    print(' | '.join(line))

# summary: how far behind the median the last workgroup is, and who it is
import collections
for k in (1, 2, 3, 5, 6):
    lag, who = [], collections.Counter()
    for r in range(64):
        x = t[:, r, :]
        v = x[:, k]
        if (v == 0).all() or (x[:, 5] == 0).all():
            continue
        lag.append(v.max() - np.median(v))
        who[int(v.argmax())] += 1
    if lag:
        print(f'{names[k]:13s}: last - median = {np.mean(lag):5.2f} us on average; most often last: {who.most_common(6)}')

# per-workgroup mean lag (own stamp - median of the round) at 'dots done' and 'totals seen'
for k in (1, 3):
    acc = np.zeros(W); cnt = 0
    for r in range(64):
        v = t[:, r, k]
        if (v == 0).all() or (t[:, r, 5] == 0).all():
            continue
        acc += v - np.median(v); cnt += 1
    if cnt:
        m = acc / cnt
        order = np.argsort(-m)
        print(f'{names[k]:13s}: mean lag per workgroup, slowest 40:', ' '.join(f'{int(w)}:{m[w]:.2f}' for w in order[:40]))
        print(f'{"":13s}  fastest 10:', ' '.join(f'{int(w)}:{m[w]:.2f}' for w in order[-10:]))
