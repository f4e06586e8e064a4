#!/usr/bin/env python3
"""Development aid: per-stream completion times of k_post_chain_r from a raw stamp dump (GSMCAL_DEVTIMING_DUMP)."""
import collections
import csv
import sys

import numpy as np

H = int(sys.argv[2]) if len(sys.argv) > 2 else 12
d = collections.defaultdict(dict)
for k, b, i, t in csv.reader(open(sys.argv[1])):
    if int(k) == 2:
        d[int(b)][int(i)] = int(t)
t0 = min(v[0] for v in d.values())
S = (max(d) + H) // H
end = np.array([max(d[s * H + w].get(10, 0) for w in range(H) if s * H + w in d) - t0 for s in range(S)]) / 100.0
print("streams %d: end p10 %.1f p50 %.1f p90 %.1f max %.1f" % ((S,) + tuple(np.percentile(end, [10, 50, 90, 100]))))
for lo in range(0, S, 16):
    print("  streams %2d-%2d: mean end %.1f" % (lo, min(S, lo + 16) - 1, end[lo:lo + 16].mean()))
