#!/usr/bin/env python3
"""Soak of the one-fused-tail-per-device gate: N contexts on N host threads, K fused-size (64-stream) steps each,
beside a context that keeps the CUs busy with scanner batches.  Every table read back must equal the single-context reference.

    python tools/soak_contexts.py [contexts=3] [steps=2000]
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsmcal  # noqa: E402


def main():
    nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    s = gsmcal.synth
    coef, coef30, ts, fc = s.fir1(46, 200e3 / s.FS), s.fir1(30, 200e3 / s.FS), s.sch_training_sequence(), 957.4e6
    distinct = np.stack([s.make_stream(dongle=8200 + d, num_frames=61)[0] for d in range(8)])
    raw = np.tile(distinct, (8, 1))
    ref = gsmcal.calibrate_batch(raw, coef, ts, fc)["table"]
    n = raw.shape[1] // 2
    caps = np.stack([s.make_stream(dongle=8300, arfcn=i, num_frames=26, bcch=i % 3 != 2)[0] for i in range(8)])
    ctxs = [gsmcal.Context(0) for _ in range(nctx + 1)]
    stop, errs, bad, reads = threading.Event(), [], [0] * nctx, [0] * nctx

    def calib(k):
        try:
            cx = ctxs[k]
            d_raw, d_tab, d_pos = cx.alloc(raw.nbytes), cx.alloc(64 * gsmcal.TABLE_COLS * 8), cx.alloc(64 * 2 * gsmcal.MAX_POS_ROWS * 8)
            cx.h2d(d_raw, raw)
            cx.sync()
            t = np.empty((64, gsmcal.TABLE_COLS))
            for step in range(steps):
                gsmcal.calibrate_batch_dev(d_raw, 64, n, coef, ts, fc, d_tab, d_pos, ctx=cx)
                if step % 50 == 49:
                    cx.sync()
                    cx.d2h(t, d_tab)
                    reads[k] += 1
                    if not np.array_equal(t, ref, equal_nan=True):
                        bad[k] += 1
        except Exception as e:  # noqa: BLE001
            errs.append((k, repr(e)))

    def hog():
        big = np.tile(caps, (200, 1))
        while not stop.is_set():
            gsmcal.fcch_scan_batch(big, coef30, ctx=ctxs[nctx])

    th = threading.Thread(target=hog)
    th.start()
    t0 = time.time()
    ws = [threading.Thread(target=calib, args=(k,)) for k in range(nctx)]
    for w in ws:
        w.start()
    for w in ws:
        w.join()
    wall = time.time() - t0
    stop.set()
    th.join()
    stats = [cx.fused_tail_stats() for cx in ctxs[:nctx]]
    print(f"soak: {nctx} contexts x {steps} steps beside a CU hog in {wall:.1f} s; tables read {sum(reads)}, differing {sum(bad)}; "
          f"errors {errs}; (fused launches, gate fall-backs) per context {stats}")
    return 1 if (errs or sum(bad)) else 0


if __name__ == "__main__":
    sys.exit(main())
