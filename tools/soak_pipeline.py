#!/usr/bin/env python3
"""Soak of calls in flight (gsmcal_ctx_set_pipeline_depth): N steps of the 64-stream batch `depth` deep, raw buffers and output sets
taken in turn, every output set compared with the one-call-at-a-time table every `--check` steps; alternating with scanner calls in
flight on the same context every `--mix` steps (calibration and scanner calls share the pipeline's slots).

    python tools/soak_pipeline.py [--steps 20000] [--depth 4] [--check 500] [--mix 0]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--check", type=int, default=500)
    ap.add_argument("--mix", type=int, default=0, help="> 0: a scanner call in flight after every MIX calibration calls")
    args = ap.parse_args()
    import numpy as np
    import torch
    import gsmcal
    s = gsmcal.synth
    dev = torch.device("cuda", 0)
    coef, ts, fc = s.fir1(46, 200e3 / s.FS), s.sch_training_sequence(), 957.4e6
    coef30 = s.fir1(30, 200e3 / s.FS)
    distinct = np.stack([s.make_stream(dongle=9100 + d, num_frames=61)[0] for d in range(8)])
    raw = np.tile(distinct, (8, 1))
    n = raw.shape[1] // 2
    caps = np.stack([s.make_stream(dongle=9200, arfcn=i, num_frames=40, bcch=i % 3 != 2)[0] for i in range(16)])
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        cx = gsmcal.Context(0, stream=st.cuda_stream)
        ref = gsmcal.calibrate_batch(raw, coef, ts, fc, ctx=cx)["table"]
        sref = gsmcal.fcch_scan_batch(caps, coef30, ctx=cx)
        nb = max(4, args.depth)
        raws = [torch.from_numpy(raw).to(dev) for _ in range(nb)]
        caps_t = torch.from_numpy(caps).to(dev)
        tabs = [torch.zeros((64, gsmcal.TABLE_COLS), dtype=torch.float64, device=dev) for _ in range(nb)]
        sout = [torch.zeros((16, 2), dtype=torch.float64, device=dev) for _ in range(nb)]
        cx.set_pipeline_depth(args.depth)
        t0 = time.time()
        bad = 0
        for k in range(args.steps):
            b = k % nb
            gsmcal.calibrate_batch_dev(raws[b].data_ptr(), 64, n, coef, ts, fc, tabs[b].data_ptr(), ctx=cx)
            if args.mix and k % args.mix == args.mix - 1:
                gsmcal.fcch_scan_batch_dev(caps_t.data_ptr(), 16, caps.shape[1] // 2, coef30, sout[b].data_ptr(), ctx=cx)
            if k % args.check == args.check - 1:
                cx.sync()
                for t in tabs:
                    bad += not np.array_equal(t.cpu().numpy(), ref, equal_nan=True)
                if args.mix:
                    for o in sout[: min(nb, (k + 1) // args.mix)]:
                        v = o.cpu().numpy()
                        bad += not (np.array_equal(v[:, 0], sref["snr"], equal_nan=True) and np.array_equal(v[:, 1], sref["num_hit"]))
        cx.sync()
        wall = time.time() - t0
        print(f"soak_pipeline: {args.steps} steps at depth {args.depth}" + (f", a scanner call every {args.mix}" if args.mix else "") +
              f": {bad} wrong output sets, {1e3 * wall / args.steps:.4f} ms per step incl. checks, fused tails {cx.fused_tail_stats()}, re-runs {cx.fused_tail_reruns()}")
        cx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
