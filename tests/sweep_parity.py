"""Randomised parity sweep (not collected by pytest): N seeded streams through the HIP batch path and the
oracle; prints mismatches.  Usage: python tests/sweep_parity.py [n_streams] [first_dongle] [frames] [batch]
(batch: streams per gsmcal_calibrate_batch call, default all at once; 64 keeps every call on the fused k_post_chain_r path)"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "1")
FC = 957.4e6


def _gen_one(args):
    d, frames, kw = args
    import gsmcal
    return gsmcal.synth.make_stream(dongle=d, num_frames=frames, **kw)[0]


def _oracle_one(args):
    d, frames, kw = args
    import gsmcal
    from oracle import gsmcal_oracle as o
    synth = gsmcal.synth
    raw, _ = synth.make_stream(dongle=d, num_frames=frames, **kw)
    coef = synth.fir1(46, 200e3 / synth.FS)
    try:
        return d, o.calibrate_stream(raw, coef, synth.sch_training_sequence(), FC), None
    except o.MatlabIndexError as e:
        return d, None, str(e)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 102
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else n
    import gsmcal
    import parity
    synth = gsmcal.synth
    coef = synth.fir1(46, 200e3 / synth.FS)
    ts = synth.sch_training_sequence()
    rng = np.random.default_rng(first)
    kws = []
    for i in range(n):   # widen the distribution: low SNR, larger ppm, occasional non-BCCH carrier
        kw = {}
        if i % 5 == 1:
            kw["snr_db"] = float(rng.uniform(5, 15))
        if i % 7 == 2:
            kw["sampling_ppm"] = float(rng.uniform(-300, 300))
        if i % 11 == 3:
            kw["bcch"] = False
        if i % 13 == 4:
            kw["carrier_ppm"] = float(rng.uniform(-60, 60))
        kws.append(kw)
    t0 = time.time()
    workers = min(128, os.cpu_count() or 1)
    jobs = [(first + i, frames, kws[i]) for i in range(n)]
    with ProcessPoolExecutor(max_workers=workers) as ex:         # the pool is created before the GPU is touched
        raw = np.stack(list(ex.map(_gen_one, jobs, chunksize=2)))
        res = list(ex.map(_oracle_one, jobs, chunksize=2))
    t1 = time.time()
    bad = 0
    n_ok = 0
    for lo in range(0, n, batch):
        hi = min(n, lo + batch)
        out = gsmcal.calibrate_batch(raw[lo:hi], coef, ts, FC)
        det = gsmcal.last_batch_details(hi - lo)
        for i in range(lo, hi):
            d, orc, err = res[i]
            k = i - lo
            if orc is None:
                if out["table"][k, 9] >= 0:
                    bad += 1
                    print(f"stream {d}: oracle raised '{err}' but gpu status {out['table'][k, 9]}")
                continue
            try:
                parity.compare_stream(orc, out["table"][k], det, k, out["pos_info"][k])
                n_ok += out["table"][k, 9] == 0
            except AssertionError as e:
                bad += 1
                print(f"stream {d} {kws[i]}: MISMATCH {e}")
    print(f"sweep: {n} streams from dongle {first} in batches of {batch}, {n_ok} calibrated, {bad} mismatches; gen+oracle {t1 - t0:.1f}s gpu {time.time() - t1:.1f}s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
