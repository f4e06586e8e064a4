"""Randomised parity sweep (not collected by pytest): N seeded streams through the HIP batch path and the
oracle; prints mismatches.  Usage: python tests/sweep_parity.py [n_streams] [first_dongle] [frames] [batch] [depth]
(batch: streams per gsmcal_calibrate_batch call, default all at once; 64 keeps every call on the fused k_post_chain_r path;
depth > 1: after the oracle comparison the same batches go through gsmcal_calibrate_batch_dev with `depth` calls in flight
(gsmcal_ctx_set_pipeline_depth: the four-launch tail and the moving search's SNR table, the form bench.py's headline runs) and
every table / pos_info / r_len must equal the one-call-at-a-time output bit for bit)"""
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


def _pipelined(gsmcal, raw, batch, depth, coef, ts, single):
    """every batch once more, `depth` calls in flight on one context; returns the number of output sets that differ"""
    import torch
    dev = torch.device("cuda", 0)
    n, n2 = raw.shape
    st = torch.cuda.Stream(device=dev)
    bad = 0
    with torch.cuda.stream(st):
        cx = gsmcal.Context(0, stream=st.cuda_stream)
        cx.set_pipeline_depth(depth)
        outs = []
        for lo in range(0, n, batch):
            hi = min(n, lo + batch)
            d = hi - lo
            r = torch.from_numpy(raw[lo:hi]).to(dev)
            tab = torch.zeros((d, gsmcal.TABLE_COLS), dtype=torch.float64, device=dev)
            pos = torch.zeros((d, 2, gsmcal.MAX_POS_ROWS), dtype=torch.float64, device=dev)
            rlen = torch.zeros((d,), dtype=torch.int64, device=dev)
            st.synchronize()                                  # (the upload is done before the call reads it from its own streams)
            gsmcal.calibrate_batch_dev(r.data_ptr(), d, n2 // 2, coef, ts, FC, tab.data_ptr(), d_pos_info=pos.data_ptr(),
                                       d_r_len=rlen.data_ptr(), ctx=cx)
            outs.append((r, tab, pos, rlen))
        cx.sync()
        for k, (_, tab, pos, rlen) in enumerate(outs):
            ref = single[k]
            t = tab.cpu().numpy()
            same = np.array_equal(t, ref["table"], equal_nan=True)
            p = pos.cpu().numpy()
            same = same and np.array_equal(rlen.cpu().numpy(), ref["r_len"])
            for i in range(t.shape[0]):
                if t[i, 8] != -1.0:                           # (pos_info rows of a calibrated stream: table column 7 counts them)
                    k_rows = int(t[i, 7])
                    same = same and np.array_equal(p[i, :, :k_rows].T, ref["pos_info"][i])
            if not same:
                bad += 1
                print(f"batch {k}: {depth} calls in flight differ from one call at a time")
        cx.close()
    return bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 102
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else n
    depth = int(sys.argv[5]) if len(sys.argv) > 5 else 1
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
    single = []
    for lo in range(0, n, batch):
        hi = min(n, lo + batch)
        out = gsmcal.calibrate_batch(raw[lo:hi], coef, ts, FC)
        single.append(out)
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
    piped = ""
    if depth > 1:
        nbad = _pipelined(gsmcal, raw, batch, depth, coef, ts, single)
        bad += nbad
        piped = f"; the same batches {depth} calls in flight: {nbad} output sets differ from one call at a time"
    print(f"sweep: {n} streams from dongle {first} in batches of {batch}, {n_ok} calibrated, {bad} mismatches{piped}; gen+oracle {t1 - t0:.1f}s gpu {time.time() - t1:.1f}s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
