"""Randomised parity sweep of the scanner path (not collected by pytest): N seeded captures through
gsmcal.fcch_scan_batch and oracle.scan_capture.  Usage: python tests/sweep_scan.py [n_captures] [first_arfcn] [batch] [depth]
(batch + depth > 1: the captures once more in batches of `batch` through gsmcal_fcch_scan_batch_dev with `depth` sweeps in flight --
the form bench.py's 200-capture scanner line runs -- compared with the oracle the same way)"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")


def _gen(args):
    a, kw = args
    import gsmcal
    return gsmcal.synth.make_stream(dongle=900, arfcn=a, num_frames=64, **kw)[0]


def _one(args):
    a, kw = args
    import gsmcal
    from oracle import gsmcal_oracle as o
    raw, _ = gsmcal.synth.make_stream(dongle=900, arfcn=a, num_frames=64, **kw)
    sc = o.scan_capture(raw, gsmcal.synth.fir1(30, 200e3 / gsmcal.synth.FS))
    return sc["snr"], sc["num_hit"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import gsmcal
    rng = np.random.default_rng(first + 1)
    kws = []
    for i in range(n):
        kw = {"bcch": bool(i % 3 != 2)}
        if i % 4 == 1:
            kw["snr_db"] = float(rng.uniform(3, 15))
        kws.append(kw)
    jobs = [(first + i, kws[i]) for i in range(n)]
    with ProcessPoolExecutor(max_workers=min(128, os.cpu_count() or 1)) as ex:      # pool first, GPU afterwards
        raw = np.stack(list(ex.map(_gen, jobs, chunksize=4)))
        res = list(ex.map(_one, jobs, chunksize=4))
    out = gsmcal.fcch_scan_batch(raw, gsmcal.synth.fir1(30, 200e3 / gsmcal.synth.FS))
    bad = 0
    for i, (snr, nh) in enumerate(res):
        if nh != out["num_hit"][i] or abs(snr - out["snr"][i]) > 1e-8:
            bad += 1
            if bad <= 5:
                print("mismatch", i, snr, out["snr"][i], nh, out["num_hit"][i])
    piped = ""
    if len(sys.argv) > 4 and int(sys.argv[4]) > 1:
        batch, depth = int(sys.argv[3]), int(sys.argv[4])
        import torch
        dev = torch.device("cuda", 0)
        st = torch.cuda.Stream(device=dev)
        pbad = 0
        with torch.cuda.stream(st):
            cx = gsmcal.Context(0, stream=st.cuda_stream)
            coef30 = gsmcal.synth.fir1(30, 200e3 / gsmcal.synth.FS)
            got = {}
            for dd in (1, depth):                            # the same batches one call at a time first: same plan, so the same bits
                cx.set_pipeline_depth(dd)
                outs = []
                for lo in range(0, n, batch):
                    r = torch.from_numpy(raw[lo:lo + batch]).to(dev)
                    o2 = torch.zeros((r.shape[0], 2), dtype=torch.float64, device=dev)
                    st.synchronize()
                    gsmcal.fcch_scan_batch_dev(r.data_ptr(), r.shape[0], raw.shape[1] // 2, coef30, o2.data_ptr(), ctx=cx)
                    outs.append((r, o2))
                cx.sync()
                got[dd] = np.concatenate([o2.cpu().numpy() for _, o2 in outs])
            cx.close()
        one, got = got[1], got[depth]
        for i, (snr, nh) in enumerate(res):
            if nh != got[i, 1] or abs(snr - got[i, 0]) > 1e-8 or not np.array_equal(got[i], one[i], equal_nan=True):
                pbad += 1
                if pbad <= 5:
                    print("mismatch (sweeps in flight)", i, snr, got[i, 0], one[i, 0], nh, got[i, 1])
        bad += pbad
        piped = f"; in batches of {batch} with {depth} sweeps in flight: {pbad} mismatches (against the oracle, and bit for bit against the same batches one call at a time)"
    print(f"scan sweep: {n} captures, {int(np.sum(out['num_hit'] > 0))} with hits, {bad} mismatches{piped}")


if __name__ == "__main__":
    main()
