#!/usr/bin/env python3
"""What does the N > 1 step cost on ONE GPU?  (0.255 ms with the collective against 0.185 without.)
One process, one rank, RCCL communicator of size 1; the 64-stream calibration step timed in variants:

  A  table stored straight into pinned host memory, no collective                  (the N = 1 headline)
  B  table in device memory, no collective, no copy                               (what the zero-copy store is worth)
  C  B + torch.distributed all_gather_into_tensor, double-buffered (bench.py r3)  (the N > 1 path of round 3)
  D  B + gsmcal_allgather_table on the context's stream (native RCCL, in line)
  E  B + gsmcal_allgather_table_async (native RCCL on the library's side stream, off the chain's critical path)

For each: ms per step (K steps between synchronisations) and the host time spent ENQUEUEING a step (if that reaches the
step time, the variant is launch-bound, not GPU-bound).   python tools/dist_cost.py [--streams 64] [--steps 200]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--distinct", type=int, default=16)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--variants", default="ABCDE")
    args = ap.parse_args()
    import bench
    import gsmcal
    from gsmcal import dist as gd
    from gsmcal import synth
    raws = bench.gen_streams([(100000 + i, 102, {}) for i in range(args.distinct)])
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = gsmcal.Context(0, stream=stream.cuda_stream)
    coef = np.ascontiguousarray(synth.fir1(46, 200e3 / synth.FS))
    ts = np.ascontiguousarray(synth.sch_training_sequence())
    D, N = args.streams, 102 * synth.FRAME_OV
    raw_t = torch.from_numpy(np.stack(raws)).to(dev).repeat(((D + len(raws) - 1) // len(raws), 1))[:D].contiguous()
    cal_host = bench.Calib(torch, gsmcal, dev, ctx, raw_t, N, "table", coef, ts, 957.4e6, zero_copy=True)
    cal_dev = bench.Calib(torch, gsmcal, dev, ctx, raw_t, N, "table", coef, ts, 957.4e6, zero_copy=False)
    tg = gd.TableGatherer([D], gsmcal.TABLE_COLS, dev)
    comm = gd.NativeComm(ctx, 1, 0, unique_id=gd.NativeComm.unique_id(ctx))
    gathered = [torch.zeros((D, gsmcal.TABLE_COLS), dtype=torch.float64, device=dev) for _ in range(2)]
    lib = ctx.lib
    have_async = hasattr(lib, "gsmcal_allgather_table_async")
    k = [0]

    def step_a():
        cal_host.launch(0)

    def step_b():
        cal_dev.launch(k[0] & 1); k[0] += 1

    def step_c():
        b = k[0] & 1; k[0] += 1
        tg.wait(b)
        cal_dev.launch(b)
        tg.post(b, cal_dev.table_t[b])

    def step_d():
        b = k[0] & 1; k[0] += 1
        cal_dev.launch(b)
        comm.allgather_table(cal_dev.table_t[b].data_ptr(), D, gsmcal.TABLE_COLS, gathered[b].data_ptr())

    def step_e():
        b = k[0] & 1; k[0] += 1
        lib.gsmcal_allgather_wait(ctx.h, b)                 # the gather posted two steps ago read this table buffer
        cal_dev.launch(b)
        ctx.check(lib.gsmcal_allgather_table_async(ctx.h, comm.h, C.c_void_p(cal_dev.table_t[b].data_ptr()), D, gsmcal.TABLE_COLS,
                                                   C.c_void_p(gathered[b].data_ptr()), b), "allgather_async")

    def fence():
        for b in range(2):
            tg.wait(b)
        if have_async:
            for b in range(2):
                lib.gsmcal_allgather_wait(ctx.h, b)
        torch.cuda.synchronize(dev)

    out = {}
    variants = {"A": step_a, "B": step_b, "C": step_c, "D": step_d}
    if have_async:
        variants["E"] = step_e
    for name, fn in variants.items():
        if name not in args.variants:
            continue
        for _ in range(10):
            fn()
        fence()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            t1 = time.perf_counter()
            fence()
            t2 = time.perf_counter()
            r = (1e3 * (t2 - t0) / args.steps, 1e3 * (t1 - t0) / args.steps)
            if best is None or r[0] < best[0]:
                best = r
        out[name] = {"ms_per_step": round(best[0], 4), "host_enqueue_ms_per_step": round(best[1], 4)}
        print(name, out[name], file=sys.stderr)
    ref = cal_host.table(0).numpy()
    for b in range(2):
        assert np.array_equal(cal_dev.table_t[b].cpu().numpy(), ref, equal_nan=True)
    print(json.dumps({"streams": D, "steps": args.steps, "variants": out}))
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
