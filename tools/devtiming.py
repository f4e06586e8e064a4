#!/usr/bin/env python3
"""Development aid: per-phase timing inside the kernels of one calibration step.

Builds multi-rtl-sdr-calibration_amd/lib/libgsmcal_dev.so with -DGSMCAL_DEVTIMING (the product library has no
timing code), runs the 64-stream batch a few times, arms the stamp buffer, runs one step and prints the report.

    python tools/devtiming.py [--streams 64] [--workload calib|scan] [--frames 102]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "multi-rtl-sdr-calibration_amd")
DEV = os.path.join(PKG, "lib", "libgsmcal_dev.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--frames", type=int, default=102)
    ap.add_argument("--distinct", type=int, default=64)
    ap.add_argument("--workload", default="calib")
    ap.add_argument("--mode", default="table")
    ap.add_argument("--light", type=int, default=-1,
                    help="kernel id (state.h KID_*): compile in only that kernel's stamps 0, 9, 10 (leaves its register allocation alone)")
    ap.add_argument("--cflags", default="", help="extra compiler flags for the development build (its file name carries a hash of them)")
    ap.add_argument("--mixed", action="store_true", help="streams drawn like bench.py's mixed_batch (low SNR, large ppm, carriers without BCCH)")
    ap.add_argument("--mask", default=None, help="with --light: bit mask of the stamps compiled in (default 0x601 = stamps 0, 9, 10)")
    args = ap.parse_args()
    global DEV
    flags = ["-DGSMCAL_DEVTIMING"]
    if args.light >= 0:
        DEV = DEV.replace("_dev.so", f"_dev_light{args.light}.so")
        flags.append(f"-DGSMCAL_DEVTIMING_LIGHT={args.light}")
        if args.mask:
            DEV = DEV.replace(".so", f"_{args.mask}.so")
            flags.append(f"-DGSMCAL_DEVTIMING_MASK={args.mask}")
    if args.cflags:
        import hashlib
        DEV = DEV.replace(".so", "_" + hashlib.sha1(args.cflags.encode()).hexdigest()[:8] + ".so")
        flags += args.cflags.split()
    src = os.path.join(PKG, "csrc", "gsmcal.hip")
    if not os.path.exists(DEV) or os.path.getmtime(DEV) < max(os.path.getmtime(os.path.join(PKG, "csrc", f)) for f in os.listdir(os.path.join(PKG, "csrc"))):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                        *flags, src, "-o", DEV, "-ldl"], check=True)
    os.environ["GSMCAL_LIB"] = DEV
    os.environ["GSMCAL_GRAPH"] = "0"
    import numpy as np
    import torch
    import gsmcal
    from gsmcal import synth
    lib = gsmcal.load()
    lib.gsmcal_devtiming_begin.argtypes = [C.c_void_p]
    lib.gsmcal_devtiming_report.argtypes = [C.c_void_p]
    dev = torch.device("cuda", 0)
    D, N = args.streams, args.frames * synth.FRAME_OV
    nd = min(args.distinct, D)
    scan = args.workload == "scan"
    coef = np.ascontiguousarray(synth.fir1(30 if scan else 46, 200e3 / synth.FS))
    ts = np.ascontiguousarray(synth.sch_training_sequence())
    cf = np.full(D, 957.4e6)
    if args.mixed:                                        # the unselected distribution of bench.py's mixed_batch sub-result
        import bench
        kws = bench.mixed_kwargs(nd, 5000)
        distinct = np.stack([synth.make_stream(dongle=5000 + i, num_frames=args.frames, **kws[i])[0] for i in range(nd)])
    else:
        distinct = np.stack([synth.make_stream(dongle=i, num_frames=args.frames)[0] for i in range(nd)])
    raw_t = torch.from_numpy(distinct).to(dev).repeat(((D + nd - 1) // nd, 1))[:D].contiguous()
    table_t = torch.zeros((D, gsmcal.TABLE_COLS), dtype=torch.float64, device=dev)
    pos_t = torch.zeros((D, 2, gsmcal.MAX_POS_ROWS), dtype=torch.float64, device=dev)
    rlen_t = torch.zeros((D,), dtype=torch.int64, device=dev)
    out_t = torch.zeros((D, 2), dtype=torch.float64, device=dev)
    r_t = torch.empty((D, N, 2), dtype=torch.float64, device=dev) if args.mode == "stream" else None
    ctx = gsmcal.Context(0)
    dp = gsmcal._lib.c_double_p

    def step():
        if scan:
            ctx.check(lib.gsmcal_fcch_scan_batch_dev(ctx.h, C.c_void_p(raw_t.data_ptr()), D, N, coef.ctypes.data_as(dp), len(coef),
                                                     C.c_void_p(out_t.data_ptr()), None, None, None), "scan")
        else:
            ctx.check(lib.gsmcal_calibrate_batch_dev(ctx.h, C.c_void_p(raw_t.data_ptr()), D, N, coef.ctypes.data_as(dp), len(coef),
                                                     ts.ctypes.data_as(dp), len(ts), cf.ctypes.data_as(dp),
                                                     C.c_void_p(table_t.data_ptr()), C.c_void_p(pos_t.data_ptr()),
                                                     C.c_void_p(r_t.data_ptr()) if r_t is not None else None,
                                                     C.c_void_p(rlen_t.data_ptr())), "calib")
    for _ in range(5):
        step()
    ctx.sync()
    lib.gsmcal_devtiming_begin(ctx.h)
    step()
    ctx.sync()
    lib.gsmcal_devtiming_report(ctx.h)


if __name__ == "__main__":
    main()
