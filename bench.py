#!/usr/bin/env python3
"""bench.py -- IQ Msample/s through the FCCH+SCH calibration chain on MI355X.

One step = one pass of the whole per-dongle body of gsm_sync_demod.m:107-124 (raw2iq -> channel
filter -> FCCH_coarse_position -> FCCH_fine_correction -> SCH_corr_rate_correction ->
carrier_correct_post_SCH -> total_ppm_calculation) over D synthetic dongle streams per GPU that are
already resident in HBM, ending with the calibration table on the device (and, for N > 1, one RCCL
all-gather of that table across ranks).  Msample/s = complex input samples of all ranks / wall time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams D] [--mode table|stream]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_PEAK_TFLOPS = 78.6       # MI355X fp64: vector peak == matrix (MFMA f64) peak
FLOP_PER_BIN_STEP = 11       # sliding DFT: complex add (2) + complex multiply (6) + |X|^2 (3)
PMC_FILE = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")   # HBM bytes per launch from rocprofv3 --pmc


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=64, help="dongle streams per GPU (weak scaling)")
    ap.add_argument("--frames", type=int, default=102, help="TDMA frames per stream (gsm_sync_demod.m:23)")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic streams per GPU (tiled to --streams)")
    ap.add_argument("--mode", choices=["table", "stream"], default="table",
                    help="table: ppm table + pos_info only (2 B/sample); stream: also write r_correct (18 B/sample)")
    ap.add_argument("--workload", choices=["calib", "scan"], default="calib",
                    help="calib: full FCCH+SCH chain (headline); scan: scanner path of multi_rtl_sdr_gsm_FCCH_scanner.m "
                         "(front end + FCCH_coarse_position + acceptance; use --frames 64 --streams 200)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the CPU-oracle baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="no HIP events at all (otherwise the dominant kernel is bracketed inside the timed region and "
                         "every kernel in a separate untimed pass)")
    ap.add_argument("--no-config2", action="store_true", help="skip the extra two-stream (BASELINE config 2) latency measurement")
    ap.add_argument("--dominant", default="k_front",
                    help="kernel (name prefix) bracketed with HIP events inside the timed region: the front-end kernel "
                         "is the one that moves the path's algorithmic bytes (every other kernel works on a few KB per "
                         "burst and is bound by latency, not by HBM or the ALUs)")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    import torch
    import torch.distributed as dist

    import gsmcal
    from gsmcal import synth

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # one process per GPU; the collective runs whenever a process group exists (GSMCAL_FORCE_DIST=1 lets a
    # single-rank torchrun exercise the RCCL path on a 1-GPU box)
    use_dist = world > 1 or (os.environ.get("GSMCAL_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.workload == "scan":
        return bench_scan(args, rank, world, dev, use_dist)
    D, frames = args.streams, args.frames
    N = frames * synth.FRAME_OV
    fc = 957.4e6                                            # gsm_sync_demod.m:14
    coef = np.ascontiguousarray(synth.fir1(46, 200e3 / synth.FS))   # gsm_sync_demod.m:34
    ts = np.ascontiguousarray(synth.sch_training_sequence())
    cf = np.full(D, fc)

    # ---- synthetic input, resident in HBM before the timed region ----
    nd = max(1, min(args.distinct, D))
    distinct = np.stack([synth.make_stream(dongle=rank * D + i, num_frames=frames)[0] for i in range(nd)])
    raw_t = torch.from_numpy(distinct).to(dev).repeat(((D + nd - 1) // nd, 1))[:D].contiguous()   # tiled on the device
    table_t = torch.zeros((D, gsmcal.TABLE_COLS), dtype=torch.float64, device=dev)
    pos_t = torch.zeros((D, 2, gsmcal.MAX_POS_ROWS), dtype=torch.float64, device=dev)
    rlen_t = torch.zeros((D,), dtype=torch.int64, device=dev)
    r_t = torch.empty((D, N, 2), dtype=torch.float64, device=dev) if args.mode == "stream" else None
    gathered = torch.zeros((world * D, gsmcal.TABLE_COLS), dtype=torch.float64, device=dev) if use_dist else None

    # a dedicated (non-default) stream: the library forks its internal lanes off this stream, and torch's
    # collectives are enqueued on it too, so one synchronize covers the whole step
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = gsmcal.Context(local_rank, stream=stream.cuda_stream)
    lib = ctx.lib
    dp = gsmcal._lib.c_double_p
    coef_p, ts_p, cf_p = coef.ctypes.data_as(dp), ts.ctypes.data_as(dp), cf.ctypes.data_as(dp)

    # The all-gather of step i overlaps the kernels of step i+1: the library writes its table alternately into one of
    # two buffers (it keeps a replay graph for each), RCCL gathers from the one just written on its own stream, and
    # the only wait is before a buffer is written again two steps later (a GPU-side stream wait, the host never
    # blocks).  The timed region ends with both collectives waited for.
    tables = [table_t, torch.zeros_like(table_t)] if use_dist else [table_t]
    gath2 = [gathered, torch.zeros_like(gathered)] if use_dist else None
    works = [None, None]
    nstep = [0]

    def step():
        b = (nstep[0] & 1) if use_dist else 0
        nstep[0] += 1
        if use_dist and works[b] is not None:
            works[b].wait()
        rc = lib.gsmcal_calibrate_batch_dev(ctx.h, C.c_void_p(raw_t.data_ptr()), D, N, coef_p, len(coef), ts_p,
                                            len(ts), cf_p, C.c_void_p(tables[b].data_ptr()),
                                            C.c_void_p(pos_t.data_ptr()),
                                            C.c_void_p(r_t.data_ptr()) if r_t is not None else None,
                                            C.c_void_p(rlen_t.data_ptr()))
        ctx.check(rc, "gsmcal_calibrate_batch_dev")
        if use_dist:
            # one RCCL all-gather of the per-dongle ppm table
            works[b] = dist.all_gather_into_tensor(gath2[b], tables[b], async_op=True)

    def fence():
        for b in range(2):
            if works[b] is not None:
                works[b].wait()
                works[b] = None
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    kernel_events = not args.no_kernel_events
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    # ---- HIP-event passes (rank 0).  Timing events cannot ride in the region above without distorting it: as soon
    # as one dispatch carries start/stop events this runtime switches the queue to profiled dispatch, and the whole
    # step slows by ~13 % (0.34 -> 0.39 ms).  So the same K steps are run again, first with events on the front-end
    # kernel only (the roofline figure), then on every kernel (the breakdown).
    prof_dom, prof = {}, {}
    if kernel_events and rank == 0:
        for filt, store in ((args.dominant, prof_dom), (None, prof)):
            ctx.profile_reset()
            ctx.profile_filter(filt)
            ctx.profile_enable(True)
            for _ in range(args.steps):
                ctx.check(lib.gsmcal_calibrate_batch_dev(ctx.h, C.c_void_p(raw_t.data_ptr()), D, N, coef_p, len(coef), ts_p,
                                                         len(ts), cf_p, C.c_void_p(table_t.data_ptr()),
                                                         C.c_void_p(pos_t.data_ptr()),
                                                         C.c_void_p(r_t.data_ptr()) if r_t is not None else None,
                                                         C.c_void_p(rlen_t.data_ptr())), "gsmcal_calibrate_batch_dev")
            torch.cuda.synchronize(dev)
            store.update(ctx.profile_get())
            ctx.profile_enable(False)
    # BASELINE config 2 (gsm_sync_demod.m on 2 dongle streams): the same call on the first two streams, for the record
    cfg2 = None
    if rank == 0 and D >= 2 and args.mode == "table" and not use_dist and not args.no_config2:
        def two():
            ctx.check(lib.gsmcal_calibrate_batch_dev(ctx.h, C.c_void_p(raw_t.data_ptr()), 2, N, coef_p, len(coef), ts_p, len(ts),
                                                     cf_p, C.c_void_p(table_t.data_ptr()), C.c_void_p(pos_t.data_ptr()), None,
                                                     C.c_void_p(rlen_t.data_ptr())), "gsmcal_calibrate_batch_dev")
        for _ in range(3):
            two()
        torch.cuda.synchronize(dev)
        c0 = time.perf_counter()
        for _ in range(args.steps):
            two()
        torch.cuda.synchronize(dev)
        t2 = (time.perf_counter() - c0) / args.steps
        cfg2 = {"streams": 2, "ms_per_call": round(1e3 * t2, 4), "Msample_per_s": round(2 * N / t2 / 1e6, 1)}
        step()                                  # restore the full batch's outputs and lane bookkeeping
        torch.cuda.synchronize(dev)
    if use_dist:
        fence()
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- results of the last step ----
    table = tables[(nstep[0] - 1) & 1 if use_dist else 0].cpu().numpy()
    det = gsmcal.last_batch_details(min(D, nd), ctx=ctx)
    n_ok = int(np.sum(table[:, 9] == 0))
    total_samples = world * D * N * args.steps
    value = total_samples / elapsed / 1e6

    out = {
        "metric": "IQ Msamples/s through FCCH+SCH calib",
        "value": round(value, 3), "unit": "Msample/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64",
        "data": f"synthetic 8x-oversampled GSM BCCH-carrier uint8 IQ (seed {synth.DEFAULT_SEED}); {nd} distinct "
                f"streams per GPU tiled to {D}",
        "config": {"workload": f"cfg4-style full chain gsm_sync_demod.m:107-124: {D} dongle streams/GPU x {N} IQ samples "
                               f"({frames} frames), fir1(46), FCCH+SCH+total_ppm_calculation",
                   "streams_per_gpu": D, "samples_per_stream": N, "output": args.mode,
                   "bytes_per_sample_algorithmic": 2 if args.mode == "table" else 18,
                   "collective": "all_gather(table) over RCCL" if use_dist else "none",
                   "streams_calibrated_ok": n_ok},
    }
    if cfg2:
        out["baseline_config2_two_streams"] = cfg2

    if rank == 0:
        # ---- roofline (HIP events on the launch streams, inside the timed region) ----
        # The chain reads every raw byte exactly once, in the front-end kernel; all later kernels touch a few KB per
        # burst.  The HBM roofline of the path (north_star: "fraction of HBM roofline") is therefore that kernel's.
        if prof:
            tot = {k: v[0] for k, v in prof.items()}
            dom = max(tot, key=tot.get)
            out["kernels_ms_per_step_untimed_pass"] = {k: round(v[0] / args.steps, 4)
                                                       for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
            front = [k for k in prof_dom if k.startswith("k_front") and prof_dom[k][1]] or \
                    [k for k in prof if k.startswith("k_front") and prof[k][1]]
            traffic = {}
            if os.path.exists(PMC_FILE):
                with open(PMC_FILE) as f:
                    pmc = json.load(f)
                # (the PMC pass ran the default configuration: one launch over all D streams)
                if pmc.get("streams_per_gpu") == D and pmc.get("samples_per_stream") == N and D < 128:
                    traffic = pmc.get("hbm_bytes_per_launch", {})
            if front:
                k = front[0]
                tot_ms, launches = prof_dom[k] if k in prof_dom and prof_dom[k][1] else prof[k]
                avg = tot_ms / launches
                per_launch = D * N * 2.25 * args.steps / launches        # the batch is split over the library's lanes
                ach = per_launch / 1e9 / (avg * 1e-3)
                out["roofline"] = {"kernel": k, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                   "traffic": next((v for kk, v in traffic.items() if kk.startswith("k_front")), None),
                                   "avg_launch_ms": round(avg, 5), "launches_per_step": launches // args.steps,
                                   "algorithmic_bytes_per_launch": int(per_launch),
                                   "timed_with": "HIP events on the kernel's own dispatch (hipExtLaunchKernel start/stop), "
                                                 "second run of the same K steps right after the timed region",
                                   "note": "2 B/sample raw read + 16/64 B/sample decimated complex-double write; the "
                                           "only kernel of the chain that streams from HBM (launches of the "
                                           "library's concurrent lanes overlap other kernels)"}
            out["time_dominant_kernel"] = {"kernel": dom, "ms_per_step": round(tot[dom] / args.steps, 4),
                                           "note": "largest single kernel by time; the per-burst kernels are serial "
                                                   "fp64 decision chains bound by instruction latency (DESIGN.md section 4)"}
        # ---- CPU baseline: the oracle (fp64 NumPy/SciPy restatement) on the host cores, bounded sample ----
        if world == 1 and not args.no_cpu_baseline:
            from oracle import gsmcal_oracle as oracle
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import parity
            done, t_cpu, checked = 0, 0.0, 0
            while t_cpu < args.cpu_seconds and done < 64:
                i = done % nd
                c0 = time.perf_counter()
                orc = oracle.calibrate_stream(distinct[i], coef, ts, fc)
                t_cpu += time.perf_counter() - c0
                done += 1
                if done <= nd:      # checker: the GPU result of this very stream must match the oracle
                    parity.compare_stream(orc, table[i], det, i, _pos_info(pos_t, table, i))
                    checked += 1
            out["cpu_baseline"] = {"value": round(done * N / t_cpu / 1e6, 4), "unit": "Msample/s", "cores": 1,
                                   "kind": "port",
                                   "sample": f"{done} streams x {N} samples through oracle.calibrate_stream "
                                             f"(NumPy/SciPy fp64 restatement, 1 thread) in {t_cpu:.1f} s"}
            out["parity_checked_streams"] = checked
            out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
            # the same restatement on every host core (one worker process per core over independent streams): SURVEY 8d
            # asks for the all-core figure next to the single-thread one
            ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            if ncpu > 1:
                from concurrent.futures import ProcessPoolExecutor
                nrun = min(4 * ncpu, 1024)
                with ProcessPoolExecutor(max_workers=ncpu) as ex:
                    list(ex.map(_oracle_warm, range(2 * ncpu), chunksize=1))      # workers up and imports done, untimed
                    c0 = time.perf_counter()
                    list(ex.map(_oracle_one, [(distinct[i % nd], coef, ts, fc) for i in range(nrun)], chunksize=1))
                    t_all = time.perf_counter() - c0
                out["cpu_baseline_all_cores"] = {"value": round(nrun * N / t_all / 1e6, 4), "unit": "Msample/s", "cores": ncpu,
                                                 "kind": "port",
                                                 "sample": f"{nrun} streams over {ncpu} worker processes in {t_all:.1f} s"}
                out["speedup_vs_cpu_all_cores"] = round(value / out["cpu_baseline_all_cores"]["value"], 1)
        print(json.dumps(out))
    if use_dist:
        if rank == 0 and gathered is not None:
            last = (nstep[0] - 1) & 1
            assert torch.equal(gath2[last][:D], tables[last]), "all-gathered table differs from the local rows"
        dist.destroy_process_group()


def bench_scan(args, rank, world, dev, use_dist):
    """Scanner path (BASELINE configs 3/5): D captures resident in HBM -> snr, num_hit per capture."""
    import torch
    import torch.distributed as dist

    import gsmcal
    from gsmcal import synth
    D, frames = args.streams, args.frames
    N = frames * synth.FRAME_OV
    coef = np.ascontiguousarray(synth.fir1(30, 200e3 / synth.FS))     # multi_rtl_sdr_gsm_FCCH_scanner.m:53
    nd = max(1, min(args.distinct, D))
    distinct = np.stack([synth.make_stream(dongle=1000 + rank, arfcn=i, num_frames=frames, bcch=(i % 4 != 3))[0]
                         for i in range(nd)])
    raw_t = torch.from_numpy(distinct).to(dev).repeat(((D + nd - 1) // nd, 1))[:D].contiguous()   # tiled on the device
    out_t = torch.zeros((D, 2), dtype=torch.float64, device=dev)
    gathered = torch.zeros((world * D, 2), dtype=torch.float64, device=dev) if use_dist else None
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = gsmcal.Context(int(os.environ.get("LOCAL_RANK", "0")), stream=stream.cuda_stream)
    cp = coef.ctypes.data_as(gsmcal._lib.c_double_p)

    def step():
        ctx.check(ctx.lib.gsmcal_fcch_scan_batch_dev(ctx.h, C.c_void_p(raw_t.data_ptr()), D, N, cp, len(coef),
                                                     C.c_void_p(out_t.data_ptr()), None, None, None), "scan")
        if use_dist:
            dist.all_gather_into_tensor(gathered, out_t)

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    prof = {}
    if not args.no_kernel_events and rank == 0:     # untimed pass: per-kernel breakdown (HIP events on the launch stream)
        ctx.profile_reset()
        ctx.profile_filter(None)
        ctx.profile_enable(True)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        prof = ctx.profile_get()
        ctx.profile_enable(False)
    res = out_t.cpu().numpy()
    value = world * D * N * args.steps / elapsed / 1e6
    out = {"metric": "IQ Msamples/s through FCCH scanner path", "value": round(value, 3), "unit": "Msample/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
           "data": f"synthetic GSM uint8 IQ, {nd} distinct captures tiled to {D} (3 of 4 carry a BCCH carrier)",
           "config": {"workload": f"scanner path multi_rtl_sdr_gsm_FCCH_scanner.m:132-135,164-185: {D} captures/GPU x {N} "
                                  f"IQ samples ({frames} frames), fir1(30)", "captures_per_gpu": D,
                      "captures_with_hits": int(np.sum(res[:, 1] > 0)), "bytes_per_sample_algorithmic": 2.25},
           "hbm_GBps_algorithmic": round(value * 1e6 * 2.25 / 1e9, 1)}
    if rank == 0:
        if prof:
            out["kernels_ms_per_step_untimed_pass"] = {k: round(v[0] / args.steps, 4)
                                                       for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
            for k, v in prof.items():
                if k.startswith("k_front") and v[1]:
                    ms = v[0] / v[1]
                    per_launch = D * N * 2.25 * args.steps / v[1]     # the batch may be split over internal lanes
                    ach = per_launch / 1e9 / (ms * 1e-3)
                    out["roofline"] = {"kernel": k, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                                       "avg_launch_ms": round(ms, 5),
                                       "note": "2 B/sample raw read + 16/64 B/sample decimated complex-double write"}
        if not args.no_cpu_baseline and world == 1:
            from oracle import gsmcal_oracle as oracle
            t_cpu, done = 0.0, 0
            while t_cpu < args.cpu_seconds and done < nd:
                c0 = time.perf_counter()
                o = oracle.scan_capture(distinct[done], coef)
                t_cpu += time.perf_counter() - c0
                assert o["num_hit"] == res[done, 1] and abs(o["snr"] - res[done, 0]) < 1e-8, "scan parity"
                done += 1
            out["cpu_baseline"] = {"value": round(done * N / t_cpu / 1e6, 4), "unit": "Msample/s", "cores": 1, "kind": "port",
                                   "sample": f"{done} captures through oracle.scan_capture in {t_cpu:.1f} s"}
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


def _oracle_warm(_):
    from oracle import gsmcal_oracle as oracle  # noqa: F401
    time.sleep(0.05)
    return 0


def _oracle_one(args):
    raw, coef, ts, fc = args
    from oracle import gsmcal_oracle as oracle
    return oracle.calibrate_stream(raw, coef, ts, fc)["total_sampling_ppm"]


def _pos_info(pos_t, table, i):
    k = int(table[i, 7])
    if table[i, 8] == -1.0:
        return -np.ones((k, 2))
    return pos_t[i, :, :k].cpu().numpy().T.copy()


if __name__ == "__main__":
    main()
