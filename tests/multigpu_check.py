#!/usr/bin/env python3
"""Multi-GPU check, for whoever has a node with N >= 2 MI355X (the build box has one GPU, the CPU suite covers the
layout under gloo: tests/test_dist_cpu.py).  One process per GPU; runs BOTH collectives on real calibration tables:

    python tests/multigpu_check.py --gpus N                          (starts its own ranks)
    python -m torch.distributed.run --nproc-per-node N tests/multigpu_check.py --gpus N

  1. units = 7 synthetic dongle streams (the uneven split: shards differ by one row) and 16 (even), sharded
     block-contiguously (gsmcal.dist.shard_range: gsm_sync_demod.m:112 / multi_rtl_sdr_gsm_FCCH_scanner.m:60-65);
  2. every rank calibrates ITS shard on its GPU (gsmcal_calibrate_batch through the C ABI);
  3. the table is gathered (a) by torch.distributed over RCCL (gsmcal.dist.allgather_table and the bench's double-buffered
     TableGatherer), (b) by the native gsmcal_allgather_table, bootstrapped through an id file carrying this launch's
     nonce, with a stale id file of another launch planted at the path first, and (c) by bench.py's current N > 1
     exchange: NativeTableGatherer (in line and on the side stream) on a communicator bootstrapped through the process group,
     and (d) through the decisions bench.py takes before its timed loop (gsmcal.dist.choose_gatherer with a checked trial exchange,
     autotune_placement): every rank must end on the native gatherer and the same placement; (e) the SCANNER workload's
     (snr, num_hit) table of 13 captures (uneven split) through the native collective with two columns, in line and on the side
     stream, with the digest check of every peer's block; (f) pipelined calibration calls with the collective behind each;
  4. every rank compares all gathered tables, row by row and bit for bit, with the table it computes for ALL units on its
     own GPU (unit independence makes that the expected result), and rank 0 checks unit 0 against the CPU oracle.

Exit code 0 and one "multigpu_check OK" line per rank on success."""
from __future__ import annotations

import argparse
import os
import struct
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("OMP_NUM_THREADS", "1")


def self_launch(n):
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))          # a child process: this one never touches the GPU


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--frames", type=int, default=102)
    args = ap.parse_args()
    if "RANK" not in os.environ:
        self_launch(args.gpus)
    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    import numpy as np
    import torch
    import torch.distributed as dist

    ndev = torch.cuda.device_count()
    if ndev < world or local_rank >= ndev:
        raise SystemExit(f"multigpu_check: {world} ranks need {world} devices, this node exposes {ndev}")
    import gsmcal
    from gsmcal import dist as gd
    from gsmcal import synth

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = gsmcal.Context(local_rank, stream=stream.cuda_stream)
    coef = np.ascontiguousarray(synth.fir1(46, 200e3 / synth.FS))
    ts = np.ascontiguousarray(synth.sch_training_sequence())
    fc = 957.4e6
    # this launch's nonce: rank 0 draws it, everybody learns it through the process group
    nonce_t = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == 0:
        nonce_t[0] = int.from_bytes(os.urandom(7), "little") | 1
    dist.broadcast(nonce_t, 0)
    nonce = int(nonce_t.item())
    id_path = os.path.join(tempfile.gettempdir(), "gsmcal_multigpu_check_id")
    if rank == 0:                                    # a record another launch left behind: must be ignored by the readers
        with open(id_path, "wb") as f:
            f.write(struct.pack("<QQ", 0x3144494C41434D47, nonce ^ 0x55) + bytes([0xEE]) * 128)
    dist.barrier()
    comm = gd.NativeComm(ctx, world, rank, id_file=id_path, nonce=nonce)

    for num_units in (7, 16):
        raw = np.stack([synth.make_stream(dongle=4000 + u, num_frames=args.frames)[0] for u in range(num_units)])
        full = gsmcal.calibrate_batch(raw, coef, ts, fc, ctx=ctx)["table"]             # expected: every unit, this GPU
        lo, hi = gd.shard_range(num_units, world, rank)
        sizes = gd.shard_sizes(num_units, world)
        mx = max(sizes)
        if hi > lo:
            local = gsmcal.calibrate_batch(raw[lo:hi], coef, ts, fc, ctx=ctx)["table"]
        else:
            local = np.zeros((0, gsmcal.TABLE_COLS))
        assert np.array_equal(local, full[lo:hi], equal_nan=True), "a shard's rows differ from the full batch (unit independence)"
        local_t = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
        # (a) torch.distributed over RCCL
        got = gd.allgather_table(local_t, num_units).cpu().numpy()
        assert np.array_equal(got, full, equal_nan=True), f"torch all-gather, {num_units} units"
        tg = gd.TableGatherer(sizes, gsmcal.TABLE_COLS, dev)
        for step in range(4):                                                           # bench.py's double-buffered exchange
            b = step & 1
            tg.wait(b)
            tg.post(b, local_t + float(step))
        for b in (0, 1):
            rows = tg.rows(b).cpu().numpy()
            assert np.array_equal(rows, full + float(2 + b), equal_nan=True), f"TableGatherer buffer {b}, {num_units} units"
        # (b) native collective of the C ABI: padded blocks of mx rows per rank
        send = torch.full((mx, gsmcal.TABLE_COLS), float("nan"), dtype=torch.float64, device=dev)
        send[: hi - lo] = local_t
        recv = torch.zeros((world * mx, gsmcal.TABLE_COLS), dtype=torch.float64, device=dev)
        stream.synchronize()
        comm.allgather_table(send.data_ptr(), mx, gsmcal.TABLE_COLS, recv.data_ptr())
        ctx.sync()
        r = recv.cpu().numpy()
        nat = np.concatenate([r[k * mx: k * mx + sizes[k]] for k in range(world)])
        assert np.array_equal(nat, full, equal_nan=True), f"native gsmcal_allgather_table, {num_units} units"
        # (c) bench.py's N > 1 exchange as it runs now: the native communicator bootstrapped through the process group,
        # NativeTableGatherer in line on the chain's stream and on the library's side stream, switching between the two the way the
        # bench's warm-up autotune does
        comm2 = gd.native_comm_from_process_group(ctx, dev)
        try:
            ntg = gd.NativeTableGatherer(ctx, comm2, sizes, gsmcal.TABLE_COLS, dev, mode="inline")
            for mode in ("inline", "async", "inline"):
                torch.cuda.synchronize(dev)
                ntg.work = [None, None]
                ntg.mode = mode
                for step in range(4):
                    b = step & 1
                    ntg.wait(b)
                    ntg.post(b, local_t + float(step))
                for b in (0, 1):
                    rows = ntg.rows(b)
                    torch.cuda.synchronize(dev)
                    if mode == "async":
                        ctx.check(ctx.lib.gsmcal_allgather_sync(ctx.h, b), "gsmcal_allgather_sync")
                    assert np.array_equal(rows.cpu().numpy(), full + float(2 + b), equal_nan=True), f"NativeTableGatherer {mode} buffer {b}, {num_units} units"
                    assert np.array_equal(ntg.own_rows(b).cpu().numpy(), full[lo:hi] + float(2 + b), equal_nan=True)
        finally:
            torch.cuda.synchronize(dev)
            comm2.close()
        # (d) the DECISIONS bench.py takes before its timed loop, on the real communicator: the id through the process group (one
        # collective on every rank), the native set-up + a checked trial exchange under a time-out, the agreement of the ranks, and
        # the placement autotune (max over ranks) -- every rank must end on the native gatherer and on the same placement
        uid = gd.broadcast_unique_id(ctx, dev)
        held = {}

        def make_native():
            with torch.cuda.device(dev), torch.cuda.stream(stream):
                held["comm"] = gd.native_comm_from_process_group(ctx, dev, unique_id=uid)
                return gd.NativeTableGatherer(ctx, held["comm"], sizes, gsmcal.TABLE_COLS, dev, stream=stream)

        def verify(g_):
            with torch.cuda.device(dev), torch.cuda.stream(stream):
                gd.verify_gatherer(g_, gsmcal.TABLE_COLS, dev, lambda: torch.cuda.synchronize(dev))

        tg2, kind, why = gd.choose_gatherer(make_native, lambda: gd.TableGatherer(sizes, gsmcal.TABLE_COLS, dev), dev, verify=verify, timeout_s=120.0)
        assert kind == "native" and why is None, f"choose_gatherer fell back: {why}"
        try:
            import time

            def measure(mode):
                torch.cuda.synchronize(dev)
                dist.barrier()
                t0 = time.perf_counter()
                for step in range(8):
                    b = step & 1
                    tg2.wait(b)
                    tg2.post(b, local_t + float(step))
                for b in (0, 1):
                    tg2.rows(b)
                torch.cuda.synchronize(dev)
                return (time.perf_counter() - t0) / 8

            tune = gd.autotune_placement(tg2, measure, dev)
            chosen = torch.tensor([0 if tune["chosen"] == "inline" else 1], dtype=torch.int32, device=dev)
            lo_c, hi_c = chosen.clone(), chosen.clone()
            dist.all_reduce(lo_c, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi_c, op=dist.ReduceOp.MAX)
            assert int(lo_c.item()) == int(hi_c.item()) and tg2.mode == tune["chosen"], "the ranks disagree on the placement"
            for step in range(4):
                b = step & 1
                tg2.wait(b)
                tg2.post(b, local_t + float(step))
            for b in (0, 1):
                rows = tg2.rows(b)
                torch.cuda.synchronize(dev)
                if tg2.mode == "async":
                    ctx.check(ctx.lib.gsmcal_allgather_sync(ctx.h, b), "gsmcal_allgather_sync")
                assert np.array_equal(rows.cpu().numpy(), full + float(2 + b), equal_nan=True), f"chosen gatherer ({tg2.mode}) buffer {b}, {num_units} units"
        finally:
            torch.cuda.synchronize(dev)
            held["comm"].close()
        if rank == 0 and num_units == 7:
            from oracle import gsmcal_oracle as oracle
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import parity
            res = gsmcal.calibrate_batch(raw[:1], coef, ts, fc, ctx=ctx)
            parity.compare_stream(oracle.calibrate_stream(raw[0], coef, ts, fc), res["table"][0],
                                  gsmcal.last_batch_details(1, ctx=ctx), 0, res["pos_info"][0])
    # (e) the scanner workload's exchange (BASELINE config 5, multi_rtl_sdr_gsm_FCCH_scanner.m:60-65,163-186): 13 captures (the
    # uneven split) through gsmcal_fcch_scan_batch_dev on every rank's shard, the (snr, num_hit) table gathered by the native
    # collective with TWO columns -- in line and on the side stream, double-buffered as bench.py --workload scan posts it -- every
    # rank comparing the gathered table with the table it computes for ALL captures, and every peer's block with that peer's
    # digest (gsmcal.dist.check_gathered_table through an independent gloo group)
    coef30 = np.ascontiguousarray(synth.fir1(30, 200e3 / synth.FS))
    ncap, nscan = 13, 64 * synth.FRAME_OV
    caps = np.stack([synth.make_stream(dongle=4100, arfcn=u, num_frames=64, bcch=(u % 4 != 3))[0] for u in range(ncap)])
    full_sc = gsmcal.fcch_scan_batch(caps, coef30, ctx=ctx)
    full_tab = np.stack([np.asarray(full_sc["snr"], dtype=np.float64), np.asarray(full_sc["num_hit"], dtype=np.float64)], axis=1)
    lo, hi = gd.shard_range(ncap, world, rank)
    sizes = gd.shard_sizes(ncap, world)
    chk = dist.new_group(backend="gloo")
    comm3 = gd.native_comm_from_process_group(ctx, dev)
    try:
        raw_sc = torch.from_numpy(np.ascontiguousarray(caps[lo:hi])).to(dev) if hi > lo else torch.zeros((0, 2 * nscan), dtype=torch.uint8, device=dev)
        outs = [torch.zeros((hi - lo, 2), dtype=torch.float64, device=dev) for _ in range(2)]
        for mode in ("inline", "async"):
            stg = gd.NativeTableGatherer(ctx, comm3, sizes, 2, dev, mode=mode, stream=stream)
            for step in range(4):
                b = step & 1
                stg.wait(b)
                if hi > lo:
                    gsmcal.fcch_scan_batch_dev(raw_sc.data_ptr(), hi - lo, nscan, coef30, outs[b].data_ptr(), ctx=ctx)
                stg.post(b, outs[b])
            for b in (0, 1):
                rows = stg.rows(b)
                torch.cuda.synchronize(dev)
                if mode == "async":
                    ctx.check(ctx.lib.gsmcal_allgather_sync(ctx.h, b), "gsmcal_allgather_sync")
                got = rows.cpu().numpy()
                assert np.array_equal(got, full_tab, equal_nan=True), f"scan table, native gatherer {mode}, buffer {b}"
                gd.check_gathered_table(got, outs[b].cpu().numpy(), sizes, rank, group=chk)
    finally:
        torch.cuda.synchronize(dev)
        comm3.close()
    # (f) pipelined calibration calls (gsmcal_ctx_set_pipeline_depth(2)) with the native collective right behind each call: the
    # all-gather rides on the stream of the call's last stage; after gsmcal_sync every gathered table is the full table
    num_units = 16
    raw = np.stack([synth.make_stream(dongle=4000 + u, num_frames=args.frames)[0] for u in range(num_units)])
    full = gsmcal.calibrate_batch(raw, coef, ts, fc, ctx=ctx)["table"]
    lo, hi = gd.shard_range(num_units, world, rank)
    sizes = gd.shard_sizes(num_units, world)
    if all(sz == sizes[0] for sz in sizes):          # (even shards only: the pad copies of uneven ones are torch's, on the context's stream)
        comm4 = gd.native_comm_from_process_group(ctx, dev)
        try:
            raw_t = torch.from_numpy(np.ascontiguousarray(raw[lo:hi])).to(dev)
            tabs = [torch.zeros((hi - lo, gsmcal.TABLE_COLS), dtype=torch.float64, device=dev) for _ in range(4)]
            ptg = gd.NativeTableGatherer(ctx, comm4, sizes, gsmcal.TABLE_COLS, dev, mode="inline", stream=stream)
            ctx.set_pipeline_depth(2)
            for step in range(4):
                gsmcal.calibrate_batch_dev(raw_t.data_ptr(), hi - lo, raw.shape[1] // 2, coef, ts, fc, tabs[step].data_ptr(), ctx=ctx)
                ptg.post(step & 1, tabs[step])
            ctx.sync()
            ctx.set_pipeline_depth(1)
            for b in (0, 1):
                assert np.array_equal(ptg.rows(b).cpu().numpy(), full, equal_nan=True), f"pipelined calls + in-line collective, buffer {b}"
        finally:
            ctx.set_pipeline_depth(1)
            torch.cuda.synchronize(dev)
            comm4.close()
    comm.close()
    assert not os.path.exists(id_path), "rank 0 should have removed the id file once the communicator was up"
    dist.barrier()
    print(f"multigpu_check OK: rank {rank}/{world}, units 7 (uneven) and 16, torch + native all-gather, scan table of 13 captures, pipelined calls, tables bit-identical")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
