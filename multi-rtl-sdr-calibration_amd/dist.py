"""Multi-GPU layer: shard independent (dongle, ARFCN) units across ranks and collect the per-unit
result table with ONE all-gather (RCCL over xGMI on the GPU box; gloo in the CPU tests).

The reference has no distributed layer: dongles are matrix columns processed by a serial loop
(gsm_sync_demod.m:112) and the scanner splits the ARFCN list across dongles
(multi_rtl_sdr_gsm_FCCH_scanner.m:60-65).  Units never read each other's data, so the only exchange
step is gathering the table every rank needs for the inter-dongle comparison
(gsm_sync_demod.m:151-158)."""
from __future__ import annotations

import numpy as np


def shard_range(num_units: int, world: int, rank: int):
    """Block-contiguous partition: rank g owns units [g*U//G, (g+1)*U//G)."""
    return (rank * num_units) // world, ((rank + 1) * num_units) // world


def shard_sizes(num_units: int, world: int):
    return [shard_range(num_units, world, r)[1] - shard_range(num_units, world, r)[0] for r in range(world)]


def allgather_table(local_table, num_units: int, group=None):
    """local_table: torch tensor (n_local, C) on this rank's device -> (num_units, C) on every rank,
    rows in global unit order.  One collective; uneven shards are padded to the largest shard."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = shard_sizes(num_units, world)
    mx = max(sizes)
    c = local_table.shape[1]
    if local_table.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("local table does not match this rank's shard")
    send = local_table
    if send.shape[0] != mx:
        pad = torch.full((mx - send.shape[0], c), float("nan"), dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], dim=0)
    send = send.contiguous()
    out = torch.empty((world * mx, c), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(out, send, group=group)
    if all(s == mx for s in sizes):
        return out
    return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


class TableGatherer:
    """The exchange step of bench.py: the all-gather of step i runs while step i+1 computes.  Two send / receive buffer
    pairs alternate; `post(b, local)` pads this rank's rows to the largest shard and starts the (asynchronous)
    collective on pair b, `wait(b)` completes it and `rows(b)` returns the table in global unit order.  Works on any
    torch.distributed backend (RCCL on the GPU box, gloo in the CPU tests)."""

    def __init__(self, sizes, cols, device, dtype=None, group=None, pairs=2):
        """pairs: send / receive buffer pairs taken in turn (2: the gather of step i under step i+1; more when more steps are in flight)"""
        import torch
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.pairs = int(pairs)
        self.sizes = list(sizes)
        self.world = len(self.sizes)
        self.rank = dist.get_rank(group)
        self.mx = max(self.sizes)
        dtype = dtype or torch.float64
        self.send = [torch.full((self.mx, cols), float("nan"), dtype=dtype, device=device) for _ in range(self.pairs)]
        self.recv = [torch.zeros((self.world * self.mx, cols), dtype=dtype, device=device) for _ in range(self.pairs)]
        self.work = [None] * self.pairs

    def post(self, b, local):
        if local.shape[0] != self.sizes[self.rank]:
            raise ValueError("local table does not match this rank's shard")
        self.wait(b)
        src = local
        if local.shape[0] != self.mx:                      # uneven shards: pad to the largest (NaN rows are dropped by rows())
            self.send[b][: local.shape[0]].copy_(local)
            src = self.send[b]
        self.work[b] = self.dist.all_gather_into_tensor(self.recv[b], src.contiguous(), group=self.group, async_op=True)

    def wait(self, b):
        if self.work[b] is not None:
            self.work[b].wait()
            self.work[b] = None

    def rows(self, b):
        import torch
        self.wait(b)
        if all(s == self.mx for s in self.sizes):
            return self.recv[b]
        return torch.cat([self.recv[b][r * self.mx: r * self.mx + self.sizes[r]] for r in range(self.world)], dim=0)

    def own_rows(self, b):
        """this rank's rows inside the gathered buffer (for the self-check)"""
        off = self.rank * self.mx
        return self.recv[b][off: off + self.sizes[self.rank]]

    def reset(self, mode=None):
        for b in range(self.pairs):
            self.wait(b)


class NativeComm:
    """The same collective without PyTorch: gsmcal_allgather_table of the C ABI (RCCL, enqueued on the context's
    stream).  Bootstrap by a file every rank can see (rank 0 writes the 128-byte id together with the
    launch's `nonce`; the others accept only a record carrying it) or by an id passed in."""

    def __init__(self, ctx, world, rank, id_file=None, unique_id=None, nonce=None):
        import ctypes as C
        self.ctx, self.world, self.rank = ctx, int(world), int(rank)
        h = C.c_void_p()
        if unique_id is not None:
            buf = C.create_string_buffer(bytes(unique_id), 128)
            rc = ctx.lib.gsmcal_comm_init_rank(ctx.h, buf, self.world, self.rank, C.byref(h))
        elif nonce is not None:       # run-specific id file: readers ignore anything an earlier launch left at the path
            rc = ctx.lib.gsmcal_comm_init_file_nonce(ctx.h, str(id_file).encode(), int(nonce) & (2**64 - 1), self.world, self.rank, C.byref(h))
        else:
            rc = ctx.lib.gsmcal_comm_init_file(ctx.h, str(id_file).encode(), self.world, self.rank, C.byref(h))
        ctx.check(rc, "gsmcal_comm_init")
        self.h = h

    @staticmethod
    def unique_id(ctx):
        import ctypes as C
        buf = C.create_string_buffer(128)
        ctx.check(ctx.lib.gsmcal_comm_get_unique_id(buf), "gsmcal_comm_get_unique_id")
        return buf.raw

    def allgather_table(self, d_local, rows_per_rank, cols, d_all):
        """device pointers (ints); enqueues only"""
        import ctypes as C
        self.ctx.check(self.ctx.lib.gsmcal_allgather_table(self.ctx.h, self.h, C.c_void_p(d_local), int(rows_per_rank),
                                                           int(cols), C.c_void_p(d_all)), "gsmcal_allgather_table")

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.gsmcal_comm_destroy(self.h)
            self.h = None


class NativeTableGatherer:
    """TableGatherer's interface on the C ABI's own collective (gsmcal_allgather_table: native RCCL enqueued by the library,
    no torch.distributed call per step).  mode "inline": the all-gather sits on the context's stream right behind the kernels
    that fill the table (one C call, no event traffic between streams -- measured +2 us per 64-stream step on one rank,
    against +15 us for torch's collective or for any cross-stream event choreography); mode "async":
    gsmcal_allgather_table_async -- the collective on the library's side stream behind an event, the gather of step i under
    the kernels of step i+1, at the price of that event (+14 us per step, tools/dist_cost.py)."""

    def __init__(self, ctx, comm, sizes, cols, device, mode="inline", dtype=None, stream=None, pairs=2):
        """`stream`: the torch stream that wraps the context's HIP stream (default: torch's current stream at construction).
        The collective is enqueued by the library on the CONTEXT's stream, the pad copies and the re-assembly of uneven shards
        by torch: both must be the same stream or the copy races the all-gather -- checked here when the context
        knows its stream, and every torch operation of this class runs inside `torch.cuda.stream(stream)`."""
        import torch
        self.ctx, self.comm, self.mode = ctx, comm, mode
        self.pairs = int(pairs)
        if not 1 <= self.pairs <= 4:
            raise ValueError("NativeTableGatherer: 1..4 buffer pairs (gsmcal_allgather_table_async has four slots)")
        self.sizes = list(sizes)
        self.world, self.rank = len(self.sizes), comm.rank
        self.mx = max(self.sizes)
        self.cols = int(cols)
        dtype = dtype or torch.float64
        if dtype != torch.float64:
            raise TypeError("gsmcal_allgather_table moves doubles (ncclDouble): the table must be float64")
        self.stream = stream if stream is not None else (torch.cuda.current_stream(device) if torch.device(device).type == "cuda" else None)
        ctx_stream = getattr(ctx, "stream_handle", None)
        # a context that owns its (non-blocking) stream: no torch stream wraps it, so stream order cannot put torch's pad copies /
        # re-assembly and the library's collective in sequence -- this class then synchronises on the host around them
        # (slower, correct); contexts created on a torch stream (bench.py) pay nothing
        self._foreign = self.stream is not None and ctx_stream is None
        if self.stream is not None and ctx_stream is not None and int(self.stream.cuda_stream) != int(ctx_stream):
            raise ValueError("NativeTableGatherer: the torch stream is not the context's stream -- pad copies would race the collective")
        with self._on_stream():
            self.send = [torch.full((self.mx, cols), float("nan"), dtype=dtype, device=device) for _ in range(self.pairs)]
            self.recv = [torch.zeros((self.world * self.mx, cols), dtype=dtype, device=device) for _ in range(self.pairs)]
        self.work = [None] * self.pairs

    def _on_stream(self):
        import contextlib
        import torch
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def post(self, b, local):
        import ctypes as C
        import torch
        if local.shape[0] != self.sizes[self.rank]:
            raise ValueError("local table does not match this rank's shard")
        if local.dtype != torch.float64:
            raise TypeError("gsmcal_allgather_table moves doubles (ncclDouble): the table must be float64")
        self.wait(b)
        src = local
        if local.shape[0] != self.mx or not local.is_contiguous():     # uneven shards: pad to the largest (NaN rows are dropped by rows())
            if self._foreign:
                self.ctx.sync()                                       # (`local` was written on the context's stream)
            with self._on_stream():
                self.send[b][: local.shape[0]].copy_(local)
                if self._foreign:
                    self.stream.synchronize()                         # (... and the collective below must see the copy)
            src = self.send[b]
        if self.mode == "async":
            self.ctx.check(self.ctx.lib.gsmcal_allgather_table_async(self.ctx.h, self.comm.h, C.c_void_p(src.data_ptr()), self.mx, self.cols,
                                                                     C.c_void_p(self.recv[b].data_ptr()), b), "gsmcal_allgather_table_async")
        else:
            self.comm.allgather_table(src.data_ptr(), self.mx, self.cols, self.recv[b].data_ptr())
        self.work[b] = True

    def wait(self, b):
        """the context's stream is ordered behind buffer pair b's collective (a GPU-side wait; stream order alone when inline)"""
        if self.work[b] is not None and self.mode == "async":
            self.ctx.check(self.ctx.lib.gsmcal_allgather_wait(self.ctx.h, b), "gsmcal_allgather_wait")
        self.work[b] = None

    def rows(self, b):
        import torch
        if self.work[b] is not None and self.mode == "async":
            self.ctx.check(self.ctx.lib.gsmcal_allgather_wait(self.ctx.h, b), "gsmcal_allgather_wait")
        if self._foreign:
            self.ctx.sync()                                           # (readers of the returned tensor run on torch's stream)
        if all(s == self.mx for s in self.sizes):
            return self.recv[b]
        with self._on_stream():
            return torch.cat([self.recv[b][r * self.mx: r * self.mx + self.sizes[r]] for r in range(self.world)], dim=0)

    def own_rows(self, b):
        off = self.rank * self.mx
        return self.recv[b][off: off + self.sizes[self.rank]]

    def reset(self, mode=None):
        """forget the posted buffers (between two timing trials) and optionally switch the placement of the collective; a
        collective still in flight on the side stream is waited for first (the context's stream is ordered behind it)"""
        for b in range(self.pairs):
            if self.work[b] is not None and self.mode == "async":
                self.ctx.check(self.ctx.lib.gsmcal_allgather_wait(self.ctx.h, b), "gsmcal_allgather_wait")
            self.work[b] = None
        if mode is not None:
            self.mode = mode


# ---- N > 1 decisions that every rank must take identically: which collective, and where it sits ----
def all_max(value, device="cpu", group=None):
    """MAX over the ranks of one float (the agreement primitive of the two functions below)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def call_with_timeout(fn, timeout_s):
    """fn() on a helper thread; TimeoutError when it has not returned after timeout_s seconds (the thread is left behind as a
    daemon: a bootstrap stuck inside ncclCommInitRank cannot be cancelled, only abandoned).  timeout_s None / <= 0: plain call."""
    if not timeout_s or timeout_s <= 0:
        return fn()
    import threading
    box = {}

    def run():
        try:
            box["v"] = fn()
        except BaseException as e:  # noqa: BLE001
            box["e"] = e

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        raise TimeoutError(f"no answer within {timeout_s:g} s")
    if "e" in box:
        raise box["e"]
    return box.get("v")


def choose_gatherer(make_native, make_torch, device="cpu", group=None, want="native", verify=None, timeout_s=90.0):
    """The table gatherer every rank of the job uses -- the SAME kind on every rank.

    want "torch": torch.distributed's collective (TableGatherer).  Otherwise `make_native()` builds the C ABI's own
    communicator + NativeTableGatherer and `verify(tg)` (optional) runs one checked trial exchange; either may raise or
    hang (timeout_s each).  The ranks then agree by one all-reduce: if ANY rank failed, ALL ranks fall back to
    `make_torch()` -- a rank never keeps a native communicator its peers gave up on.
    Returns (gatherer, kind, fallback_reason or None); kind is "torch" or `want`."""
    if want == "torch":
        return make_torch(), "torch", None
    tg, err = None, None
    try:
        tg = call_with_timeout(make_native, timeout_s)
        if verify is not None:
            call_with_timeout(lambda: verify(tg), timeout_s)
    except BaseException as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if all_max(0.0 if err is None else 1.0, device, group) > 0.0:
        return make_torch(), "torch", err or "native collective failed on another rank"
    return tg, want, None


def autotune_placement(tg, measure, device="cpu", group=None, margin=0.98, modes=("inline", "async")):
    """Where the native all-gather sits, decided by measurement, identically on every rank: in line on the chain's stream
    (costs the collective's latency per step) or on the side stream behind an event (hides it under the next step, costs
    the event).  `measure(mode)` -> seconds per step on THIS rank with tg.mode == mode (called in the same order on every
    rank: it contains collectives).  The MAX over ranks of each figure decides -- a rank that happens to prefer the other
    placement still follows -- and the second mode must win by `margin` to displace the first.
    Returns {"<mode>_ms_per_step": ..., "chosen": mode}."""
    worst = {}
    for m in modes:
        tg.reset(m)
        worst[m] = all_max(measure(m), device, group)
    chosen = modes[1] if worst[modes[1]] < margin * worst[modes[0]] else modes[0]
    tg.reset(chosen)
    out = {f"{m}_ms_per_step": round(1e3 * worst[m], 4) for m in modes}
    out["chosen"] = chosen
    return out


def autotune_choice(labels, measure, device="cpu", group=None, margin=0.98):
    """The fastest of several configurations, decided by measurement, identically on every rank.  `measure(label)` puts the job into
    configuration `label` and returns seconds per step on THIS rank (called in the same order on every rank: it contains collectives);
    the MAX over ranks of each figure decides, and a later label must beat the best so far by `margin` to displace it (the first
    label is the default).  Returns {"<label>_ms_per_step": ..., "chosen": label}."""
    worst = {m: all_max(measure(m), device, group) for m in labels}
    chosen = labels[0]
    for m in labels[1:]:
        if worst[m] < margin * worst[chosen]:
            chosen = m
    out = {f"{m}_ms_per_step": round(1e3 * worst[m], 4) for m in labels}
    out["chosen"] = chosen
    return out


def verify_gatherer(tg, cols, device, sync, timeout_s=60.0):
    """One checked exchange on both buffer pairs of a gatherer before the job commits to it: rank r sends rows filled with
    1000 r + row + col/16, every rank checks every peer's block.  `sync()` must complete the work enqueued so far (and may
    itself be guarded by the caller's timeout).  Raises on a wrong table."""
    import torch
    want = torch.cat([1000.0 * r + torch.arange(n, dtype=torch.float64).reshape(n, 1) + torch.arange(cols, dtype=torch.float64).reshape(1, cols) / 16.0
                      for r, n in enumerate(tg.sizes)], dim=0)
    n = tg.sizes[tg.rank]
    local = (1000.0 * tg.rank + torch.arange(n, dtype=torch.float64).reshape(n, 1) + torch.arange(cols, dtype=torch.float64).reshape(1, cols) / 16.0).to(device)
    npairs = getattr(tg, "pairs", 2)
    for b in range(npairs):
        tg.post(b, local)
    import contextlib
    on_stream = tg._on_stream() if hasattr(tg, "_on_stream") else contextlib.nullcontext()
    with on_stream:                                           # (the clones run on the stream the gatherer's own torch work runs on)
        got = [tg.rows(b).clone() for b in range(npairs)]
    call_with_timeout(sync, timeout_s)
    for b in range(npairs):
        if not torch.equal(got[b].cpu(), want):
            raise RuntimeError(f"trial all-gather returned a wrong table on rank {tg.rank} (buffer pair {b})")
        tg.work[b] = None


def check_gathered_table(gathered, local, sizes, rank, group=None):
    """The collective's result against what every rank says it sent: SHA-256 digests of the ranks' local rows travel through
    `group` (an independent group -- gloo over TCP in bench.py -- never the communicator under test), and this rank compares
    EVERY peer's block of its gathered table with that peer's digest, not only its own rows.  gathered: host array
    [sum(sizes), cols] in global unit order; local: this rank's rows [sizes[rank], cols].  Serves both tables of the ABI: the
    calibration table (10 columns, gsm_sync_demod.m:123-124) and the scanner's (snr, num_hit) table
    (multi_rtl_sdr_gsm_FCCH_scanner.m:163-186).  Raises AssertionError naming the rank whose rows differ; returns True."""
    import hashlib
    import torch.distributed as dist
    gathered = np.ascontiguousarray(gathered)
    local = np.ascontiguousarray(local)
    world = len(sizes)
    assert gathered.shape[0] == sum(sizes), f"rank {rank}: gathered table has {gathered.shape[0]} rows, the shards add up to {sum(sizes)}"
    assert local.shape[0] == sizes[rank] and local.shape[1:] == gathered.shape[1:]
    mine = hashlib.sha256(local.tobytes()).hexdigest()
    digests = [None] * world
    dist.all_gather_object(digests, mine, group=group)
    off = 0
    for r in range(world):
        blk = np.ascontiguousarray(gathered[off: off + sizes[r]])
        assert hashlib.sha256(blk.tobytes()).hexdigest() == digests[r], f"rank {rank}: rank {r}'s rows in the gathered table differ from what it sent"
        off += sizes[r]
    return True


def broadcast_unique_id(ctx, device, group=None):
    """Rank 0 draws an RCCL id and the process group broadcasts it: the ONE use of torch.distributed in the native bootstrap.
    Every rank makes exactly this one collective call whatever happens on rank 0 (a failure there travels as a flag byte), so
    the group's collective order stays the same on all ranks -- the joining itself (NativeComm) then touches no torch collective
    and may fail or hang on any one rank without desynchronising the group.  Returns the 128-byte id, or None if rank 0 had none."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    idt = torch.zeros(129, dtype=torch.uint8, device=device)
    if rank == 0:
        try:
            raw = NativeComm.unique_id(ctx)
            idt[:128].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
            idt[128] = 1
        except Exception:  # noqa: BLE001 - reported to every rank through the flag
            pass
    dist.broadcast(idt, 0, group=group)
    host = idt.cpu().numpy()
    return bytes(host[:128].tobytes()) if int(host[128]) == 1 else None


def native_comm_from_process_group(ctx, device, group=None, unique_id=None):
    """A NativeComm spanning the ranks of a torch.distributed process group: rank 0 draws the RCCL id, the group broadcasts it
    (broadcast_unique_id; pass `unique_id` when that already happened), every rank joins with ncclCommInitRank through the C ABI."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if unique_id is None:
        unique_id = broadcast_unique_id(ctx, device, group)
    if unique_id is None:
        raise RuntimeError("rank 0 could not draw an RCCL unique id")
    return NativeComm(ctx, world, rank, unique_id=unique_id)


def sampling_phase_difference(pos_info_a, pos_info_b):
    """gsm_sync_demod.m:151-158: per-burst start difference between two dongles' pos_info (8x units)."""
    a = np.atleast_2d(np.asarray(pos_info_a, dtype=np.float64))
    b = np.atleast_2d(np.asarray(pos_info_b, dtype=np.float64))
    n = min(len(a), len(b))
    return b[:n, 0] - a[:n, 0]


def _matlab_round(x):
    x = np.asarray(x, dtype=np.float64)
    return np.sign(x) * np.floor(np.abs(x) + 0.5)


def burst_map(pos_info, oversampling_ratio=8):
    """gsm_sync_demod.m:130-134: per GSM frame index (1-based, round(start / 10000) at 8x) the burst type found in
    slot 0 -- 0 FCCH, 1 SCH, 2 BCCH, NaN nothing; later types overwrite earlier ones as in the reference."""
    p = np.atleast_2d(np.asarray(pos_info, dtype=np.float64))
    if p.size == 0 or np.all(p == -1):
        return np.empty(0)
    per_frame = 1250.0 * oversampling_ratio
    idx = _matlab_round(p[:, 0] / per_frame).astype(np.int64)
    a = np.full(int(idx.max()), np.nan)
    for kind in (0, 1, 2):
        sel = idx[p[:, 1] == kind]
        a[sel[sel >= 1] - 1] = kind
    return a


def sampling_phase_frames(pos_info_a, pos_info_b, oversampling_ratio=8):
    """x axis of the reference's phase-difference plot (gsm_sync_demod.m:151-155): frame index of each compared burst,
    taken from the dongle with fewer rows."""
    a = np.atleast_2d(np.asarray(pos_info_a, dtype=np.float64))
    b = np.atleast_2d(np.asarray(pos_info_b, dtype=np.float64))
    short = a if len(a) <= len(b) else b
    n = min(len(a), len(b))
    return _matlab_round(short[:n, 0] / (1250.0 * oversampling_ratio))


def scan_frequency_plan(start_freq, end_freq, freq_step, num_dongle):
    """multi_rtl_sdr_gsm_FCCH_scanner.m:60-65: the ARFCN grid split across dongles.

    freq_orig = start:step:end, padded at the end (continuing the grid) to a multiple of num_dongle, then
    vec2mat(freq, num_freq_per_sub_band): row i = the consecutive sub-band dongle i sweeps.  Returns
    (freq[num_dongle, num_freq_per_sub_band], num_pad).  Unit index of (dongle i, point j) in the gathered
    scan table: i*num_freq_per_sub_band + j (the order of s_all / snr / num_hit, :69,:163-186)."""
    n = int(np.floor((end_freq - start_freq) / freq_step + 1e-9)) + 1           # length(start:step:end)
    freq_orig = start_freq + freq_step * np.arange(n, dtype=np.float64)
    per = -(-n // num_dongle)                                                   # ceil
    num_pad = per * num_dongle - n
    freq = np.concatenate([freq_orig, freq_orig[-1] + freq_step * np.arange(1, num_pad + 1)])
    return freq.reshape(num_dongle, per), num_pad


def scan_record(snr, num_hit, start_freq, end_freq, freq_step, num_dongle, gain, num_samples, sampling_rate, coef):
    """The fields the scanner saves (multi_rtl_sdr_gsm_FCCH_scanner.m:206-207), from the gathered scan table."""
    freq, _ = scan_frequency_plan(start_freq, end_freq, freq_step, num_dongle)
    snr = np.asarray(snr, dtype=np.float64).ravel()
    num_hit = np.asarray(num_hit, dtype=np.float64).ravel()
    if snr.size != freq.size or num_hit.size != freq.size:
        raise ValueError("scan table does not match the frequency plan")
    return {"snr": snr, "num_hit": num_hit, "start_freq": start_freq, "end_freq": end_freq, "freq_step": freq_step,
            "observe_time": num_samples / sampling_rate, "gain": gain, "sampling_rate": sampling_rate,
            "coef": np.asarray(coef, dtype=np.float64), "freq": freq,
            "filename": "FCCH_scan_%s_%s_gain%s_%sdongles.mat" % (_num2str(start_freq), _num2str(end_freq),
                                                                  _num2str(gain), _num2str(num_dongle))}


def _num2str(v):
    """MATLAB num2str for the integers / short decimals the scanner prints ('%.Ng' with N = digits + 4)."""
    if float(v) == int(v):
        return str(int(v))
    return ("%11.5g" % v).strip()
