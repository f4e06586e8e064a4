"""world_size-2 gloo test of the multi-GPU layer: block-contiguous sharding of units and the single
all-gather of the per-unit table (on the GPU box the same code runs over RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unit_row(u, cols):
    return np.array([u * 100.0 + c for c in range(cols)])


def _worker(rank, world, num_units, port, q):
    sys.path.insert(0, ROOT)
    import gsmcal
    from gsmcal import dist as gd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = gd.shard_range(num_units, world, rank)
    local = torch.tensor(np.stack([_unit_row(u, gsmcal.TABLE_COLS) for u in range(lo, hi)]).reshape(hi - lo, gsmcal.TABLE_COLS))
    full = gd.allgather_table(local, num_units)
    q.put((rank, full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_units", [8, 7])
def test_allgather_table_world2(num_units):
    import gsmcal
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + num_units
    procs = [ctx.Process(target=_worker, args=(r, world, num_units, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([_unit_row(u, gsmcal.TABLE_COLS) for u in range(num_units)])
    for r in range(world):
        assert np.array_equal(res[r], want)


def _gatherer_worker(rank, world, num_units, port, q):
    sys.path.insert(0, ROOT)
    import gsmcal
    from gsmcal import dist as gd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = gd.shard_sizes(num_units, world)
    lo, hi = gd.shard_range(num_units, world, rank)
    tg = gd.TableGatherer(sizes, gsmcal.TABLE_COLS, torch.device("cpu"))
    outs = []
    for step in range(5):                                   # bench.py's loop: post step i, wait for it two steps later
        b = step & 1
        tg.wait(b)
        local = torch.tensor(np.stack([_unit_row(u, gsmcal.TABLE_COLS) + 1000.0 * step for u in range(lo, hi)]))
        tg.post(b, local)
        if step >= 1:
            outs.append(tg.rows(1 - b).clone().numpy())    # the previous step's table, complete by now
            assert np.array_equal(tg.own_rows(1 - b).numpy(), outs[-1][lo:hi])
    outs.append(tg.rows((5 - 1) & 1).clone().numpy())
    q.put((rank, np.stack(outs)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_units", [7, 64])
def test_bench_exchange_step_under_gloo(num_units):
    """The distributed half of bench.py (gsmcal.dist.TableGatherer: double-buffered, padded all-gather of the table) on
    two gloo ranks, with the uneven 7-unit split (4 + 3 rows) and BASELINE config 4's 64 streams (strong scaling)."""
    import gsmcal
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + num_units
    procs = [ctx.Process(target=_gatherer_worker, args=(r, world, num_units, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert res[r].shape == (5, num_units, gsmcal.TABLE_COLS)
        for step in range(5):
            want = np.stack([_unit_row(u, gsmcal.TABLE_COLS) + 1000.0 * step for u in range(num_units)])
            assert np.array_equal(res[r][step], want), f"rank {r} step {step}"


def _scan_row(u):
    """(snr, num_hit) of capture u as the scanner's acceptance rule leaves it: -1 / 0 for a capture without an FCCH"""
    return np.array([-1.0, 0.0]) if u % 4 == 3 else np.array([17.25 + 0.125 * u, 3.0 + (u % 3)])


def _scan_exchange_worker(rank, world, num_caps, corrupt, port, q):
    sys.path.insert(0, ROOT)
    from gsmcal import dist as gd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = gd.shard_sizes(num_caps, world)
    lo, hi = gd.shard_range(num_caps, world, rank)
    tg = gd.TableGatherer(sizes, 2, torch.device("cpu"))
    chk = dist.new_group(backend="gloo")
    outs = [torch.zeros((hi - lo, 2), dtype=torch.float64) for _ in range(2)]
    host_gath = [None, None]
    nstep = 0
    for _ in range(5):                                       # bench.py --workload scan: the step of bench_scan()
        b = nstep & 1
        nstep += 1
        tg.wait(b)
        outs[b].copy_(torch.tensor(np.stack([_scan_row(u) for u in range(lo, hi)]).reshape(hi - lo, 2)) + 1000.0 * nstep)   # the "kernel" fills buffer b
        tg.post(b, outs[b])
    for b in range(2):                                       # its fence: the GATHERED table to the host
        if tg.work[b] is not None:
            host_gath[b] = tg.rows(b).clone().numpy()
    last = (nstep - 1) & 1
    g_all = host_gath[last]
    if corrupt and rank == 1:
        g_all = g_all.copy()
        g_all[0, 0] += 1e-9                                  # one bit of rank 0's block arrives wrong on rank 1
    res = {"rank": rank, "rows": g_all.copy()}
    try:
        gd.check_gathered_table(g_all, outs[last].numpy(), sizes, rank, group=chk)
        res["ok"] = True
    except AssertionError as e:
        res["ok"], res["why"] = False, str(e)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_caps,corrupt", [(13, False), (400, False), (13, True)])
def test_scan_exchange_step_under_gloo(num_caps, corrupt):
    """BASELINE config 5's exchange (multi_rtl_sdr_gsm_FCCH_scanner.m:60-65,163-186): bench.py --workload scan's
    step -- the (snr, num_hit) table of this rank's captures into one of two buffers, ONE all-gather of it per step, the gathered
    table to the host at the fence -- on two gloo ranks with an UNEVEN capture split (13 -> 6 + 7), followed by the check every
    rank makes of every peer's block against that peer's digest (gsmcal.dist.check_gathered_table).  A single wrong bit in a
    peer's block on one rank must be caught there and only there."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + (os.getpid() % 2000) + num_caps % 89 + (7 if corrupt else 0)
    procs = [ctx.Process(target=_scan_exchange_worker, args=(r, world, num_caps, corrupt, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([_scan_row(u) for u in range(num_caps)]) + 1000.0 * 5
    assert res[0]["ok"] and np.array_equal(res[0]["rows"], want)
    if corrupt:
        assert not res[1]["ok"] and "rank 0's rows" in res[1]["why"]
    else:
        assert res[1]["ok"] and np.array_equal(res[1]["rows"], want)


class _FakeNative:
    """stands in for NativeTableGatherer in the decision-logic tests: same post / rows / reset surface, moves the rows with
    a process group of its OWN (as the native communicator is a communicator of its own: a rank stuck in it does not
    disturb the job's group, on which the ranks agree what to do)"""

    def __init__(self, sizes, cols, rank, group, mode="inline"):
        self.sizes, self.rank, self.mode, self.cols, self.group = list(sizes), rank, mode, cols, group
        self.mx = max(self.sizes)
        self.work = [None, None]
        self.out = [None, None]
        self.resets = []

    def post(self, b, local):
        send = torch.full((self.mx, self.cols), float("nan"), dtype=torch.float64)
        send[: local.shape[0]] = local
        recv = torch.zeros((len(self.sizes) * self.mx, self.cols), dtype=torch.float64)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        self.out[b] = torch.cat([recv[r * self.mx: r * self.mx + n] for r, n in enumerate(self.sizes)], dim=0)
        self.work[b] = True

    def rows(self, b):
        return self.out[b]

    def reset(self, mode=None):
        self.work = [None, None]
        if mode is not None:
            self.mode = mode
        self.resets.append(mode)


def _decision_worker(rank, world, scenario, port, q):
    sys.path.insert(0, ROOT)
    import time

    import gsmcal
    from gsmcal import dist as gd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = gd.shard_sizes(7, world)
    cols = gsmcal.TABLE_COLS
    side = dist.new_group(backend="gloo", timeout=__import__("datetime").timedelta(seconds=20))

    def make_native():
        if scenario == "rank1_setup_raises" and rank == 1:
            raise RuntimeError("ncclCommInitRank: unhandled system error (planted)")
        if scenario == "rank1_setup_hangs" and rank == 1:
            time.sleep(30.0)
        return _FakeNative(sizes, cols, rank, side)

    def verify(tg):
        if scenario == "rank0_trial_wrong" and rank == 0:
            raise RuntimeError("trial all-gather returned a wrong table (planted)")
        gd.verify_gatherer(tg, cols, "cpu", lambda: None)

    tg, kind, why = gd.choose_gatherer(make_native, lambda: gd.TableGatherer(sizes, cols, torch.device("cpu")), "cpu",
                                       want="torch" if scenario == "torch_asked" else "native", verify=verify, timeout_s=3.0)
    out = {"kind": kind, "why": why, "type": type(tg).__name__}
    if kind == "native":
        # the two ranks measure OPPOSITE preferences: the max over ranks must decide, identically
        if scenario == "opposite_preferences":
            cost = {"inline": 0.100 if rank == 0 else 0.300, "async": 0.200 if rank == 0 else 0.150}
        elif scenario == "async_wins":
            cost = {"inline": 0.300, "async": 0.100 + 0.01 * rank}
        else:
            cost = {"inline": 0.100 + 0.001 * rank, "async": 0.099}        # within the margin: the first mode stays
        seen = []

        def measure(mode):
            assert tg.mode == mode
            seen.append(mode)
            dist.barrier()                                   # (a measurement contains collectives: same order on every rank)
            return cost[mode]

        out["tune"] = gd.autotune_placement(tg, measure, "cpu")
        out["mode"], out["seen"] = tg.mode, seen
    # whatever was chosen must move a table correctly on both ranks
    lo, hi = gd.shard_range(7, world, rank)
    local = torch.tensor(np.stack([_unit_row(u, cols) for u in range(lo, hi)]))
    tg.post(0, local)
    out["table_ok"] = bool(np.array_equal(tg.rows(0).numpy(), np.stack([_unit_row(u, cols) for u in range(7)])))
    q.put((rank, out))
    dist.barrier()
    if out["why"] and "TimeoutError" in out["why"]:
        # a helper thread is still stuck inside the abandoned trial exchange: tearing the process group down under it aborts.
        # Leave without the tear-down (what bench.py does after such a fall-back, once its line is printed).
        q.close()
        q.join_thread()
        os._exit(0)
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["all_fine", "rank1_setup_raises", "rank1_setup_hangs", "rank0_trial_wrong", "torch_asked",
                                      "opposite_preferences", "async_wins"])
def test_every_rank_takes_the_same_collective_decision(scenario):
    """bench.py's N > 1 decisions (gsmcal.dist.choose_gatherer / autotune_placement) on two gloo ranks --
    the native set-up fails or hangs on ONE rank, the trial exchange fails on one rank, the two ranks measure opposite
    placements: both ranks must end on the same gatherer kind and the same placement, and the table must still arrive."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + (hash(scenario) % 97)
    procs = [ctx.Process(target=_decision_worker, args=(r, world, scenario, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res[0], res[1]
    assert a["kind"] == b["kind"] and a["type"] == b["type"], (a, b)
    assert a["table_ok"] and b["table_ok"]
    if scenario in ("rank1_setup_raises", "rank1_setup_hangs", "rank0_trial_wrong"):
        assert a["kind"] == "torch" and a["type"] == "TableGatherer"
        failed, other = (b, a) if scenario != "rank0_trial_wrong" else (a, b)
        assert "planted" in failed["why"] or "TimeoutError" in failed["why"], failed
        # the healthy rank either learns of it at the agreement step or times out in a trial exchange its peer never joined
        assert other["why"] == "native collective failed on another rank" or "TimeoutError" in other["why"], other
    elif scenario == "torch_asked":
        assert a["kind"] == "torch" and a["why"] is None and b["why"] is None
    else:
        assert a["kind"] == "native" and a["why"] is None
        assert a["tune"] == b["tune"] and a["mode"] == b["mode"] == a["tune"]["chosen"]
        assert a["seen"] == b["seen"] == ["inline", "async"]
        if scenario == "opposite_preferences":           # max over ranks: inline 0.300, async 0.200
            assert a["tune"] == {"inline_ms_per_step": 300.0, "async_ms_per_step": 200.0, "chosen": "async"}
        elif scenario == "async_wins":
            assert a["tune"]["chosen"] == "async"
        else:
            assert a["tune"]["chosen"] == "inline"


def _choice_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from gsmcal import dist as gd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the two ranks measure different favourites: the MAX over ranks decides, identically on both; a later label must win by the margin
    cost = [{"inline_depth4": 0.230, "async_depth4": 0.140, "inline_depth1": 0.180},
            {"inline_depth4": 0.150, "async_depth4": 0.260, "inline_depth1": 0.181}][rank]
    seen = []

    def measure(label):
        seen.append(label)
        dist.barrier()
        return cost[label]
    out = gd.autotune_choice(list(cost), measure, "cpu")
    tie = gd.autotune_choice(["a", "b"], lambda m: {"a": 0.100, "b": 0.099}[m], "cpu")      # within the margin: the first stays
    q.put((rank, (out, seen, tie)))
    dist.barrier()
    dist.destroy_process_group()


def test_depth_and_placement_are_chosen_identically_on_every_rank():
    """bench.py (N > 1, round 6): which of (collective in line | on the side stream) x (steps in flight | one at a time) the timed loop
    runs is measured, max over ranks (gsmcal.dist.autotune_choice) -- two gloo ranks with opposite favourites end on the same one."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 38500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_choice_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]
    out, seen, tie = res[0]
    assert seen == ["inline_depth4", "async_depth4", "inline_depth1"]
    assert out == {"inline_depth4_ms_per_step": 230.0, "async_depth4_ms_per_step": 260.0, "inline_depth1_ms_per_step": 181.0, "chosen": "inline_depth1"}
    assert tie["chosen"] == "a"


def test_shard_ranges_cover_all_units():
    from gsmcal import dist as gd
    for u in (1, 7, 64, 102400):
        for w in (1, 2, 3, 8):
            spans = [gd.shard_range(u, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == u
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(gd.shard_sizes(u, w)) - min(gd.shard_sizes(u, w)) <= 1


def test_sampling_phase_difference():
    from gsmcal import dist as gd
    a = np.array([[100.0, 0], [10100.0, 1], [20100.0, 2]])
    b = np.array([[103.0, 0], [10103.0, 1]])
    assert np.array_equal(gd.sampling_phase_difference(a, b), [3.0, 3.0])


def test_scan_frequency_plan_matches_the_scanner_grid():
    from gsmcal import dist as gd
    # multi_rtl_sdr_gsm_FCCH_scanner.m:28-34,60-65: 935:0.2:960 MHz = 126 points; 2 dongles -> 63 each, no padding
    freq, pad = gd.scan_frequency_plan(935e6, 960e6, 0.2e6, 2)
    assert freq.shape == (2, 63) and pad == 0
    assert freq[0, 0] == 935e6 and abs(freq[1, -1] - 960e6) < 1e-3 and abs(freq[1, 0] - (935e6 + 63 * 0.2e6)) < 1e-3
    # 4 dongles: ceil(126/4) = 32 per sub-band, 2 padded points continuing the grid past end_freq
    freq, pad = gd.scan_frequency_plan(935e6, 960e6, 0.2e6, 4)
    assert freq.shape == (4, 32) and pad == 2 and abs(freq[3, -1] - (960e6 + 2 * 0.2e6)) < 1e-3
    rec = gd.scan_record(np.zeros(128), np.zeros(128), 935e6, 960e6, 0.2e6, 4, 0, 640000, 8 * 1625e3 / 6, np.ones(31))
    assert rec["filename"] == "FCCH_scan_935000000_960000000_gain0_4dongles.mat"
    assert abs(rec["observe_time"] - 640000 / (8 * 1625e3 / 6)) < 1e-15 and rec["freq"].shape == (4, 32)


def test_burst_map_and_phase_frames():
    from gsmcal import dist as gd
    # FCCH at frame 1, SCH at frame 2, BCCH at frames 3-4 (8x: 10 000 samples per frame)
    pi = np.array([[10010.0, 0], [20010.0, 1], [30010.0, 2], [40010.0, 2]])
    m = gd.burst_map(pi, 8)
    assert len(m) == 4 and list(m) == [0.0, 1.0, 2.0, 2.0]
    pi2 = np.array([[60010.0, 0]])
    m2 = gd.burst_map(pi2, 8)
    assert len(m2) == 6 and np.all(np.isnan(m2[:5])) and m2[5] == 0
    assert gd.burst_map(np.array([[-1.0, -1.0]])).size == 0
    assert list(gd.sampling_phase_frames(pi, pi[:2] + [3.0, 0], 8)) == [1.0, 2.0]


def _exchange(lib, path, nonce, world, rank, ident, timeout):
    import ctypes as C
    buf = C.create_string_buffer(ident, 128)
    rc = lib.gsmcal_comm_id_file_exchange(str(path).encode(), nonce, world, rank, buf, timeout)
    return rc, buf.raw


@pytest.mark.parametrize("nonce", [0x1234ABCD5678, 0])
def test_id_file_bootstrap_ignores_a_stale_file(tmp_path, nonce):
    """ADVICE r2 (gsmcal_comm_init_file): a file left at the path by an earlier run must not be taken for this run's id.
    Two ranks (threads; the C call releases the GIL), the reader starts first with a stale record already in place -- a
    record of the old format, one with another nonce, and (nonce 0) one whose age is beyond the stale window."""
    import struct
    import threading
    import time

    import gsmcal
    lib = gsmcal.load()
    path = tmp_path / "rccl_id"
    stale_id = bytes([0xEE]) * 128
    fresh_id = bytes(range(128))
    if nonce:
        path.write_bytes(struct.pack("<QQ", 0x3144494C41434D47, nonce ^ 1) + stale_id)      # right format, another launch
    else:
        path.write_bytes(struct.pack("<QQ", 0x3144494C41434D47, 0) + stale_id)
        old = time.time() - 3600.0
        os.utime(path, (old, old))                                                          # an hour old: beyond the window
    got = {}

    def reader():
        got["r"] = _exchange(lib, path, nonce, 2, 1, bytes(128), 20.0)

    t = threading.Thread(target=reader)
    t.start()
    time.sleep(0.5)                                         # the reader has been polling the stale file for a while
    assert t.is_alive(), "the reader accepted a stale id file"
    rc0, _ = _exchange(lib, path, nonce, 2, 0, fresh_id, 20.0)
    t.join(timeout=30)
    assert rc0 == 0 and got["r"][0] == 0
    assert got["r"][1] == fresh_id
    # rank 0's clean-up after ncclCommInitRank: nothing is left for the next launch to trip over
    assert lib.gsmcal_comm_id_file_remove(str(path).encode()) == 0 and not path.exists()
    # a bare 128-byte file (the round-2 format) is never accepted, and a reader without a writer times out
    path.write_bytes(stale_id)
    rc, _ = _exchange(lib, path, nonce, 2, 1, bytes(128), 0.3)
    assert rc < 0


def test_default_launch_nonce_comes_from_the_launcher_environment(monkeypatch):
    """gsmcal_comm_init_file without an explicit nonce must not fall back to "any record younger than 120 s" when
    the launcher identifies the launch: the nonce is GSMCAL_COMM_NONCE, else a hash of the run / job id and the rendezvous
    address; two launches differ, the ranks of one launch agree, and only a bare environment gives 0."""
    import gsmcal
    lib = gsmcal.load()
    for k in ("GSMCAL_COMM_NONCE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "SLURM_JOB_ID", "SLURM_STEP_ID", "PBS_JOBID",
              "LSB_JOBID", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    assert lib.gsmcal_comm_default_nonce() == 0
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29500")
    a = lib.gsmcal_comm_default_nonce()
    assert a != 0 and a == lib.gsmcal_comm_default_nonce()
    monkeypatch.setenv("MASTER_PORT", "29501")
    b = lib.gsmcal_comm_default_nonce()
    assert b not in (0, a)
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "run-7")
    c = lib.gsmcal_comm_default_nonce()
    assert c not in (0, a, b)
    monkeypatch.setenv("GSMCAL_COMM_NONCE", "0x1234")
    assert lib.gsmcal_comm_default_nonce() == 0x1234


def test_plain_torchrun_environment_keeps_the_age_test(tmp_path, monkeypatch):
    """under plain `torchrun` TORCHELASTIC_RUN_ID is the literal "none" and MASTER_ADDR:MASTER_PORT the static
    127.0.0.1:29500, so every launch derives the SAME nonce.  "none" must count as absent (and the restart count must tell
    two attempts of one run apart), and a record carrying a derived nonce must still be refused when it is older than the
    stale window: an hour-old file planted with exactly this launch's derived nonce is not accepted."""
    import struct
    import threading
    import time

    import gsmcal
    lib = gsmcal.load()
    for k in ("GSMCAL_COMM_NONCE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "SLURM_JOB_ID", "SLURM_STEP_ID", "PBS_JOBID",
              "LSB_JOBID", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29500")
    bare = lib.gsmcal_comm_default_nonce()
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "0")
    assert lib.gsmcal_comm_default_nonce() == bare, "RUN_ID=none must identify nothing"
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")
    assert lib.gsmcal_comm_default_nonce() not in (0, bare), "a restarted attempt must not share the first attempt's nonce"
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "0")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job-42")
    assert lib.gsmcal_comm_default_nonce() not in (0, bare)
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")
    nonce = lib.gsmcal_comm_default_nonce()
    assert nonce == bare != 0
    # an id file a crashed bootstrap of an EARLIER plain-torchrun launch left behind: same derived nonce, an hour old
    path = tmp_path / "rccl_id"
    stale_id, fresh_id = bytes([0xEE]) * 128, bytes(range(128))
    path.write_bytes(struct.pack("<QQ", 0x3144494C41434D47, nonce) + stale_id)
    old = time.time() - 3600.0
    os.utime(path, (old, old))
    import ctypes as C

    def aged(rank, ident, timeout):
        buf = C.create_string_buffer(ident, 128)
        return lib.gsmcal_comm_id_file_exchange_aged(str(path).encode(), nonce, 2, rank, buf, timeout), buf.raw

    got = {}
    t = threading.Thread(target=lambda: got.__setitem__("r", aged(1, bytes(128), 20.0)))
    t.start()
    time.sleep(0.5)
    assert t.is_alive(), "a reader with a derived nonce accepted an hour-old record of the same nonce"
    assert aged(0, fresh_id, 20.0)[0] == 0
    t.join(timeout=30)
    assert got["r"] == (0, fresh_id)
    # the caller-chosen-nonce entry point keeps its contract: the nonce alone decides (an old record of THIS nonce is taken)
    path.write_bytes(struct.pack("<QQ", 0x3144494C41434D47, 0x77) + stale_id)
    os.utime(path, (old, old))
    assert _exchange(lib, path, 0x77, 2, 1, bytes(128), 2.0) == (0, stale_id)


def test_plain_bench_gpus_n_starts_its_own_ranks_and_fails_clearly_without_the_devices():
    """`python bench.py --gpus N` (no launcher) must start N ranks itself; on a node with fewer devices
    every rank says so and the command exits non-zero -- after spawning, not on a launcher check."""
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k != "RANK"})
    assert p.returncode != 0
    # two children were spawned (the launcher's failure report names local_rank 1) and a child got as far as counting devices
    # (the launcher terminates the other as soon as the first one fails: the second message may or may not make it out)
    assert p.stderr.count("--gpus 2 needs 2 devices") >= 1 and "local_rank: 1" in p.stderr, p.stderr[-2000:]
    assert "must be launched with" not in p.stderr
