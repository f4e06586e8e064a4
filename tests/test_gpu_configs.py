"""GPU tests at the sizes of BASELINE.json configs 3 and 5 (-m gpu).

config 5: 256 dongles x 400 ARFCN over 8 GPUs = 12 800 captures x 640 000 IQ samples per GPU (16.4 GB of raw bytes,
resident in HBM, generated on the device); config 3: 2 dongles x 100 ARFCN through the scanner's own bookkeeping
(multi_rtl_sdr_gsm_FCCH_scanner.m:60-65 frequency split, :132-135 front end, :163-186 detect + accept, :206-207 record).
The captures come from gsmcal_synth_expand_dev: a seeded set of synthetic GSM captures expanded on the device into
distinct ones (rotation + counter-based dither); synth.expand_capture reproduces any of them bit for bit on the host,
so sampled captures of the 16 GB batch go through the CPU oracle without the batch ever existing on the host."""
import os

import numpy as np
import pytest

import parity
from oracle import gsmcal_oracle as o

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FRAMES = 64                      # multi_rtl_sdr_gsm_FCCH_scanner.m:39
N = FRAMES * 10000               # 640 000 complex samples per capture


def _base_captures(g, k, dongle0):
    """k seeded captures: three of four carry a BCCH carrier, a quarter of those at low SNR."""
    caps = []
    for i in range(k):
        kw = {"bcch": i % 4 != 3}
        if i % 8 == 1:
            kw["snr_db"] = 4.0 + i % 5
        caps.append(g.synth.make_stream(dongle=dongle0, arfcn=i, num_frames=FRAMES, **kw)[0])
    return np.stack(caps)


def _check_units(g, base, units, snr_numhit, positions, pos_snr, counts, coef, lo=0):
    for u in units:
        cap = g.synth.expand_capture(base, u)
        live = o.scan_capture(cap, coef)
        i = u - lo
        assert live["num_hit"] == snr_numhit[i, 1], f"unit {u}: num_hit {snr_numhit[i, 1]} vs oracle {live['num_hit']}"
        assert abs(live["snr"] - snr_numhit[i, 0]) < parity.SNR_ATOL, f"unit {u}: snr"
        n = counts[i]
        if live["coarse_pos"][0] == -1.0:
            assert n == 0 and positions[i, 0] == -1.0
        else:
            parity.assert_positions(positions[i, :n], live["coarse_pos"], f"unit {u}: positions")
            assert np.allclose(pos_snr[i, :n], live["coarse_snr"], rtol=0, atol=parity.SNR_ATOL)


def test_config5_scanner_batch_12800_captures(g_mod, ctx):
    g = g_mod
    D, K = 12800, 32
    coef = g.synth.fir1(30, 200e3 / g.synth.FS)                      # multi_rtl_sdr_gsm_FCCH_scanner.m:53
    base = _base_captures(g, K, 5000)
    H = g.MAX_HITS
    d_base = ctx.alloc(base.nbytes)
    d_raw = ctx.alloc(D * 2 * N)                                      # 16.4 GB
    d_out = ctx.alloc(D * 2 * 8)
    d_pos = ctx.alloc(D * H * 8)
    d_psn = ctx.alloc(D * H * 8)
    d_cnt = ctx.alloc(D * 4)
    try:
        ctx.h2d(d_base, base)
        g.synth_expand_dev(d_base, K, N, d_raw, D, first_unit=0, ctx=ctx)
        ctx.sync()
        # the device generator and its host twin agree byte for byte (first, a middle and the last capture)
        for u in (0, 7777, D - 1):
            got = np.empty(2 * N, dtype=np.uint8)
            ctx.d2h(got, d_raw + u * 2 * N)
            assert np.array_equal(got, g.synth.expand_capture(base, u)), f"device capture {u} differs from the host twin"

        def run(ptr, d, out_off=0):
            g.fcch_scan_batch_dev(ptr, d, N, coef, d_out + out_off * 16, d_pos + out_off * H * 8, d_psn + out_off * H * 8,
                                  d_cnt + out_off * 4, ctx=ctx)
            ctx.sync()
            sn = np.empty((d, 2)); ps = np.empty((d, H)); pn = np.empty((d, H)); cn = np.empty(d, dtype=np.int32)
            ctx.d2h(sn, d_out + out_off * 16); ctx.d2h(ps, d_pos + out_off * H * 8)
            ctx.d2h(pn, d_psn + out_off * H * 8); ctx.d2h(cn, d_cnt + out_off * 4)
            return sn, ps, pn, cn

        sn, ps, pn, cn = run(d_raw, D)
        # (1) 64 sampled captures against the CPU oracle: num_hit and positions exact, snr within 1e-8 dB
        rng = np.random.default_rng(5)
        units = sorted(set([0, 1, D - 1] + [int(x) for x in rng.integers(0, D, 61)]))
        _check_units(g, base, units, sn, ps, pn, cn, coef)
        assert np.sum(sn[:, 1] > 0) > D // 4, "most BCCH captures should be accepted"
        assert np.sum(sn[:, 1] == 0) > D // 8, "captures without a BCCH carrier must be rejected"
        # (2) determinism: the same batch again (replayed as a hipGraph) gives the same bits
        sn2, ps2, pn2, cn2 = run(d_raw, D)
        assert np.array_equal(sn, sn2) and np.array_equal(ps, ps2) and np.array_equal(pn, pn2) and np.array_equal(cn, cn2)
        # (3) units are independent: a sub-batch from the middle of the buffer (beyond 2^31 bytes) reproduces its rows
        lo, cnt = 9000, 300
        sn3, ps3, pn3, cn3 = run(d_raw + lo * 2 * N, cnt)
        assert np.array_equal(sn3, sn[lo:lo + cnt]) and np.array_equal(ps3, ps[lo:lo + cnt]) and np.array_equal(cn3, cn[lo:lo + cnt])
    finally:
        for p in (d_base, d_raw, d_out, d_pos, d_psn, d_cnt):
            ctx.free(p)


def test_config3_two_dongle_100_arfcn_sweep(g_mod, ctx):
    """multi_rtl_sdr_gsm_FCCH_scanner.m end to end for 2 dongles x 100 ARFCN: frequency plan (:60-65) -> 200 captures
    -> detect + accept (:163-186) -> saved record (:206-207), every field against the oracle."""
    g = g_mod
    from gsmcal import dist as gd
    num_dongle, start, step = 2, 935e6, 0.2e6
    end = start + step * 198.5                                         # 199 points: padded to 200 = 2 x 100 (:62-65)
    freq, num_pad = gd.scan_frequency_plan(start, end, step, num_dongle)
    assert freq.shape == (2, 100) and num_pad == 1
    D, K = freq.size, 20
    coef = g.synth.fir1(30, 200e3 / g.synth.FS)
    base = np.stack([g.synth.make_stream(dongle=6000, arfcn=i, num_frames=FRAMES, bcch=(i % 10 == 3 or i % 10 == 7))[0]
                     for i in range(K)])                              # a fifth of the ARFCNs carry a BCCH carrier
    d_base = ctx.alloc(base.nbytes)
    d_raw = ctx.alloc(D * 2 * N)
    try:
        ctx.h2d(d_base, base)
        g.synth_expand_dev(d_base, K, N, d_raw, D, first_unit=100000, ctx=ctx)
        ctx.sync()
        raw = np.empty((D, 2 * N), dtype=np.uint8)                    # unit i*100+j = (dongle i, sub-band point j), :69
        ctx.d2h(raw, d_raw)
    finally:
        ctx.free(d_base); ctx.free(d_raw)
    assert np.array_equal(raw[57], g.synth.expand_capture(base, 100057))
    out = g.fcch_scan_batch(raw, coef, ctx=ctx)                       # host-pointer entry point, like the driver's loop
    want_snr, want_hit = np.zeros(D), np.zeros(D)
    for u in range(D):
        live = o.scan_capture(raw[u], coef)
        want_snr[u], want_hit[u] = live["snr"], live["num_hit"]
        n = out["counts"][u]
        if live["coarse_pos"][0] != -1.0:
            parity.assert_positions(out["positions"][u, :n], live["coarse_pos"], f"unit {u}")
    assert np.array_equal(out["num_hit"], want_hit)
    assert np.max(np.abs(out["snr"] - want_snr)) < parity.SNR_ATOL
    assert 20 <= np.sum(want_hit > 0) <= 60
    rec = gd.scan_record(out["snr"], out["num_hit"], start, end, step, num_dongle, gain=0, num_samples=N,
                         sampling_rate=g.synth.FS, coef=coef)
    ref = gd.scan_record(want_snr, want_hit, start, end, step, num_dongle, gain=0, num_samples=N,
                         sampling_rate=g.synth.FS, coef=coef)
    for k in ("snr", "num_hit", "freq", "coef"):
        assert np.allclose(rec[k], ref[k], rtol=0, atol=parity.SNR_ATOL)
    assert rec["filename"] == ref["filename"] and rec["observe_time"] == N / g.synth.FS and rec["freq"].shape == (2, 100)


def test_ingest_ring_overlaps_copies_with_the_detector(g_mod, ctx):
    """batches written into the pinned ring (what recv_into() does), copied on the ring's stream while the
    previous batch is processed, consumed through the *_dev entry point -- results identical to the host-pointer call."""
    g = g_mod
    ing = g.ingest
    D, nbatch = 24, 5
    coef = g.synth.fir1(30, 200e3 / g.synth.FS)
    base = _base_captures(g, 12, 7000)
    batches = [np.stack([g.synth.expand_capture(base, 1000 * k + i) for i in range(D)]) for k in range(nbatch)]
    ring = ing.Ring(ctx, D * 2 * N, slots=2)
    d_out = ctx.alloc(nbatch * D * 16)
    try:
        for k in range(nbatch):
            s = k % 2
            if k >= 2:
                ring.host_ready(s)                                   # the slot's previous H2D has left the pinned buffer
            ring.host(s)[:] = batches[k].reshape(-1)                 # stand-in for the socket reader's recv_into()
            ring.submit(s)
            dev = ring.acquire(s)
            g.fcch_scan_batch_dev(dev, D, N, coef, d_out + k * D * 16, ctx=ctx)
            ring.release(s)
        ctx.sync()
        got = np.empty((nbatch, D, 2))
        ctx.d2h(got, d_out)
    finally:
        ring.close()
        ctx.free(d_out)
    for k in range(nbatch):
        ref = g.fcch_scan_batch(batches[k], coef, ctx=ctx)
        assert np.array_equal(got[k, :, 0], ref["snr"]) and np.array_equal(got[k, :, 1], ref["num_hit"]), f"batch {k}"
    assert np.sum(got[:, :, 1] > 0) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("fail_native", [False, True])
def test_bench_distributed_step_on_one_rank_and_its_fall_back(fail_native):
    """bench.py's N > 1 code path as the driver's 8-GPU run takes it, on one rank (GSMCAL_FORCE_DIST=1: process group, native
    communicator, one all-gather per step, digest check of the gathered table) -- and the fall-back to torch.distributed's
    collective on every rank together when the native communicator cannot be set up (it has never run on more than one rank
    in this project's own sessions; a failure there must not cost the scaling record)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GSMCAL_FORCE_DIST"] = "1"
    if fail_native:
        env["GSMCAL_BENCH_FAIL_NATIVE"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--streams", "4", "--distinct", "4",
           "--no-sub", "--no-cpu-baseline", "--no-kernel-events"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert p.returncode == 0 and line, p.stderr[-2000:]
    r = json.loads(line[-1])
    cfg = r["config"]
    assert cfg["gathered_table_checked_against_every_rank"] is True
    assert cfg["streams_calibrated_ok"] == 4
    if fail_native:
        assert cfg["collective"].endswith("torch.distributed over RCCL")
        assert "GSMCAL_BENCH_FAIL_NATIVE" in cfg["collective_fallback_from_native"]
    else:
        assert "native RCCL" in cfg["collective"] and "collective_fallback_from_native" not in cfg


def test_scanner_calls_in_flight_change_no_bit(g_mod, ctx):
    """gsmcal_ctx_set_pipeline_depth for the scanner path (round 6): single-stage batches side by side on the context's internal
    streams -- the detector of call i under the front kernel of call i+1.  Six calls three deep over two different capture sets, each
    into its own outputs: snr / num_hit, hit positions, their SNRs and counts bit for bit the one-call-at-a-time outputs; a batch big
    enough for the stage pipeline (joined, not pipelined) in between; and the host-buffer entry point on the pipelined context."""
    import torch
    g = g_mod
    dev = torch.device("cuda", 0)
    coef = g.synth.fir1(30, 200e3 / g.synth.FS)
    sets = [np.stack([g.synth.make_stream(dongle=7300 + k, arfcn=i, num_frames=40, bcch=i % 3 != 2)[0] for i in range(24)]) for k in range(2)]
    n = sets[0].shape[1] // 2
    refs = [g.fcch_scan_batch(s_, coef, ctx=ctx) for s_ in sets]
    H = g.MAX_HITS
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        cx = g.Context(0, stream=st.cuda_stream)
        try:
            raw_t = [torch.from_numpy(s_).to(dev) for s_ in sets]
            outs = [(torch.zeros((24, 2), dtype=torch.float64, device=dev), torch.zeros((24, H), dtype=torch.float64, device=dev),
                     torch.zeros((24, H), dtype=torch.float64, device=dev), torch.zeros((24,), dtype=torch.int32, device=dev)) for _ in range(6)]
            cx.set_pipeline_depth(3)
            for k in range(6):
                sn, ps, pn, cn = outs[k]
                g.fcch_scan_batch_dev(raw_t[k & 1].data_ptr(), 24, n, coef, sn.data_ptr(), ps.data_ptr(), pn.data_ptr(), cn.data_ptr(), ctx=cx)
            cx.sync()
            for k in range(6):
                ref = refs[k & 1]
                sn, ps, pn, cn = (t.cpu().numpy() for t in outs[k])
                assert np.array_equal(sn[:, 0], ref["snr"], equal_nan=True) and np.array_equal(sn[:, 1], ref["num_hit"]), k
                assert np.array_equal(ps, ref["positions"]) and np.array_equal(pn, ref["pos_snr"], equal_nan=True) and np.array_equal(cn, ref["counts"]), k
            again = g.fcch_scan_batch(sets[1], coef, ctx=cx)            # host-buffer entry point, pipelined context: joins by itself
            for key in ("snr", "num_hit", "positions", "pos_snr", "counts"):
                assert np.array_equal(np.asarray(again[key]), np.asarray(refs[1][key]), equal_nan=True), key
        finally:
            cx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fail_native", [False, True])
def test_bench_scan_workload_distributed_step_on_one_rank(fail_native):
    """BASELINE config 5's exchange as `bench.py --workload scan` runs it with N > 1, on one rank
    (GSMCAL_FORCE_DIST=1): process group, native communicator chosen by gsmcal.dist.choose_gatherer with its checked trial
    exchange, ONE all-gather of the (snr, num_hit) table per step (two columns), the gathered table on the host, every peer's
    block against its digest -- and the fall-back to torch's collective when the native set-up fails."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GSMCAL_FORCE_DIST"] = "1"
    if fail_native:
        env["GSMCAL_BENCH_FAIL_NATIVE"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "scan", "--steps", "3", "--warmup", "1", "--streams", "50",
           "--frames", "64", "--distinct", "8", "--no-cpu-baseline", "--no-kernel-events"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert p.returncode == 0 and line, p.stderr[-2000:]
    r = json.loads(line[-1])
    cfg = r["config"]
    assert cfg["gathered_table_checked_against_every_rank"] is True
    assert cfg["captures_total"] == 50 and cfg["captures_with_hits"] > 0 and r["parity_checked_captures"] >= 3
    if fail_native:
        assert cfg["collective"].endswith("torch.distributed over RCCL")
        assert "GSMCAL_BENCH_FAIL_NATIVE" in cfg["collective_fallback_from_native"]
    else:
        assert "native RCCL" in cfg["collective"] and "collective_fallback_from_native" not in cfg


@pytest.mark.parametrize("n_samples", [250000, 250003])
def test_scanner_pipeline_choices_change_no_bit(g_mod, ctx, monkeypatch, n_samples):
    """A scanner batch big enough for the pipeline (2 100 captures: 8 stages) through every way the library can schedule it --
    stages with the front kernel launched in two parts and the detector computing its SNRs in place (the default), one stage,
    stages with the two-kernel detector and unsplit front kernels, plain instead of non-temporal raw loads -- must give the same
    bits; sampled captures against the oracle.  250 003 samples: no capture after the first starts on a 16-byte boundary, so the
    any-geometry front kernel runs (its partial sums are addressed through the same per-part offsets)."""
    g = g_mod
    D, K, n = 2100, 8, n_samples
    coef = g.synth.fir1(30, 200e3 / g.synth.FS)
    base = np.stack([g.synth.make_stream(dongle=7100, arfcn=i, num_frames=26, bcch=i % 4 != 3,
                                         **({"snr_db": 5.0 + i} if i % 4 == 1 else {}))[0][: 2 * n] for i in range(K)])
    H = g.MAX_HITS
    d_base, d_raw = ctx.alloc(base.nbytes), ctx.alloc(D * 2 * n)
    d_out, d_pos, d_psn, d_cnt = ctx.alloc(D * 16), ctx.alloc(D * H * 8), ctx.alloc(D * H * 8), ctx.alloc(D * 4)
    variants = {"default": {}, "one_stage": {"GSMCAL_SCAN_STAGES": "1"},
                "two_kernel_detector_unsplit": {"GSMCAL_SCAN_STAGES": "2", "GSMCAL_SCAN_SPLIT": "0"},   # (1 050 captures per stage: more than one resident round -> k_coarse_snr + k_coarse_scan)
                "plain_loads_12_stages": {"GSMCAL_FRONT_NT": "0", "GSMCAL_SCAN_STAGES": "12"}}
    made = {}
    try:
        ctx.h2d(d_base, base)
        g.synth_expand_dev(d_base, K, n, d_raw, D, first_unit=0, ctx=ctx)
        ctx.sync()
        outs = {}
        for name, env in variants.items():
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            made[name] = cx = g.Context(0)
            for k in env:
                monkeypatch.delenv(k)
            for rep in range(3):                                      # eager, captured, replayed
                g.fcch_scan_batch_dev(d_raw, D, n, coef, d_out, d_pos, d_psn, d_cnt, ctx=cx)
            cx.sync()
            sn = np.empty((D, 2)); ps = np.empty((D, H)); pn = np.empty((D, H)); cn = np.empty(D, dtype=np.int32)
            cx.d2h(sn, d_out); cx.d2h(ps, d_pos); cx.d2h(pn, d_psn); cx.d2h(cn, d_cnt)
            outs[name] = (sn, ps, pn, cn)
        ref = outs["default"]
        for name, o_ in outs.items():
            for a, b in zip(ref, o_):
                assert np.array_equal(a, b, equal_nan=True), name
        rng = np.random.default_rng(11)
        units = sorted(set([0, 1, D - 1] + [int(x) for x in rng.integers(0, D, 13)]))
        _check_units(g, base, units, *ref, coef)
        assert np.sum(ref[3] > 0) > D // 4 and np.sum(ref[0][:, 1] > 0) > 0      # (26-frame captures: few reach the acceptance rule's hit count)
    finally:
        for cx in made.values():
            cx.close()
        for p_ in (d_base, d_raw, d_out, d_pos, d_psn, d_cnt):
            ctx.free(p_)


def test_scanner_batch_of_2100_distinct_captures_every_capture_against_the_oracle(g_mod, ctx):
    """a pipelined scanner batch (2 100 captures: 8 stages, front kernels launched in two parts, the detector
    computing its SNRs in place) with EVERY capture -- not a sample -- through oracle.scan_capture: num_hit and hit positions
    exact, snr and the hits' SNRs within 1e-8 dB.  The captures are distinct (gsmcal_synth_expand_dev; synth.expand_capture is
    the host twin the oracle's workers rebuild them with).  multi_rtl_sdr_gsm_FCCH_scanner.m:132-135,163-186."""
    g = g_mod
    D, K, n = 2100, 12, 260000
    coef = g.synth.fir1(30, 200e3 / g.synth.FS)
    base = np.stack([g.synth.make_stream(dongle=7300, arfcn=i, num_frames=26, bcch=i % 4 != 3,
                                         **({"snr_db": 4.0 + i} if i % 4 == 1 else {}))[0][: 2 * n] for i in range(K)])
    H = g.MAX_HITS
    d_base, d_raw = ctx.alloc(base.nbytes), ctx.alloc(D * 2 * n)
    d_out, d_pos, d_psn, d_cnt = ctx.alloc(D * 16), ctx.alloc(D * H * 8), ctx.alloc(D * H * 8), ctx.alloc(D * 4)
    try:
        ctx.h2d(d_base, base)
        g.synth_expand_dev(d_base, K, n, d_raw, D, first_unit=0, ctx=ctx)
        ctx.sync()
        per = 25
        want = [r for chunk in parity.pool_map(parity.scan_units_job, [(base, lo, min(per, D - lo), coef) for lo in range(0, D, per)],
                                               max_workers=128) for r in chunk]
        assert len(want) == D
        for rep in range(3):                                          # eager, captured, replayed
            g.fcch_scan_batch_dev(d_raw, D, n, coef, d_out, d_pos, d_psn, d_cnt, ctx=ctx)
            ctx.sync()
            sn = np.empty((D, 2)); ps = np.empty((D, H)); pn = np.empty((D, H)); cn = np.empty(D, dtype=np.int32)
            ctx.d2h(sn, d_out); ctx.d2h(ps, d_pos); ctx.d2h(pn, d_psn); ctx.d2h(cn, d_cnt)
            for u in range(D):
                live = want[u]
                assert live["num_hit"] == sn[u, 1], f"call {rep} unit {u}: num_hit {sn[u, 1]} vs oracle {live['num_hit']}"
                assert abs(live["snr"] - sn[u, 0]) < parity.SNR_ATOL, f"call {rep} unit {u}: snr"
                k = cn[u]
                if live["coarse_pos"][0] == -1.0:
                    assert k == 0 and ps[u, 0] == -1.0, f"call {rep} unit {u}: the oracle found no hit"
                else:
                    parity.assert_positions(ps[u, :k], live["coarse_pos"], f"call {rep} unit {u}: positions")
                    assert np.allclose(pn[u, :k], live["coarse_snr"], rtol=0, atol=parity.SNR_ATOL), f"call {rep} unit {u}: hit SNRs"
        assert np.sum(cn > 0) > D // 4
    finally:
        for p_ in (d_base, d_raw, d_out, d_pos, d_psn, d_cnt):
            ctx.free(p_)


def test_multigpu_preflight_script_on_one_rank():
    """tests/multigpu_check.py (the pre-flight for a node with several GPUs) with one rank: every collective path it covers --
    torch's, the native one through an id file, bench.py's NativeTableGatherer in both placements -- on real calibration
    tables; keeps the script from rotting while only one GPU is at hand."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multigpu_check.py"), "--gpus", "1", "--frames", "61"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "multigpu_check OK: rank 0/1" in p.stdout, p.stdout[-1500:] + p.stderr[-1500:]
