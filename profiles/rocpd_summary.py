#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into the small text/JSON files kept under profiles/.

    python profiles/rocpd_summary.py stats  <results.db> <out.csv> [skip_first_n_per_kernel]
    python profiles/rocpd_summary.py pmc    <fetch.db> <write.db> <out.json> <streams> <samples_per_stream>
    python profiles/rocpd_summary.py sq     <sq.db> <out.csv>
    python profiles/rocpd_summary.py valu   <out.json> <regime>=<sq.db>:<launches per step | sN = N steps profiled in all> ...
    python profiles/rocpd_summary.py timeline <results.db> <out.csv> <last_n_dispatches>

stats: per-kernel launch count, total / average / min / max duration (ns) -- the `--kernel-trace --stats` table.
sq:    per-kernel means of every counter of an SQ pass (rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU
       SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY) plus derived columns: waves per launch, VALU
       instructions per wave, share of wave-cycles with a VALU instruction in flight, share spent waiting (s_waitcnt / barrier),
       LDS bank-conflict cycles per LDS instruction.  (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles.)
pmc:   FETCH_SIZE / WRITE_SIZE per launch (KiB as reported by rocprofv3), averaged over launches, plus the HBM
       bytes per launch after the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE of a wide coalesced
       streaming read reports half the bytes: doubled; WRITE_SIZE as reported).
"""
import csv
import json
import sqlite3
import sys


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


def stats(db, out, skip=0):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    agg = {}
    for name, s, e in rows:
        agg.setdefault(short(name), []).append(e - s)
    tot_all = sum(sum(v[skip:]) for v in agg.values())
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1][skip:])):
            v = v[skip:] or v
            w.writerow([k, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / tot_all, 2), min(v), max(v)])


def timeline(db, out, last_n):
    """start / end (us, relative to the first listed) of the last `last_n` kernel dispatches: who overlaps whom"""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()[-last_n:]
    t0 = rows[0][1]
    with open(out, "w") as f:
        f.write("kernel,start_us,end_us,duration_us\n")
        for name, s_, e_ in rows:
            f.write(f"{short(name)},{(s_ - t0) / 1e3:.1f},{(e_ - t0) / 1e3:.1f},{(e_ - s_) / 1e3:.1f}\n")


def pmc_table(db, counter, counts=None):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    val_col = "value" if "value" in cols else "counter_value"
    cn_col = "counter_name" if "counter_name" in cols else "pmc_name"
    rows = cur.execute(f"select {name_col}, {cn_col}, {val_col}, dispatch_id from counters_collection").fetchall()
    per = {}
    for name, cn, val, did in rows:
        if cn != counter:
            continue
        per.setdefault(short(name), {}).setdefault(did, 0.0)
        per[short(name)][did] += float(val)
    if counts is not None:
        counts.update({k: len(v) for k, v in per.items()})
    return {k: sum(v.values()) / len(v) for k, v in per.items()}


def sq(db, out):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    val_col = "value" if "value" in cols else "counter_value"
    cn_col = "counter_name" if "counter_name" in cols else "pmc_name"
    rows = cur.execute(f"select {name_col}, {cn_col}, {val_col}, dispatch_id from counters_collection").fetchall()
    per = {}
    for name, cn, val, did in rows:
        per.setdefault(short(name), {}).setdefault(cn, {}).setdefault(did, 0.0)
        per[short(name)][cn][did] += float(val)
    counters = sorted({cn for v in per.values() for cn in v})
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Launches"] + counters + ["valu_insts_per_wave", "valu_active_share_of_wave_cycles",
                                                      "wait_share_of_wave_cycles", "lds_conflict_cycles_per_lds_inst"])
        for k in sorted(per, key=lambda k: -sum(per[k].get("SQ_BUSY_CYCLES", {0: 0.0}).values())):
            m = {cn: (sum(per[k][cn].values()) / len(per[k][cn]) if cn in per[k] else 0.0) for cn in counters}
            n = max(len(v) for v in per[k].values())
            wv, wc = m.get("SQ_WAVES", 0.0), m.get("SQ_WAVE_CYCLES", 0.0)
            li = m.get("SQ_INSTS_LDS", 0.0)
            w.writerow([k, n] + [round(m[cn], 1) for cn in counters] +
                       [round(m.get("SQ_INSTS_VALU", 0.0) / wv, 1) if wv else "",
                        round(m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4) if wc else "",
                        round(m.get("SQ_WAIT_ANY", 0.0) / wc, 4) if wc else "",
                        round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / li, 3) if li else ""])


def pmc(fetch_db, write_db, out, streams, samples):
    launches = {}
    f, w = pmc_table(fetch_db, "FETCH_SIZE", launches), pmc_table(write_db, "WRITE_SIZE")
    raw, hbm = {}, {}
    for k in sorted(set(f) | set(w)):
        raw[k] = {"FETCH_SIZE": round(f.get(k, 0.0), 1), "WRITE_SIZE": round(w.get(k, 0.0), 1)}
        hbm[k] = int(round((2.0 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024))
    with open(out, "w") as fh:
        json.dump({"streams_per_gpu": streams, "samples_per_stream": samples,
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py (tools/profile_r03.sh); "
                             " KiB per launch averaged over launches; per MI355X_MICROARCH.md "
                             "(HBM section) FETCH_SIZE of a wide coalesced stream is doubled on gfx950, WRITE_SIZE as reported",
                   "raw_kib_per_launch": raw, "hbm_bytes_per_launch": hbm,
                   "launches": launches}, fh, indent=1)     # (kernels launched a handful of times belong to the stream selection / the one
                                                            #  unpipelined check call of bench.py, not to the timed loop's chain)


def valu(out, specs):
    """specs: regime=sq.db:launches_per_step ... -> wave-level VALU instructions per step of each regime (bench.py's
    roofline_compute.valu_issue_util reads this file)"""
    res = {"method": "rocprofv3 --pmc SQ_INSTS_VALU (the SQ pass of each regime): wave-level VALU instructions per launch, averaged over "
                     "launches, times the launches per step of the regime's plan; kernels seen fewer than 4 times (stream selection) left out",
           "peak_wave_instr_per_s": 256 * 4 * 2.4e9 / 4}
    for spec in specs:
        regime, rest = spec.split("=", 1)
        db, lps = rest.rsplit(":", 1)
        nsteps = 0
        if lps.startswith("s"):                      # ":s8" = the profiled command ran 8 steps in all (kernels launched in different sizes within a step)
            nsteps, lps = int(lps[1:]), 0
        lps = int(lps)
        cur = sqlite3.connect(db).cursor()
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        name_col = "kernel_name" if "kernel_name" in cols else "name"
        val_col = "value" if "value" in cols else "counter_value"
        cn_col = "counter_name" if "counter_name" in cols else "pmc_name"
        rows = cur.execute(f"select {name_col}, {cn_col}, {val_col}, dispatch_id from counters_collection").fetchall()
        per = {}
        for name, cn, val, did in rows:
            if cn != "SQ_INSTS_VALU" or not short(name).startswith("k_"):
                continue
            per.setdefault(short(name), {}).setdefault(did, 0.0)
            per[short(name)][did] += float(val)
        if nsteps:
            det = {k: int(round(sum(v.values()) / nsteps)) for k, v in per.items() if len(v) >= 4}
            res[regime] = {"valu_wave_instr_per_step": int(sum(det.values())), "per_kernel": det, "steps_profiled": nsteps,
                           "launches_per_kernel": {k: len(v) for k, v in per.items() if len(v) >= 4}}
        else:
            det = {k: int(round(sum(v.values()) / len(v) * lps)) for k, v in per.items() if len(v) >= 4}
            res[regime] = {"valu_wave_instr_per_step": int(sum(det.values())), "per_kernel": det, "launches_per_step_per_kernel": lps}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "timeline":
        timeline(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    elif sys.argv[1] == "valu":
        valu(sys.argv[2], sys.argv[3:])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    elif sys.argv[1] == "sq":
        sq(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]))
