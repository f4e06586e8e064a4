#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into the small text/JSON files kept under profiles/.

    python profiles/rocpd_summary.py stats  <results.db> <out.csv> [skip_first_n_per_kernel]
    python profiles/rocpd_summary.py pmc    <fetch.db> <write.db> <out.json> <streams> <samples_per_stream>

stats: per-kernel launch count, total / average / min / max duration (ns) -- the `--kernel-trace --stats` table.
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


def pmc_table(db, counter):
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
    return {k: sum(v.values()) / len(v) for k, v in per.items()}


def pmc(fetch_db, write_db, out, streams, samples):
    f, w = pmc_table(fetch_db, "FETCH_SIZE"), pmc_table(write_db, "WRITE_SIZE")
    raw, hbm = {}, {}
    for k in sorted(set(f) | set(w)):
        raw[k] = {"FETCH_SIZE": round(f.get(k, 0.0), 1), "WRITE_SIZE": round(w.get(k, 0.0), 1)}
        hbm[k] = int(round((2.0 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024))
    with open(out, "w") as fh:
        json.dump({"streams_per_gpu": streams, "samples_per_stream": samples,
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 5 "
                             "--warmup 2 --no-kernel-events`; KiB per launch averaged over launches; per MI355X_MICROARCH.md "
                             "(HBM section) FETCH_SIZE of a wide coalesced stream is doubled on gfx950, WRITE_SIZE as reported",
                   "raw_kib_per_launch": raw, "hbm_bytes_per_launch": hbm}, fh, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]))
