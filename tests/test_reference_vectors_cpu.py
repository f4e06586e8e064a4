"""Pinning the oracle to the reference itself ("a reference-pinned oracle").

The build image has no MATLAB / Octave and the reference ships no vectors, so this repository cannot produce
tests/golden/reference_vectors.json itself.  Whoever has MATLAB runs (see tests/golden/refvec.py):

    python tests/golden/refvec.py export tests/golden/captures
    matlab -batch "cd tests/golden; make_reference_vectors('/path/to/multi-rtl-sdr-calibration', 'captures')"

and commits the file.  With it present, BOTH CPU restatements are compared, stage by stage, with what the reference's own
functions returned for the same bytes; without it the test says "parity unpinned" (a warning, not a skip) and the
self-checks below keep the export and the comparison code working."""
import json
import os
import sys
import warnings

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import refvec  # noqa: E402
from oracle import gsmcal_oracle as o  # noqa: E402
from oracle import gsmcal_oracle_literal as lit  # noqa: E402

REF_FILE = os.path.join(HERE, "golden", "reference_vectors.json")


def _taps_and_template(doc):
    """The filter taps and SCH template the reference run USED are inputs of the path: take them from the file when it has
    them (fir1 / comm.GMSKModulator are toolbox code outside /root/reference), and report how far ours are from them."""
    import gsmcal
    synth = gsmcal.synth
    c46, c30, ts = synth.fir1(46, 200e3 / synth.FS), synth.fir1(30, 200e3 / synth.FS), np.asarray(synth.sch_training_sequence())
    notes = []
    if "coef46" in doc:
        r46, r30 = np.array(doc["coef46"], dtype=np.float64), np.array(doc["coef30"], dtype=np.float64)
        notes.append(f"fir1(46) max |ours - reference run's| = {np.max(np.abs(c46 - r46)):.3e}")
        c46, c30 = r46, r30
    if "sch_training_sequence" in doc:
        t = doc["sch_training_sequence"]
        rts = np.array(t["re"], dtype=np.float64) + 1j * np.array(t["im"], dtype=np.float64)
        notes.append(f"SCH template max |ours - reference run's| = {np.max(np.abs(ts - rts)):.3e}")
        ts = rts
    return c46, c30, ts, notes


def test_both_oracles_against_the_reference_run_when_present():
    if not os.path.exists(REF_FILE):
        warnings.warn("parity unpinned: tests/golden/reference_vectors.json absent (no MATLAB in the build image); "
                      "run tests/golden/make_reference_vectors.m against a checkout of the reference to pin both oracles")
        return
    doc = refvec.load(REF_FILE)
    c46, c30, ts, notes = _taps_and_template(doc)
    bad = []
    for spec in refvec.CAPTURES:
        name, kind = spec[0], spec[1]
        if name not in doc["captures"]:
            bad.append(f"{name}: missing from the reference run")
            continue
        raw = refvec.capture(spec)
        for label, mod in (("oracle", o), ("literal", lit)):
            mine = refvec.stage_vectors(mod, raw, kind, c46 if kind == "sync" else c30, ts)
            bad += refvec.compare(doc["captures"][name], mine, f"{label}:{name}")
    assert not bad, "oracle differs from the reference run:\n" + "\n".join(bad + notes)


def test_export_writes_what_the_harness_reads(tmp_path):
    out = refvec.export(str(tmp_path / "captures"))
    lines = open(os.path.join(out, "manifest.txt")).read().split("\n")[:-1]
    assert len(lines) == len(refvec.CAPTURES)
    for line, spec in zip(lines, refvec.CAPTURES):
        name, kind, n = line.split()
        assert (name, kind) == (spec[0], spec[1]) and int(n) == spec[4] * 10000
        assert os.path.getsize(os.path.join(out, name + ".bin")) == 2 * int(n)
    assert os.path.getsize(os.path.join(out, "fir1_46.f64")) == 47 * 8
    assert os.path.getsize(os.path.join(out, "sch_training_sequence_8x.f64")) == 512 * 16
    # the harness itself is part of the repository, calls the reference's functions by name and holds none of its code
    m = open(os.path.join(HERE, "golden", "make_reference_vectors.m")).read()
    for fn in ("raw2iq(", "FCCH_coarse_position(", "FCCH_fine_correction(", "SCH_corr_rate_correction(", "carrier_correct_post_SCH(",
               "total_ppm_calculation(", "addpath(ref_dir)"):
        assert fn in m


def test_comparison_lines_the_second_oracle_up_with_the_first_and_catches_a_wrong_value(tmp_path):
    """Self-check of the machinery on a file written by the vectorised oracle (NOT a pin): the literal restatement must
    agree with it on every stage, front-end probes included, and a falsified value must be reported."""
    import gsmcal
    synth = gsmcal.synth
    c46, c30, ts = synth.fir1(46, 200e3 / synth.FS), synth.fir1(30, 200e3 / synth.FS), synth.sch_training_sequence()
    doc = {"captures": {}}
    specs = [refvec.CAPTURES[0], refvec.CAPTURES[5]]
    for spec in specs:
        doc["captures"][spec[0]] = refvec.stage_vectors(o, refvec.capture(spec), spec[1], c46 if spec[1] == "sync" else c30, ts)
    path = tmp_path / "ref.json"
    path.write_text(json.dumps(refvec._jsonable(doc)))
    back = refvec.load(str(path))
    for spec in specs:
        mine = refvec.stage_vectors(lit, refvec.capture(spec), spec[1], c46 if spec[1] == "sync" else c30, ts)
        assert refvec.compare(back["captures"][spec[0]], mine, spec[0]) == []
    rec = back["captures"][specs[0][0]]
    assert rec["fcch_pos"][0] > 0 and rec["raw2iq"]["n"] == 1020000 and len(rec["filter"]["idx"]) == 32
    rec["fcch_pos"][1] += 1.0                                # one sample off in one position
    rec["total_carrier_ppm"] *= 1.0 + 3e-6                   # outside north_star's 1e-6
    rec["filter"]["re"][7] += 1e-6 * max(abs(v) for v in rec["filter"]["re"])
    mine = refvec.stage_vectors(o, refvec.capture(specs[0]), "sync", c46, ts)
    msgs = refvec.compare(rec, mine, "falsified")
    assert len(msgs) == 3 and any("fcch_pos" in m for m in msgs) and any("total_carrier_ppm" in m for m in msgs) and any("filter.re" in m for m in msgs)
