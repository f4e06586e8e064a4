"""CPU tests of the boundary: the C-ABI library builds/loads, exports every symbol include/gsmcal.h
declares, host-only logic works, and the product path fails loudly without a GPU (no CPU fallback)."""
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "gsmcal.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gsmcal_[A-Za-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(gsmcal_mod):
    lib = gsmcal_mod.load()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/gsmcal.h but not exported by libgsmcal.so"
    assert sorted(gsmcal_mod.SIGNATURES) == syms, "ctypes prototypes and header disagree"
    assert b"gfx950" in lib.gsmcal_version()


def test_header_constants_match_python(gsmcal_mod):
    txt = open(os.path.join(ROOT, "include", "gsmcal.h")).read()
    assert int(re.search(r"#define GSMCAL_MAX_HITS (\d+)", txt).group(1)) == gsmcal_mod.MAX_HITS
    assert int(re.search(r"#define GSMCAL_TABLE_COLS (\d+)", txt).group(1)) == gsmcal_mod.TABLE_COLS
    assert gsmcal_mod.MAX_POS_ROWS == 6 * gsmcal_mod.MAX_HITS


def test_total_ppm_calculation_host_entry_point(gsmcal_mod):
    # total_ppm_calculation.m:5-21 is pure host arithmetic in the ABI (no device needed)
    from oracle import gsmcal_oracle as o
    for v in ([34.78, -1.08], [0.0, 0.0], [np.inf, np.inf], [12.5, np.inf], [-400.0, 3999.0]):
        a, b = gsmcal_mod.total_ppm_calculation(v), o.total_ppm_calculation(v)
        assert (math.isinf(a) and math.isinf(b)) or a == b


def test_no_cpu_fallback_context_fails_loudly_without_gpu(gsmcal_mod):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(gsmcal_mod.GsmcalError):
        gsmcal_mod.Context(0)


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "multi-rtl-sdr-calibration_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f"{f} imports the oracle"


def test_synth_is_seeded_and_template_is_unit_modulus(gsmcal_mod):
    s = gsmcal_mod.synth
    a, ta = s.make_stream(dongle=3, arfcn=7, num_frames=8)
    b, tb = s.make_stream(dongle=3, arfcn=7, num_frames=8)
    c, _ = s.make_stream(dongle=4, arfcn=7, num_frames=8)
    assert a.dtype == np.uint8 and a.shape == (2 * 8 * 10000,) and np.array_equal(a, b) and ta == tb
    assert not np.array_equal(a, c)
    ts = s.sch_training_sequence()
    assert ts.shape == (512,) and np.allclose(np.abs(ts), 1.0)
    # FCCH = all-zero bits -> pure tone at +symbol_rate/4 (target_freq of FCCH_fine_correction.m:157)
    x = s.gmsk_modulate(s.diff_precode(np.zeros(148, dtype=np.int8)))
    step = np.angle(x[200:1000] * np.conj(x[199:999]))
    w = 2 * np.pi * (s.SYMBOL_RATE / 4) / s.FS
    assert np.allclose(step, w, atol=1e-4) and abs(step.mean() - w) < 1e-12   # ripple of the truncated pulse


MEX_TARGETS = ("raw2iq", "chn_filter_8x_4x", "chn_filter_4x", "move_fft_snr_runtime_avg", "specific_fft_snr_fix_avg",
               "FCCH_coarse_position", "FCCH_fine_correction", "SCH_corr_rate_correction", "carrier_correct_post_SCH",
               "total_ppm_calculation", "gsmcal_calibrate", "gsmcal_fcch_scan")


@pytest.mark.parametrize("api", ["interleaved", "split"])
@pytest.mark.parametrize("target", MEX_TARGETS)
def test_mex_gateway_compiles_against_the_abi(target, api):
    """mex/gsmcal_mex.c cannot be BUILT here (no MATLAB): every target goes through `gcc -fsyntax-only` against a
    declaration-only mex.h (tests/mex_stub) -- syntax and every call into include/gsmcal.h are checked, once per MEX
    complex-storage API (interleaved = -R2018a; split = mxGetPr / mxGetPi, the API of the reference's era: SURVEY 8b)."""
    import subprocess
    r = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-std=c99", f"-DGSMCAL_FN_{target}"] +
                       (["-DGSMCAL_STUB_SPLIT"] if api == "split" else []) +
                       ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "mex_stub"),
                        os.path.join(ROOT, "mex", "gsmcal_mex.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_mex_split_complex_helpers_round_trip(tmp_path):
    """The split-API helpers of the gateway (interleave on the way in, de-interleave on the way out) run for real: the
    gateway source is compiled with the split stub plus a few-line mxArray implementation, and a complex column goes
    in through cplx_in() and back out through cplx_col()."""
    import subprocess
    drv = tmp_path / "drv.c"
    drv.write_text(r'''
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#define GSMCAL_FN_total_ppm_calculation 1
#include "gsmcal_mex.c"
struct mxArray_tag { size_t m, n; int cplx; double *pr, *pi; };
size_t mxGetM(const mxArray* a) { return a->m; }
size_t mxGetN(const mxArray* a) { return a->n; }
size_t mxGetNumberOfElements(const mxArray* a) { return a->m * a->n; }
int mxIsComplex(const mxArray* a) { return a->cplx; }
int mxIsUint8(const mxArray* a) { (void)a; return 0; }
double mxGetScalar(const mxArray* a) { return a->pr[0]; }
double* mxGetPr(const mxArray* a) { return a->pr; }
double* mxGetPi(const mxArray* a) { return a->pi; }
void* mxGetData(const mxArray* a) { return a->pr; }
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity f) {
    mxArray* a = calloc(1, sizeof(*a)); a->m = m; a->n = n; a->cplx = f == mxCOMPLEX;
    a->pr = calloc(m * n + 1, sizeof(double)); a->pi = a->cplx ? calloc(m * n + 1, sizeof(double)) : NULL; return a; }
mxArray* mxCreateDoubleScalar(double v) { mxArray* a = mxCreateDoubleMatrix(1, 1, mxREAL); a->pr[0] = v; return a; }
mxArray* mxCreateLogicalScalar(mxLogical v) { return mxCreateDoubleScalar(v); }
mxArray* mxCreateCellMatrix(mwSize m, mwSize n) { return mxCreateDoubleMatrix(m, n, mxREAL); }
void mxSetCell(mxArray* c, mwIndex i, mxArray* v) { (void)c; (void)i; (void)v; }
void* mxMalloc(size_t n) { return malloc(n); }
void* mxCalloc(size_t n, size_t s) { return calloc(n, s); }
void mxFree(void* p) { free(p); }
int mexAtExit(void (*fn)(void)) { (void)fn; return 0; }
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) { (void)id; (void)fmt; exit(3); }
int gsmcal_ctx_create(int d, gsmcal_ctx** o) { (void)d; (void)o; return -1; }
void gsmcal_ctx_destroy(gsmcal_ctx* c) { (void)c; }
const char* gsmcal_last_error(gsmcal_ctx* c) { (void)c; return ""; }
long gsmcal_last_call_report(gsmcal_ctx* c, char* b, size_t n) { (void)c; (void)b; (void)n; return 0; }
int mexPrintf(const char* fmt, ...) { (void)fmt; return 0; }
int gsmcal_total_ppm_calculation(const double* in, int n, double* out) { (void)in; (void)n; *out = 0; return 0; }
int main(void) {
    mwSize n = 0, i;
    mxArray* a = mxCreateDoubleMatrix(5, 1, mxCOMPLEX);
    for (i = 0; i < 5; ++i) { a->pr[i] = 1.5 + i; a->pi[i] = -0.25 * i; }
    const double* inter = cplx_in(a, &n);
    if (n != 5) return 1;
    for (i = 0; i < 5; ++i) if (inter[2 * i] != 1.5 + i || inter[2 * i + 1] != -0.25 * i) return 2;
    mxArray* b = cplx_col(inter, n);
    if (!b->cplx || b->m != 5 || b->n != 1) return 4;
    for (i = 0; i < 5; ++i) if (b->pr[i] != a->pr[i] || b->pi[i] != a->pi[i]) return 5;
    mxArray* r = mxCreateDoubleMatrix(3, 1, mxREAL);          /* a real input is widened with zero imaginary parts */
    r->pr[0] = 7; r->pr[1] = 8; r->pr[2] = 9;
    inter = cplx_in(r, &n);
    if (n != 3 || inter[0] != 7 || inter[1] != 0 || inter[4] != 9 || inter[5] != 0) return 6;
    (void)ctx; (void)chk; (void)scalar;
    puts("split helpers ok");
    return 0;
}
''')
    exe = tmp_path / "drv"
    r = subprocess.run(["gcc", "-std=c99", "-DGSMCAL_STUB_SPLIT", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "tests", "mex_stub"), "-I" + os.path.join(ROOT, "mex"), str(drv), "-o", str(exe), "-lm"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "split helpers ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_mex_gateway_covers_every_reference_function():
    src = open(os.path.join(ROOT, "mex", "gsmcal_mex.c")).read()
    for t in MEX_TARGETS:
        assert f"defined(GSMCAL_FN_{t})" in src


def test_params_defaults_are_the_reference_literals(gsmcal_mod):
    p = gsmcal_mod._lib.Params()
    gsmcal_mod.load().gsmcal_params_default(p)
    assert (p.coarse_th_db, p.coarse_mv_factor, p.coarse_max_offset) == (10.0, 10, 5)          # FCCH_coarse_position.m:21,22,45
    assert (p.min_hits, p.fine_max_offset, p.fine_max_ppm, p.fine_gate_snr_db) == (5, 64, 4000.0, 5.0)   # FCCH_fine_correction.m:12,30,83,192
    assert (p.sch_max_offset, p.sch_max_ppm, p.post_min_bcch) == (8, 400.0, 4)               # SCH_corr..m:36,94; carrier_correct..m:15
    assert (p.scan_min_hits, p.scan_spacing, p.scan_spacing_idle, p.scan_tol) == (3, 12500.0, 13750.0, 50.0)   # ..FCCH_scanner.m:169-176


def test_named_library_is_never_rebuilt_and_must_exist(tmp_path):
    """GSMCAL_LIB names another build of the library (tools/ab_session.sh, tools/devtiming.py): it is loaded as it is -- a missing
    file is an error, not a reason to compile the current sources into that name (which made two 'different' builds equal)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import importlib.util, os\n"
            "spec = importlib.util.spec_from_file_location('b', os.path.join(%r, 'multi-rtl-sdr-calibration_amd', 'build.py'))\n"
            "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            "try:\n"
            "    b.needs_build(); print('no error')\n"
            "except RuntimeError as e:\n"
            "    print('raised', e)\n") % (ROOT, ROOT)
    env = dict(os.environ, GSMCAL_LIB=str(tmp_path / "absent.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
    assert out.startswith("raised") and "absent.so" in out
    assert not (tmp_path / "absent.so").exists()


def test_committed_counter_profiles_describe_the_kernels_in_the_tree():
    """roofline.traffic / valu_* in bench.py's line are read from committed rocprofv3 counter passes.  The newest
    ones must have been taken on exactly these sources: tools/profile.sh records the hash of csrc/ + include/gsmcal.h next to
    them, and a kernel edit without a profile refresh turns this test red (bench.py then withholds the figures and says STALE)."""
    import glob
    import json
    import gsmcal
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = gsmcal.build.csrc_hash()
    for suffix in ("pmc_traffic.json", "valu_per_step.json"):
        newest = sorted(glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]_" + suffix)))[-1]
        with open(newest) as f:
            prof = json.load(f)
        assert prof.get("csrc_sha256") == want, f"{os.path.basename(newest)} was taken on other kernel sources: run tools/profile.sh and commit its summaries"


def test_num2str_formats_like_matlab():
    """The console diagnostics (gsmcal_last_call_report) format numbers with the library's restatement of MATLAB's num2str for row
    vectors: integers as %{digits+2}d, everything else as %{d+7}.{d}g with d = max(floor(log10(max|x|)) + 5, 5), trimmed.
    Known MATLAB answers (documentation examples and the forms the reference's disp() lines produce)."""
    import gsmcal
    want = {(3.14159265,): "3.1416", (1, 2, 3): "1  2  3", (67708.333, 67710.125): "67708.333       67710.125", (-12.3456,): "-12.3456",
            (1e-5,): "1e-05", (100000,): "100000", (99999, 100000, -5): "99999   100000       -5", (0.001234,): "0.001234",
            (12500, 13750, 12500): "12500  13750  12500", (-1,): "-1", (34.78260869565217,): "34.7826"}
    for x, s_ in want.items():
        assert gsmcal.num2str(list(x)) == s_, (x, gsmcal.num2str(list(x)))
