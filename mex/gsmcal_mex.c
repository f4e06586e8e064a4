/* gsmcal_mex.c -- MEX gateway: lets gsm_sync_demod.m / multi_rtl_sdr_gsm_FCCH_scanner.m call the
 * MI355X kernels through the reference's own function names, unchanged.
 *
 * NOT compiled in the build image (no MATLAB, no mex.h); it is the binding a maintainer of the
 * reference would add.  Build one MEX file per function, named like the .m file it shadows, and put
 * the output directory ahead of the reference on the MATLAB path:
 *
 *   for f in raw2iq chn_filter_8x_4x chn_filter_4x move_fft_snr_runtime_avg specific_fft_snr_fix_avg \
 *            FCCH_coarse_position FCCH_fine_correction SCH_corr_rate_correction \
 *            carrier_correct_post_SCH total_ppm_calculation gsmcal_calibrate gsmcal_fcch_scan; do
 *     mex -R2018a -DGSMCAL_FN_$f -output $f mex/gsmcal_mex.c -Iinclude -Lmulti-rtl-sdr-calibration_amd/lib -lgsmcal
 *   done
 *
 * The last two are not shadows of .m files: they are the fused entry points (one call per driver loop body)
 *   [table, pos_info] = gsmcal_calibrate(s, coef, sch_training_sequence, freq)   replaces gsm_sync_demod.m:107-124
 *   [snr, num_hit]    = gsmcal_fcch_scan(s, coef)                                replaces ..FCCH_scanner.m:132-135,163-186
 * with s the 2N x D uint8 matrix fread() delivers (gsm_sync_demod.m:96; pass uint8(s) if it was read as double).
 * tests/test_abi_cpu.py compiles every target against a declaration-only mex.h (tests/mex_stub) as a prototype check.
 *
 * Both MEX complex-storage APIs are handled: with -R2018a (MX_HAS_INTERLEAVED_COMPLEX = 1) complex arrays
 * are interleaved like the ABI's double[2] and are passed through without a copy; without the flag -- the only API of the
 * MATLAB releases the reference was written for (README: Ubuntu 12.04, R2008b-era .fda files) and still the default of
 * `mex` -- complex arrays are split (mxGetPr / mxGetPi) and the gateway interleaves inputs / de-interleaves outputs:
 *     mex -DGSMCAL_FN_$f -output $f mex/gsmcal_mex.c -Iinclude -L... -lgsmcal          (split API, any release)
 * Argument lists, 1-based positions, row/column shapes and sentinels follow the .m files
 * (cited per function); negative ABI status codes become mexErrMsgIdAndTxt errors, positive ones
 * (reference sentinels) return normally with the sentinel outputs, as the .m files do.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "gsmcal.h"
#include "mex.h"

static gsmcal_ctx* g_ctx = NULL;

static void at_exit(void) {
    if (g_ctx) { gsmcal_ctx_destroy(g_ctx); g_ctx = NULL; }
}

static gsmcal_ctx* ctx(void) {   /* created lazily, like the `persistent coef` of chn_filter_8x_4x.m:6-11 */
    if (!g_ctx) {
        if (gsmcal_ctx_create(0, &g_ctx) != 0) mexErrMsgIdAndTxt("gsmcal:nodevice", "no usable MI355X (gfx950) device");
        mexAtExit(at_exit);
    }
    return g_ctx;
}

static void chk(int rc, const char* what) {
    if (rc < 0) mexErrMsgIdAndTxt("gsmcal:error", "%s failed (%d): %s", what, rc, gsmcal_last_error(g_ctx));
}

/* The .m file this MEX file shadows disp()s its intermediate results (FCCH_coarse_position.m:92-94, FCCH_fine_correction.m:66,
 * 116,156-161,190, SCH_corr_rate_correction.m:80,118, carrier_correct_post_SCH.m:73-79, and a warning at every early exit): print
 * the same lines, from the library's report of the call just made (getenv GSMCAL_QUIET=1: run silent). */
static void say(void) {
    char buf[8192];
    const char* q = getenv("GSMCAL_QUIET");
    if (q && q[0] == '1') return;
    if (gsmcal_last_call_report(g_ctx, buf, sizeof(buf)) > 0) mexPrintf("%s", buf);
}

/* ---- the two complex-storage APIs behind one set of helpers ------------------------------------------------------ */
#if defined(MX_HAS_INTERLEAVED_COMPLEX) && MX_HAS_INTERLEAVED_COMPLEX
#define GSMCAL_INTERLEAVED 1
#define REAL_PTR(a) mxGetDoubles(a)
#define U8_PTR(a) ((const uint8_t*)mxGetUint8s(a))
#else
#define GSMCAL_INTERLEAVED 0
#define REAL_PTR(a) mxGetPr(a)
#define U8_PTR(a) ((const uint8_t*)mxGetData(a))
#endif

/* complex (or real, widened) array -> interleaved doubles; *n = number of elements.  Interleaved API: the array's own
 * storage when it is complex.  Split API: always a copy (freed by MATLAB when the MEX call returns, like every mxCalloc). */
static const double* cplx_in(const mxArray* a, mwSize* n) {
    *n = mxGetNumberOfElements(a);
#if GSMCAL_INTERLEAVED
    if (mxIsComplex(a)) return (const double*)mxGetComplexDoubles(a);
#endif
    {
        double* t = (double*)mxCalloc(2 * (*n) + 2, sizeof(double));
        const double* re = REAL_PTR(a);
        mwSize i;
        for (i = 0; i < *n; ++i) t[2 * i] = re[i];
#if !GSMCAL_INTERLEAVED
        if (mxIsComplex(a)) {
            const double* im = mxGetPi(a);
            for (i = 0; i < *n; ++i) t[2 * i + 1] = im[i];
        }
#endif
        return t;
    }
}

static mxArray* scalar(double v) { return mxCreateDoubleScalar(v); }

/* m x n complex output the ABI fills as interleaved doubles: cplx_out_begin() gives the buffer to hand to the ABI,
 * cplx_out_end() the finished mxArray (interleaved API: the array's own storage, no copy; split API: de-interleaved) */
typedef struct { mxArray* arr; double* buf; mwSize m, n; } cplx_out;
static double* cplx_out_begin(cplx_out* o, mwSize m, mwSize n) {
    o->m = m; o->n = n;
#if GSMCAL_INTERLEAVED
    o->arr = mxCreateDoubleMatrix(m, n, mxCOMPLEX);
    o->buf = (double*)mxGetComplexDoubles(o->arr);
#else
    o->arr = NULL;
    o->buf = (double*)mxMalloc((2 * m * n + 2) * sizeof(double));
#endif
    return o->buf;
}
static mxArray* cplx_out_end(cplx_out* o) {
#if !GSMCAL_INTERLEAVED
    mwSize i, tot = o->m * o->n;
    double *re, *im;
    o->arr = mxCreateDoubleMatrix(o->m, o->n, mxCOMPLEX);
    re = mxGetPr(o->arr); im = mxGetPi(o->arr);
    for (i = 0; i < tot; ++i) { re[i] = o->buf[2 * i]; im[i] = o->buf[2 * i + 1]; }
    mxFree(o->buf);
#endif
    return o->arr;
}

static mxArray* cplx_col(const double* data, mwSize n) {   /* n x 1 complex column from interleaved doubles */
    cplx_out o;
    double* b = cplx_out_begin(&o, n, 1);
    memcpy(b, data, 2 * n * sizeof(double));
    return cplx_out_end(&o);
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
#if defined(GSMCAL_FN_raw2iq)
    /* b = raw2iq(a)                                   raw2iq.m:5 */
    mwSize rows = mxGetM(prhs[0]), d = mxGetN(prhs[0]);
    cplx_out o;
    double* b = cplx_out_begin(&o, rows / 2, d);
    chk(gsmcal_raw2iq(ctx(), REAL_PTR(prhs[0]), (long)rows, (int)d, b), "raw2iq");
    plhs[0] = cplx_out_end(&o);

#elif defined(GSMCAL_FN_chn_filter_8x_4x)
    /* r = chn_filter_8x_4x(s)                         chn_filter_8x_4x.m:5 */
    mwSize n = mxGetM(prhs[0]), d = mxGetN(prhs[0]), tot;
    const double* s = cplx_in(prhs[0], &tot);
    cplx_out o;
    double* b = cplx_out_begin(&o, (n + 1) / 2, d);
    chk(gsmcal_chn_filter_8x_4x(ctx(), s, (long)n, (int)d, NULL, 0, b), "chn_filter_8x_4x");
    plhs[0] = cplx_out_end(&o);

#elif defined(GSMCAL_FN_chn_filter_4x)
    /* r = chn_filter_4x(s)                            chn_filter_4x.m:5 */
    mwSize n = mxGetM(prhs[0]), d = mxGetN(prhs[0]), tot;
    const double* s = cplx_in(prhs[0], &tot);
    cplx_out o;
    double* b = cplx_out_begin(&o, n, d);
    chk(gsmcal_chn_filter_4x(ctx(), s, (long)n, (int)d, NULL, 0, b), "chn_filter_4x");
    plhs[0] = cplx_out_end(&o);

#elif defined(GSMCAL_FN_move_fft_snr_runtime_avg)
    /* [hit_flag,hit_idx,hit_avg_snr,hit_snr] = move_fft_snr_runtime_avg(s,mv_len,fft_len,th) */
    mwSize n;
    const double* s = cplx_in(prhs[0], &n);
    int hf; double hi, ha, hs;
    chk(gsmcal_move_fft_snr_runtime_avg(ctx(), s, (long)n, (int)mxGetScalar(prhs[1]), (int)mxGetScalar(prhs[2]),
                                        mxGetScalar(prhs[3]), &hf, &hi, &ha, &hs), "move_fft_snr_runtime_avg");
    plhs[0] = mxCreateLogicalScalar(hf != 0);
    if (nlhs > 1) plhs[1] = scalar(hi);
    if (nlhs > 2) plhs[2] = scalar(ha);
    if (nlhs > 3) plhs[3] = scalar(hs);

#elif defined(GSMCAL_FN_specific_fft_snr_fix_avg)
    /* [hit_flag,hit_idx,hit_snr] = specific_fft_snr_fix_avg(s,target_set,fft_len,th,avg_snr) */
    mwSize n;
    const double* s = cplx_in(prhs[0], &n);
    int hf; double hi, hs;
    chk(gsmcal_specific_fft_snr_fix_avg(ctx(), s, (long)n, REAL_PTR(prhs[1]), (int)mxGetScalar(prhs[2]),
                                        mxGetScalar(prhs[3]), mxGetScalar(prhs[4]), &hf, &hi, &hs),
        "specific_fft_snr_fix_avg");
    plhs[0] = mxCreateLogicalScalar(hf != 0);
    if (nlhs > 1) plhs[1] = scalar(hi);
    if (nlhs > 2) plhs[2] = scalar(hs);

#elif defined(GSMCAL_FN_FCCH_coarse_position)
    /* [position,snr] = FCCH_coarse_position(s,decimation_ratio)     (row vectors; -1,-1 when none) */
    mwSize n;
    const double* s = cplx_in(prhs[0], &n);
    double pos[GSMCAL_MAX_HITS], snr[GSMCAL_MAX_HITS];
    int cnt = 0;
    chk(gsmcal_FCCH_coarse_position(ctx(), s, (long)n, (int)mxGetScalar(prhs[1]), pos, snr, GSMCAL_MAX_HITS, &cnt),
        "FCCH_coarse_position");
    say();
    plhs[0] = mxCreateDoubleMatrix(1, cnt, mxREAL);
    memcpy(REAL_PTR(plhs[0]), pos, cnt * sizeof(double));
    if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix(1, cnt, mxREAL); memcpy(REAL_PTR(plhs[1]), snr, cnt * sizeof(double)); }

#elif defined(GSMCAL_FN_FCCH_fine_correction)
    /* [FCCH_pos,r,sampling_ppm,carrier_ppm] = FCCH_fine_correction(s,base_position,ov,carrier_freq) */
    mwSize n;
    const double* s = cplx_in(prhs[0], &n);
    double pos[GSMCAL_MAX_HITS], sp, cp;
    int npos = 0; long lr = -1;
    double* r = (double*)mxMalloc(2 * n * sizeof(double));
    chk(gsmcal_FCCH_fine_correction(ctx(), s, (long)n, REAL_PTR(prhs[1]), (int)mxGetNumberOfElements(prhs[1]),
                                    (int)mxGetScalar(prhs[2]), mxGetScalar(prhs[3]), pos, GSMCAL_MAX_HITS, &npos,
                                    r, (long)n, &lr, &sp, &cp), "FCCH_fine_correction");
    say();
    plhs[0] = mxCreateDoubleMatrix(1, npos, mxREAL);
    memcpy(REAL_PTR(plhs[0]), pos, npos * sizeof(double));
    if (nlhs > 1) plhs[1] = lr < 0 ? scalar(-1.0) : cplx_col(r, (mwSize)lr);
    if (nlhs > 2) plhs[2] = scalar(sp);
    if (nlhs > 3) plhs[3] = scalar(cp);
    mxFree(r);

#elif defined(GSMCAL_FN_SCH_corr_rate_correction)
    /* [pos_info,r,sampling_ppm] = SCH_corr_rate_correction(s,FCCH_pos,sch_training_sequence,ov) */
    mwSize n = 0, nts;
    const double* s = mxGetNumberOfElements(prhs[0]) > 1 ? cplx_in(prhs[0], &n) : NULL;   /* r = -1 upstream */
    const double* ts = cplx_in(prhs[2], &nts);
    double pi[2 * GSMCAL_MAX_POS_ROWS], sp;
    int rows = 0, i; long lr = -1;
    double* r = n ? (double*)mxMalloc(2 * n * sizeof(double)) : NULL;
    chk(gsmcal_SCH_corr_rate_correction(ctx(), s, (long)n, REAL_PTR(prhs[1]), (int)mxGetNumberOfElements(prhs[1]),
                                        ts, (int)nts, (int)mxGetScalar(prhs[3]), pi, GSMCAL_MAX_POS_ROWS, &rows,
                                        r, (long)n, &lr, &sp), "SCH_corr_rate_correction");
    say();
    plhs[0] = mxCreateDoubleMatrix(rows, 2, mxREAL);
    for (i = 0; i < rows; ++i) {
        REAL_PTR(plhs[0])[i] = pi[i];
        REAL_PTR(plhs[0])[rows + i] = pi[GSMCAL_MAX_POS_ROWS + i];
    }
    if (nlhs > 1) plhs[1] = lr < 0 ? scalar(-1.0) : cplx_col(r, (mwSize)lr);
    if (nlhs > 2) plhs[2] = scalar(sp);
    if (r) mxFree(r);

#elif defined(GSMCAL_FN_carrier_correct_post_SCH)
    /* [r,carrier_ppm] = carrier_correct_post_SCH(s,pos_info,ov,carrier_freq) */
    mwSize n = 0;
    const double* s = mxGetNumberOfElements(prhs[0]) > 1 ? cplx_in(prhs[0], &n) : NULL;
    int rows = (int)mxGetM(prhs[1]);
    double cp; long lr = -1;
    double* r = n ? (double*)mxMalloc(2 * n * sizeof(double)) : NULL;
    chk(gsmcal_carrier_correct_post_SCH(ctx(), s, (long)n, REAL_PTR(prhs[1]), rows, rows, (int)mxGetScalar(prhs[2]),
                                        mxGetScalar(prhs[3]), r, (long)n, &lr, &cp), "carrier_correct_post_SCH");
    say();
    plhs[0] = lr < 0 ? scalar(-1.0) : cplx_col(r, (mwSize)lr);
    if (nlhs > 1) plhs[1] = scalar(cp);
    if (r) mxFree(r);

#elif defined(GSMCAL_FN_total_ppm_calculation)
    /* ppm_out = total_ppm_calculation(ppm_in) */
    double out;
    if (gsmcal_total_ppm_calculation(REAL_PTR(prhs[0]), (int)mxGetNumberOfElements(prhs[0]), &out) == GSMCAL_S_ALL_INF)
        mexPrintf("total PPM calculation: No valid PPM input!\n");          /* total_ppm_calculation.m:8 */
    plhs[0] = scalar(out);
#elif defined(GSMCAL_FN_gsmcal_calibrate)
    /* [table, pos_info] = gsmcal_calibrate(s, coef, sch_training_sequence, freq)      gsm_sync_demod.m:107-124 for all dongles
     * s: 2N x D uint8 (column = one dongle's interleaved I,Q bytes, exactly the ABI's capture-major layout);
     * table: D x 10 (columns: GSMCAL_T_*); pos_info: 1 x D cell, pos_info{i} = R x 2 (or the all -1 sentinel) */
    mwSize rows2n = mxGetM(prhs[0]), d = mxGetN(prhs[0]), nts, i, k;
    const double* ts = cplx_in(prhs[2], &nts);
    const mwSize ncf = mxGetNumberOfElements(prhs[3]);
    double* cf = (double*)mxMalloc(d * sizeof(double));
    double* tab = (double*)mxMalloc(d * GSMCAL_TABLE_COLS * sizeof(double));
    double* pi = (double*)mxMalloc(d * 2 * GSMCAL_MAX_POS_ROWS * sizeof(double));
    if (!mxIsUint8(prhs[0])) mexErrMsgIdAndTxt("gsmcal:type", "s must be uint8 (the bytes fread(...,'uint8') delivers)");
    for (i = 0; i < d; ++i) cf[i] = REAL_PTR(prhs[3])[ncf == d ? i : 0];
    chk(gsmcal_calibrate_batch(ctx(), U8_PTR(prhs[0]), (int)d, (long)(rows2n / 2), REAL_PTR(prhs[1]),
                               (int)mxGetNumberOfElements(prhs[1]), ts, (int)nts, cf, tab, pi, NULL, NULL), "gsmcal_calibrate");
    plhs[0] = mxCreateDoubleMatrix(d, GSMCAL_TABLE_COLS, mxREAL);
    for (i = 0; i < d; ++i)
        for (k = 0; k < GSMCAL_TABLE_COLS; ++k) REAL_PTR(plhs[0])[k * d + i] = tab[i * GSMCAL_TABLE_COLS + k];
    if (nlhs > 1) {
        plhs[1] = mxCreateCellMatrix(1, d);
        for (i = 0; i < d; ++i) {
            const mwSize r = (mwSize)tab[i * GSMCAL_TABLE_COLS + GSMCAL_T_N_POS_ROWS];
            mxArray* m = mxCreateDoubleMatrix(r, 2, mxREAL);
            for (k = 0; k < r; ++k) {
                REAL_PTR(m)[k] = pi[i * 2 * GSMCAL_MAX_POS_ROWS + k];
                REAL_PTR(m)[r + k] = pi[i * 2 * GSMCAL_MAX_POS_ROWS + GSMCAL_MAX_POS_ROWS + k];
            }
            mxSetCell(plhs[1], i, m);
        }
    }
    mxFree(cf); mxFree(tab); mxFree(pi);

#elif defined(GSMCAL_FN_gsmcal_fcch_scan)
    /* [snr, num_hit] = gsmcal_fcch_scan(s, coef)      multi_rtl_sdr_gsm_FCCH_scanner.m:132-135,163-186 for all captures
     * s: 2N x F uint8, one column per (dongle, frequency) capture in the order of s_all (:135); snr, num_hit: 1 x F */
    mwSize rows2n = mxGetM(prhs[0]), f = mxGetN(prhs[0]);
    if (!mxIsUint8(prhs[0])) mexErrMsgIdAndTxt("gsmcal:type", "s must be uint8 (the bytes fread(...,'uint8') delivers)");
    plhs[0] = mxCreateDoubleMatrix(1, f, mxREAL);
    {
        mxArray* nh = mxCreateDoubleMatrix(1, f, mxREAL);
        chk(gsmcal_fcch_scan_batch(ctx(), U8_PTR(prhs[0]), (int)f, (long)(rows2n / 2), REAL_PTR(prhs[1]),
                                   (int)mxGetNumberOfElements(prhs[1]), REAL_PTR(plhs[0]), REAL_PTR(nh), NULL, NULL, NULL),
            "gsmcal_fcch_scan");
        if (nlhs > 1) plhs[1] = nh;
    }
#else
#error "define one GSMCAL_FN_<function> (see the header comment)"
#endif
    (void)nrhs; (void)nlhs;
}
