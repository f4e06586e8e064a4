/* gsmcal.h -- C ABI of libgsmcal.so: the GSM FCCH+SCH calibration DSP chain of
 * JiaoXianjun/multi-rtl-sdr-calibration on AMD MI355X (gfx950), hand-written HIP kernels.
 *
 * The reference has no FFI layer: the path sits behind MATLAB function calls
 * (gsm_sync_demod.m:107-124, multi_rtl_sdr_gsm_FCCH_scanner.m:132-135,164).  Each entry point
 * below replaces one reference function -- the file:line it replaces is cited -- with the same
 * argument meaning, the same 1-based positions held in doubles, and the same sentinel outputs
 * (position = -1, r = -1, ppm = inf, pos_info = [-1 -1]).  A MEX gateway (mex/gsmcal_mex.c) or the
 * ctypes mirror (multi-rtl-sdr-calibration_amd/api.py) binds them; see INTEGRATION.md.
 *
 * Conventions
 *   - complex data: interleaved double[2] (re, im), column-major like MATLAB's interleaved API.
 *   - host entry points (no suffix) take HOST pointers, copy, run on the GPU, and return after the
 *     stream has drained.  `_dev` entry points take DEVICE pointers, only enqueue work on the
 *     context's HIP stream and return; call gsmcal_sync() before reading results.
 *   - return value: 0 = ok; > 0 = the reference's algorithmic sentinel was produced (outputs hold
 *     the sentinel values, exactly as the .m file would return them); < 0 = API / HIP failure
 *     (outputs undefined).  Where MATLAB itself would stop with an index error (SURVEY 8a pitfall
 *     12) the call returns GSMCAL_E_INDEX instead of reading out of bounds.
 *   - a context is bound to one GPU and one HIP stream and is not thread-safe.
 *   - there is no CPU fallback: without a usable gfx950 device gsmcal_ctx_create fails.
 */
#ifndef GSMCAL_H
#define GSMCAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gsmcal_ctx gsmcal_ctx;

/* ---- status codes ---------------------------------------------------------------------------- */
enum {
    GSMCAL_OK = 0,
    /* > 0: algorithmic sentinels (the reference returns normally with sentinel outputs) */
    GSMCAL_S_NO_FCCH = 1,         /* FCCH_coarse_position.m:27-30   position = snr = -1          */
    GSMCAL_S_FEW_HITS = 2,        /* FCCH_fine_correction.m:12-15 / SCH_corr_rate_correction.m:11 */
    GSMCAL_S_FINE_FEW = 3,        /* FCCH_fine_correction.m:69: fewer than 5 fine positions       */
    GSMCAL_S_FINE_SPACING = 4,    /* FCCH_fine_correction.m:95-102  FCCH_pos = -1                 */
    GSMCAL_S_FINE_FEW_BURSTS = 5, /* FCCH_fine_correction.m:135-142 <5 bursts left, carrier skipped*/
    GSMCAL_S_FINE_LOW_SNR = 6,    /* FCCH_fine_correction.m:192-196 FCCH_pos = -1                 */
    GSMCAL_S_SCH_EDGE = 7,        /* SCH_corr_rate_correction.m:59-63 pos_info = [-1 -1]          */
    GSMCAL_S_SCH_FEW = 8,         /* SCH_corr_rate_correction.m:84: fewer than 5 SCH positions    */
    GSMCAL_S_SCH_SPACING = 9,     /* SCH_corr_rate_correction.m:106-112                           */
    GSMCAL_S_POST_NO_POS = 10,    /* carrier_correct_post_SCH.m:10-13                             */
    GSMCAL_S_POST_FEW_BCCH = 11,  /* carrier_correct_post_SCH.m:15-19                             */
    GSMCAL_S_ALL_INF = 12,        /* total_ppm_calculation.m:7-11                                 */
    /* < 0: failures */
    GSMCAL_E_ARG = -1,
    GSMCAL_E_HIP = -2,
    GSMCAL_E_NO_DEVICE = -3,
    GSMCAL_E_CAPACITY = -4,       /* an output buffer capacity argument is too small             */
    GSMCAL_E_INDEX = -5,          /* MATLAB would raise "index exceeds matrix dimensions"         */
    GSMCAL_E_UNSUPPORTED = -6
};

#define GSMCAL_MAX_HITS 24          /* capacity for FCCH/SCH hits per stream (ceil(len/1562.5)) */
#define GSMCAL_MAX_POS_ROWS (6 * GSMCAL_MAX_HITS)
#define GSMCAL_TABLE_COLS 10

/* columns of one row of the calibration table (doubles; this row is what ranks all-gather) */
enum {
    GSMCAL_T_SAMPLING_PPM_FCCH = 0, /* FCCH_fine_correction sampling_ppm      (gsm_sync_demod.m:118) */
    GSMCAL_T_SAMPLING_PPM_SCH = 1,  /* SCH_corr_rate_correction sampling_ppm  (:119)                 */
    GSMCAL_T_CARRIER_PPM_FCCH = 2,  /* FCCH_fine_correction carrier_ppm       (:118)                 */
    GSMCAL_T_CARRIER_PPM_POST = 3,  /* carrier_correct_post_SCH carrier_ppm   (:120)                 */
    GSMCAL_T_TOTAL_SAMPLING_PPM = 4,/* total_ppm_calculation                  (:123)                 */
    GSMCAL_T_TOTAL_CARRIER_PPM = 5, /* total_ppm_calculation                  (:124)                 */
    GSMCAL_T_N_FCCH = 6,            /* length(FCCH_pos) after FCCH_fine_correction (1 if sentinel)   */
    GSMCAL_T_N_POS_ROWS = 7,        /* rows of pos_info (1 if sentinel)                              */
    GSMCAL_T_FIRST_FCCH_POS = 8,    /* pos_info(1,1) or -1                                           */
    GSMCAL_T_STATUS = 9             /* first non-zero status code met along the chain               */
};

/* ---- algorithm thresholds (the constants the reference hard-codes inside its functions) ---------- */
/* Defaults = the reference's literals.  gsmcal_set_params() takes effect from the next call on; the fields marked
 * "geometry" size windows and launch shapes and must keep their default (any other value: GSMCAL_E_UNSUPPORTED). */
typedef struct gsmcal_params {
    double coarse_th_db;        /* FCCH_coarse_position.m:21          th = 10            hit iff snr - avg > th            */
    int coarse_mv_factor;       /* FCCH_coarse_position.m:22          mv_len = 10*fft_len                     (geometry)  */
    int coarse_max_offset;      /* FCCH_coarse_position.m:45          max_offset = 5                          (geometry)  */
    int min_hits;               /* FCCH_fine_correction.m:12,69,142; SCH_corr_rate_correction.m:11,84        5            */
    int fine_max_offset;        /* FCCH_fine_correction.m:30          max_offset = 64 symbols                 (geometry)  */
    double fine_max_ppm;        /* FCCH_fine_correction.m:83          max_ppm = 4000     spacing classes                  */
    double fine_gate_snr_db;    /* FCCH_fine_correction.m:192         FCCH_snr < 5  ->  FCCH_pos = -1                     */
    double fine_noise_bw_hz;    /* FCCH_fine_correction.m:22          200e3: half_noise_len                   (geometry)  */
    int sch_max_offset;         /* SCH_corr_rate_correction.m:36      max_offset = 8 symbols                  (geometry)  */
    double sch_max_ppm;         /* SCH_corr_rate_correction.m:94      max_ppm = 400                                       */
    int post_min_bcch;          /* carrier_correct_post_SCH.m:15      fewer than 4 BCCH rows -> r = -1                    */
    int scan_min_hits;          /* multi_rtl_sdr_gsm_FCCH_scanner.m:169  at least 3 hits                                  */
    double scan_spacing;        /* ..FCCH_scanner.m:171               12500 (1x symbols between FCCH bursts)              */
    double scan_spacing_idle;   /* ..FCCH_scanner.m:176               12500 + 1250 across the idle frame                  */
    double scan_tol;            /* ..FCCH_scanner.m:171,176           50                                                  */
} gsmcal_params;
void gsmcal_params_default(gsmcal_params* p);
int gsmcal_set_params(gsmcal_ctx* ctx, const gsmcal_params* p);
int gsmcal_get_params(gsmcal_ctx* ctx, gsmcal_params* p);

/* ---- context --------------------------------------------------------------------------------- */
/* Creates a context on GPU `device_id` with its own non-blocking HIP stream. */
int gsmcal_ctx_create(int device_id, gsmcal_ctx** out);
/* Same, but enqueue on an existing hipStream_t (passed as void*; 0 = the default stream). */
int gsmcal_ctx_create_on_stream(int device_id, void* hip_stream, gsmcal_ctx** out);
void gsmcal_ctx_destroy(gsmcal_ctx* ctx);
int gsmcal_sync(gsmcal_ctx* ctx);
const char* gsmcal_last_error(gsmcal_ctx* ctx);
/* The reference's console diagnostics.  The .m files disp() their intermediate results -- "FCCH coarse: hit successive
 * 10 FCCH. pos ...", "FCCH fine: first round diff ...", "FCCH fine: FCCH freq ...", "SCH: sampling error ppm ...", the warnings of
 * every early exit (FCCH_coarse_position.m:6,28,92-94, FCCH_fine_correction.m:6,13,66,96-99,116,156-161,190-193,
 * SCH_corr_rate_correction.m:6,12,60,80,107-110,118, carrier_correct_post_SCH.m:6,11,17,73-79) -- and a MEX file that shadows one
 * would otherwise run silent.  After gsmcal_FCCH_coarse_position / _FCCH_fine_correction / _SCH_corr_rate_correction /
 * _carrier_correct_post_SCH this returns the lines that call's .m file would have printed, '\n'-separated, numbers formatted as
 * MATLAB's num2str formats them (gsmcal_num2str: the same formatter, for row vectors).  buf may be NULL; the return value is the
 * text's length without the terminator.  The gateway mexPrintf()s it; the Python mirror prints it when GSMCAL_VERBOSE=1. */
long gsmcal_last_call_report(gsmcal_ctx* ctx, char* buf, size_t cap);
long gsmcal_num2str(const double* x, int n, char* buf, size_t cap);
const char* gsmcal_version(void);
/* Pipelined batch calls (round 6).  gsm_sync_demod.m:107-124 is a serial chain per batch of dongles; a service that calibrates batch
 * after batch does not need batch i finished before batch i+1 starts.  With depth D > 1, up to D consecutive
 * gsmcal_calibrate_batch_dev calls that run on one lane (up to 127 streams) are in flight at once: call i runs on internal HIP
 * stream i mod D, in workspace i mod D, with the four-launch tail (whose kernels never wait for each other), and the kernels of the
 * calls in flight interleave on the GPU -- 64 streams x 1 020 000: 0.177 ms per call at depth 1, 0.148 at 3, 0.138 at 4 (the device
 * schedules four hardware queues: more gain nothing; with r_correct written calls in flight are supported and gain nothing).  The
 * same holds for single-stage gsmcal_fcch_scan_batch_dev calls (fewer than 1 200 captures; multi_rtl_sdr_gsm_FCCH_scanner.m:132-135,
 * 163-186 over consecutive sweeps): the detector of call i runs under the front kernel of call i+1 -- 200 captures 0.086 -> 0.068 ms.
 *   depth 1 (default): every call is complete in the context's stream order when it returns: the semantics of every earlier release.
 *   depth D = 2..8: the outputs of call i (table, pos_info, r_len) are complete in the context's stream order at the start of call
 *     i+D, or after gsmcal_sync(), or after ANY other entry point of this context (they all join the calls in flight first).  The
 *     raw bytes and output buffers of call i must stay untouched until then: give D consecutive calls distinct output buffers.
 *     Each call starts behind whatever the context's stream held when it was made (its input is ordered like at depth 1).
 *     gsmcal_allgather_table[_async] right behind a pipelined call is enqueued behind THAT call (the gathered table completes where
 *     the call's own outputs do).  gsmcal_last_batch_details describes the most recent call.
 * Results are identical at every depth (same kernels, same order per call).  Returns GSMCAL_E_ARG for depth < 1 or > 8.
 * (The staged forms of the first design -- a call cut into front end | fine search | fused tail on stage streams -- measured no gain
 * and are not in the library: profiles/experiments_r06/staged_pipeline_and_side_fused.patch, profiles/NOTES_r06.md.) */
int gsmcal_ctx_set_pipeline_depth(gsmcal_ctx* ctx, int depth);
int gsmcal_ctx_get_pipeline_depth(gsmcal_ctx* ctx);
/* How many internal streams the context has SEEN running side by side (= hardware queues its calls in flight are spread over; the
 * runtime hands out four): 0 before the first call in flight.  The context creates a dozen candidate streams at that call, keeps
 * those that pass a 200-us pairwise probe and maps slot s to kept stream s mod that number -- which streams share a hardware queue
 * depends on every stream the process created before (under PyTorch four streams created in a row landed on three queues). */
int gsmcal_ctx_pipeline_queues(gsmcal_ctx* ctx);
/* Diagnostics of the batch path's fused tail (one launch for everything behind the fine search's chunk sweep: its workgroups
 * exchange results inside the launch).  Several contexts may drive one GPU from several host threads; the library lets only one
 * such launch of the process be in flight per device and gives the later caller the four-launch tail (same results).
 * fused_launches: batch calls of this context that took the fused tail; gate_fallbacks: calls that would have but found
 * another context's fused tail unfinished (or ran inside the caller's own stream capture, where replays cannot be gated). */
int gsmcal_fused_tail_stats(gsmcal_ctx* ctx, unsigned long long* fused_launches, unsigned long long* gate_fallbacks);
/* The gate above is per process.  A fused tail stalled by ANOTHER process's fused tail on the same GPU gives up after
 * GSMCAL_FUSED_POLL_S seconds (default 3) and reports it through a pinned host word; gsmcal_sync() and the host-buffer entry
 * points then run the affected calls again with the four-launch tail (same inputs -- which the caller leaves untouched until it
 * has synchronised -- same outputs) and return 0: the time-out costs seconds, not the call.  A caller that synchronises its
 * stream without gsmcal_sync() sees GSMCAL_E_HIP in the status column of the affected rows instead.  Returns how often that
 * happened on this context (test hook: GSMCAL_TEST_FUSED_STALL=<stage 1..4> makes one workgroup never publish). */
long long gsmcal_fused_tail_reruns(gsmcal_ctx* ctx);
/* device memory helpers so a host program needs no HIP headers */
int gsmcal_dev_alloc(gsmcal_ctx* ctx, size_t bytes, void** dptr);
int gsmcal_dev_free(gsmcal_ctx* ctx, void* dptr);
int gsmcal_memcpy_h2d(gsmcal_ctx* ctx, void* dst, const void* src, size_t bytes);
int gsmcal_memcpy_d2h(gsmcal_ctx* ctx, void* dst, const void* src, size_t bytes);
/* Per-kernel timing with HIP events on the context's stream.  enable=1 brackets every kernel launch
 * with events (serialises nothing, adds two event records per launch).  Stats accumulate until reset. */
int gsmcal_profile_enable(gsmcal_ctx* ctx, int enable);
int gsmcal_profile_reset(gsmcal_ctx* ctx);
/* Restrict the bracketing to kernels whose name contains `substr` (NULL or "" = every kernel), so a
 * timed region can carry events for one kernel only. */
int gsmcal_profile_filter(gsmcal_ctx* ctx, const char* substr);
/* Returns the number of distinct kernels seen; fills up to `cap` entries (name pointers stay valid
 * for the life of the library). total_ms[i] is the summed duration, launches[i] the launch count. */
int gsmcal_profile_get(gsmcal_ctx* ctx, int cap, const char** names, double* total_ms, long* launches);

/* ---- the nine reference functions, MATLAB signatures (host pointers, synchronous) -------------- */

/* b = raw2iq(a)                                                     raw2iq.m:5-8
 * a: 2N x D doubles holding byte values (column-major); b: N x D complex. */
int gsmcal_raw2iq(gsmcal_ctx* ctx, const double* a, long rows_2n, int d, double* b);
/* same with the bytes as they come off the wire (fread(...,'uint8') before MATLAB widens them) */
int gsmcal_raw2iq_u8(gsmcal_ctx* ctx, const uint8_t* a, long rows_2n, int d, double* b);

/* r = chn_filter_8x_4x(s)                                           chn_filter_8x_4x.m:5-15
 * s: N x D complex; r: ceil(N/2) x D complex.  `num`/`ntaps`: the numerator the reference loads
 * from gsm_chn_filter_8x.mat (:9-10); pass NULL/0 to use the built-in 60 taps of
 * gsm_chn_filter_8x.fda. */
int gsmcal_chn_filter_8x_4x(gsmcal_ctx* ctx, const double* s, long n, int d,
                            const double* num, int ntaps, double* r);
/* r = chn_filter_4x(s)                                              chn_filter_4x.m:5-13
 * Single-rate channel filter at 4x oversampling: s, r: N x D complex.  `num`/`ntaps`: the numerator the
 * reference loads from gsm_chn_filter_4x.mat (:8-9); NULL/0 = the built-in 30 taps of gsm_chn_filter_4x.fda. */
int gsmcal_chn_filter_4x(gsmcal_ctx* ctx, const double* s, long n, int d, const double* num, int ntaps, double* r);
/* r = filter(coef, 1, s) column-wise (gsm_sync_demod.m:110, multi_rtl_sdr_gsm_FCCH_scanner.m:133);
 * keep every `decim`-th row starting with row 1 (decim = 1: all rows; the drivers' r(1:64:end,i)). */
int gsmcal_filter(gsmcal_ctx* ctx, const double* coef, int ntaps, const double* s, long n, int d,
                  int decim, double* r);

/* [hit_flag,hit_idx,hit_avg_snr,hit_snr] = move_fft_snr_runtime_avg(s,mv_len,fft_len,th)
 *                                                                    move_fft_snr_runtime_avg.m:5-50 */
int gsmcal_move_fft_snr_runtime_avg(gsmcal_ctx* ctx, const double* s, long len, int mv_len, int fft_len,
                                    double th, int* hit_flag, double* hit_idx, double* hit_avg_snr,
                                    double* hit_snr);
/* [hit_flag,hit_idx,hit_snr] = specific_fft_snr_fix_avg(s,target_set,fft_len,th,avg_snr)
 *                                                                    specific_fft_snr_fix_avg.m:5-34 */
int gsmcal_specific_fft_snr_fix_avg(gsmcal_ctx* ctx, const double* s, long len, const double target_set[2],
                                    int fft_len, double th, double avg_snr, int* hit_flag,
                                    double* hit_idx, double* hit_snr);
/* [position,snr] = FCCH_coarse_position(s,decimation_ratio)         FCCH_coarse_position.m:5-94
 * position/snr: capacity `cap` doubles; *count = number of hits (sentinel: *count = 1, both -1). */
int gsmcal_FCCH_coarse_position(gsmcal_ctx* ctx, const double* s, long len, int decimation_ratio,
                                double* position, double* snr, int cap, int* count);

/* [FCCH_pos,r,sampling_ppm,carrier_ppm] = FCCH_fine_correction(s,base_position,ov,carrier_freq)
 *                                                                    FCCH_fine_correction.m:5-197
 * r: capacity cap_r complex samples; *len_r = samples written (sentinel r = -1: *len_r = -1 and
 * nothing written).  r may be NULL (cap_r = 0) when only positions/ppm are wanted. */
int gsmcal_FCCH_fine_correction(gsmcal_ctx* ctx, const double* s, long len, const double* base_position,
                                int num_base, int oversampling_ratio, double carrier_freq,
                                double* fcch_pos, int cap_pos, int* num_pos,
                                double* r, long cap_r, long* len_r,
                                double* sampling_ppm, double* carrier_ppm);

/* [pos_info,r,sampling_ppm] = SCH_corr_rate_correction(s,FCCH_pos,sch_training_sequence,ov)
 *                                                                    SCH_corr_rate_correction.m:5-181
 * pos_info: cap_rows x 2 doubles, column-major with leading dimension cap_rows; *num_rows rows valid
 * (sentinel: *num_rows = 1, row = [-1 -1]). */
int gsmcal_SCH_corr_rate_correction(gsmcal_ctx* ctx, const double* s, long len, const double* fcch_pos,
                                    int num_fcch, const double* sch_training_sequence, int len_ts,
                                    int oversampling_ratio, double* pos_info, int cap_rows, int* num_rows,
                                    double* r, long cap_r, long* len_r, double* sampling_ppm);

/* [r,carrier_ppm] = carrier_correct_post_SCH(s,pos_info,ov,carrier_freq)
 *                                                                    carrier_correct_post_SCH.m:5-83
 * pos_info: rows x 2 column-major, leading dimension `ld`. */
int gsmcal_carrier_correct_post_SCH(gsmcal_ctx* ctx, const double* s, long len, const double* pos_info,
                                    int rows, int ld, int oversampling_ratio, double carrier_freq,
                                    double* r, long cap_r, long* len_r, double* carrier_ppm);

/* Front end of SCH_demod(s,pos_info,training_sequence,ov)                SCH_demod.m:53-59,79-90
 * Per SCH row of pos_info (type 1): the burst with 8 symbols either side and the traceback depth behind it
 * (len_fde_ov = 194*ov samples), its frequency-domain channel estimate against the training sequence and the equalised
 * burst x_eq = ifft(fft(x) ./ (fft(x_training) ./ fft(training))).  The Viterbi GMSK demodulator that follows in the
 * reference (Communications Toolbox, output discarded) is not part of this library.
 * x_eq: cap_bursts x len_fde_ov complex, burst-major; *num_bursts SCH bursts written, *len_fde_ov their length.
 * pos_info all -1 (:8-11): returns GSMCAL_S_POST_NO_POS with *num_bursts = 0. */
int gsmcal_SCH_equalise(gsmcal_ctx* ctx, const double* s, long len, const double* pos_info, int rows, int ld,
                        const double* sch_training_sequence, int len_ts, int oversampling_ratio,
                        double* x_eq, int cap_bursts, int* num_bursts, int* len_fde_ov);

/* ppm_out = total_ppm_calculation(ppm_in)                           total_ppm_calculation.m:5-21
 * (pure host arithmetic; no context needed) */
int gsmcal_total_ppm_calculation(const double* ppm_in, int n, double* ppm_out);

/* ---- batched hot path (what the drivers' loops become) ----------------------------------------- */

/* Front end of both drivers for D captures at once: raw2iq + filter(coef,1,.) + r(1:decim:end)
 * (gsm_sync_demod.m:107,110,117; multi_rtl_sdr_gsm_FCCH_scanner.m:132-135).
 * raw: D x 2N bytes (capture-major: capture d starts at raw + d*2N). out: D x ceil(N/decim) complex. */
int gsmcal_frontend_batch(gsmcal_ctx* ctx, const uint8_t* raw, int d, long n, const double* coef,
                          int ntaps, int decim, double* out);
int gsmcal_frontend_batch_dev(gsmcal_ctx* ctx, const uint8_t* d_raw, int d, long n, const double* coef,
                              int ntaps, int decim, double* d_out);

/* Scanner detect loop for D captures (multi_rtl_sdr_gsm_FCCH_scanner.m:132-135 front end,
 * :164 FCCH_coarse_position, :168-185 acceptance).  Outputs per capture: snr, num_hit (as the
 * driver's arrays), optional positions/snrs [D][GSMCAL_MAX_HITS] and counts [D] (NULL to skip). */
int gsmcal_fcch_scan_batch(gsmcal_ctx* ctx, const uint8_t* raw, int d, long n, const double* coef,
                           int ntaps, double* snr, double* num_hit, double* positions,
                           double* pos_snr, int* counts);
int gsmcal_fcch_scan_batch_dev(gsmcal_ctx* ctx, const uint8_t* d_raw, int d, long n, const double* coef,
                               int ntaps, double* d_snr_numhit /* [D][2] */, double* d_positions,
                               double* d_pos_snr, int* d_counts);

/* Per-dongle body of gsm_sync_demod.m:107-124 for D streams at once, from raw bytes to the
 * calibration table.  carrier_freq: [D].  table: [D][GSMCAL_TABLE_COLS].
 * Optional outputs (NULL to skip): pos_info [D][2][GSMCAL_MAX_POS_ROWS] (per stream column-major,
 * ld = GSMCAL_MAX_POS_ROWS); r_correct [D][N] complex + r_len [D] (the corrected stream the
 * reference hands to SCH_demod; r_len = -1 where the reference returns r = -1). */
int gsmcal_calibrate_batch(gsmcal_ctx* ctx, const uint8_t* raw, int d, long n, const double* coef,
                           int ntaps, const double* sch_training_sequence, int len_ts,
                           const double* carrier_freq, double* table, double* pos_info,
                           double* r_correct, long* r_len);
/* The d_* outputs of the *_dev entry points may be any memory the GPU can store to, in particular pinned host memory
 * (hipHostMalloc): the kernels that finish a batch then write the table / (snr, num_hit) rows where the host reads them
 * after synchronising the stream, and no device-to-host copy has to be queued behind the batch (bench.py does this). */
int gsmcal_calibrate_batch_dev(gsmcal_ctx* ctx, const uint8_t* d_raw, int d, long n, const double* coef,
                               int ntaps, const double* sch_training_sequence, int len_ts,
                               const double* carrier_freq, double* d_table, double* d_pos_info,
                               double* d_r_correct, long* d_r_len);

/* ---- multi-GPU: one process per GPU, ONE collective ---------------------------------------------------------------
 * The reference has no distributed layer; units (dongle streams, ARFCN captures) are independent (gsm_sync_demod.m:112,
 * multi_rtl_sdr_gsm_FCCH_scanner.m:60-65,163), so ranks shard them block-contiguously and exchange only the result
 * table every rank needs for the inter-dongle comparison (gsm_sync_demod.m:151-158).  RCCL (librccl.so, loaded on first
 * use) over xGMI; the all-gather is enqueued on the context's stream right behind the kernels that fill the table.
 * Bootstrap: rank 0 obtains a 128-byte id and hands it to the other ranks by any means (gsmcal_comm_init_rank), or all
 * ranks name the same file on a shared filesystem (gsmcal_comm_init_file: rank 0 writes the id, the others wait for it). */
typedef struct gsmcal_comm gsmcal_comm;
#define GSMCAL_COMM_ID_BYTES 128
int gsmcal_comm_get_unique_id(void* id_out /* GSMCAL_COMM_ID_BYTES */);
int gsmcal_comm_init_rank(gsmcal_ctx* ctx, const void* id, int world, int rank, gsmcal_comm** out);
int gsmcal_comm_init_file(gsmcal_ctx* ctx, const char* path, int world, int rank, gsmcal_comm** out);
/* The id file is run-specific: it carries a magic word and a 64-bit nonce next to the id.  Rank 0 removes whatever sits at
 * `path` before it generates the id, publishes the new file by an atomic rename, and removes it again once the communicator
 * is up (ncclCommInitRank returns only after every rank has joined), so a file survives only a crashed bootstrap.  Readers
 * accept nothing but a complete file with the right magic AND the caller's nonce; with nonce 0 ("none") they instead
 * refuse files older than GSMCAL_COMM_STALE_S seconds (default 120) -- unsafe for a relaunch inside that window, which
 * would accept the dead launch's id.  Give every launch its own nonce (job id, launcher pid, start time).
 * gsmcal_comm_init_file derives one itself: GSMCAL_COMM_NONCE if set, else a hash of the launcher's run / job id
 * (TORCHELASTIC_RUN_ID unless it is torchrun's literal default "none", TORCHELASTIC_RESTART_COUNT, SLURM_JOB_ID,
 * SLURM_STEP_ID, PBS_JOBID, LSB_JOBID) and MASTER_ADDR:MASTER_PORT; 0 only if none of these exist.  A nonce DERIVED this way
 * may repeat from launch to launch (plain `torchrun`: RUN_ID "none", 127.0.0.1:29500 every time), so gsmcal_comm_init_file
 * keeps the age test on for it: a record is accepted only with the right nonce AND younger than the stale window.  Only a
 * nonce the caller chose (GSMCAL_COMM_NONCE, gsmcal_comm_init_file_nonce with nonce != 0) switches the age test off. */
int gsmcal_comm_init_file_nonce(gsmcal_ctx* ctx, const char* path, unsigned long long nonce, int world, int rank,
                                gsmcal_comm** out);
/* The nonce gsmcal_comm_init_file derives from the environment (see above); 0 = the environment identifies no launch. */
unsigned long long gsmcal_comm_default_nonce(void);
/* The file protocol alone (no GPU, no RCCL; what the two functions above run before ncclCommInitRank): rank 0 publishes
 * id_inout, the other ranks wait up to timeout_s seconds for it and receive it in id_inout.  GSMCAL_E_ARG on time-out.
 * gsmcal_comm_id_file_remove: rank 0's clean-up after the communicator is up. */
int gsmcal_comm_id_file_exchange(const char* path, unsigned long long nonce, int world, int rank,
                                 void* id_inout /* GSMCAL_COMM_ID_BYTES */, double timeout_s);
/* ... with the age test on whatever the nonce (what gsmcal_comm_init_file runs for a nonce derived from the environment). */
int gsmcal_comm_id_file_exchange_aged(const char* path, unsigned long long nonce, int world, int rank,
                                      void* id_inout /* GSMCAL_COMM_ID_BYTES */, double timeout_s);
int gsmcal_comm_id_file_remove(const char* path);
void gsmcal_comm_destroy(gsmcal_comm* comm);
/* d_all[r][i][c] = rank r's d_local[i][c]: rows_per_rank x cols doubles per rank (ranks with fewer units pad their block,
 * e.g. with NaN); device pointers; enqueued on the context's stream (gsmcal_sync() before reading d_all on the host). */
int gsmcal_allgather_table(gsmcal_ctx* ctx, gsmcal_comm* comm, const double* d_local, int rows_per_rank, int cols,
                           double* d_all);

/* The same collective off the critical path of the caller's stream: RCCL runs on a side stream of the context, behind an event
 * recorded on the context's stream at the time of the call; the context's stream does NOT wait for it, so the kernels of the
 * next batch start at once and the gather of batch i travels under the kernels of batch i+1 (gsm_sync_demod.m:151-158 only
 * needs the gathered table after the per-dongle loop).  `slot` (0..3) names the buffer pair in flight:
 *   gsmcal_allgather_wait(ctx, slot)  the context's stream waits -- on the GPU -- for that slot's collective: call it before
 *                                     enqueueing whatever overwrites d_local or reads d_all of the slot (no-op for a slot never posted);
 *   gsmcal_allgather_sync(ctx, slot)  blocks the host until that slot's collective has finished. */
int gsmcal_allgather_table_async(gsmcal_ctx* ctx, gsmcal_comm* comm, const double* d_local, int rows_per_rank, int cols,
                                 double* d_all, int slot);
int gsmcal_allgather_wait(gsmcal_ctx* ctx, int slot);
int gsmcal_allgather_sync(gsmcal_ctx* ctx, int slot);

/* ---- ingest ring: rtl_tcp bytes -> pinned host ring -> asynchronous H2D ----------------------------------
 * The reference reads every capture with fread(tcp_obj, 2*num_sample, 'uint8') into MATLAB memory and processes it in
 * place (gsm_sync_demod.m:94-104, multi_rtl_sdr_gsm_FCCH_scanner.m:117-131).  Here the socket reader writes straight into
 * a pinned host slot, the slot goes to its device twin on a copy stream, and the copy of batch k+1 runs under the kernels
 * of batch k.  A ring has `slots` (>= 2) slots of `batch_bytes`; per slot the cycle is
 *     host = gsmcal_ring_host(ring, s)          the producer fills it (recv() target, zero copy)
 *     gsmcal_ring_submit(ring, s)               async H2D on the ring's copy stream (waits, on the GPU, for the
 *                                               consumer's previous use of the slot)
 *     dev = gsmcal_ring_acquire(ring, s)        the context's stream waits (on the GPU) for that copy; returns the
 *                                               device buffer to hand to a *_dev entry point
 *     gsmcal_ring_release(ring, s)              after enqueueing the consumer: the slot may be overwritten once it is done
 *     gsmcal_ring_host_ready(ring, s)           host-side wait until the slot's H2D has finished, i.e. until the pinned
 *                                               host buffer may be refilled
 * No call blocks the host except gsmcal_ring_host_ready. */
typedef struct gsmcal_ring gsmcal_ring;
int gsmcal_ring_create(gsmcal_ctx* ctx, size_t batch_bytes, int slots, gsmcal_ring** out);
void gsmcal_ring_destroy(gsmcal_ring* ring);
void* gsmcal_ring_host(gsmcal_ring* ring, int slot);
int gsmcal_ring_submit(gsmcal_ring* ring, int slot, size_t bytes);
void* gsmcal_ring_acquire(gsmcal_ring* ring, int slot);
int gsmcal_ring_release(gsmcal_ring* ring, int slot);
int gsmcal_ring_host_ready(gsmcal_ring* ring, int slot);

/* Synthetic-input utility for benchmarks and tests (NOT part of the reference's path; SURVEY 8d: the 131 GB scanner
 * configuration is generated on the device).  Expands k seeded base captures (d_base: [k][2n] bytes) into d distinct
 * captures d_out: [d][2n]: capture first_unit+j = base[(first_unit+j) mod k] rotated by a per-capture number of
 * samples, every byte dithered by -1/0/+1 from a counter-based hash of (seed, unit, sample), clipped to 0..255.
 * multi-rtl-sdr-calibration_amd/synth.py:expand_capture() is the bit-identical host twin. */
int gsmcal_synth_expand_dev(gsmcal_ctx* ctx, const uint8_t* d_base, int k, long n, uint8_t* d_out, long d,
                            long first_unit, unsigned long long seed);

/* Debug/parity taps into the last calibrate/scan batch: copies per-stream intermediates to host.
 * coarse_pos/coarse_snr/fine_first/fcch_pos/sch_first: [D][GSMCAL_MAX_HITS]; counts: [D][5]
 * (n_coarse, n_fine_first, n_fcch, n_sch_first, n_pos_rows).  Any pointer may be NULL. */
int gsmcal_last_batch_details(gsmcal_ctx* ctx, int d, double* coarse_pos, double* coarse_snr,
                              double* fine_first, double* fcch_pos, double* sch_first, int* counts);

/* Parity tap: the per-window SNR table (move_fft_snr_runtime_avg.m:18-27, 1-based window i at index i-1) the coarse detector of
 * the last batch call built for `stream`.  *n_moving entries are the moving search's windows (every one computed in full);
 * entries beyond them (latency path only, up to *n_table) serve the hop walk and hold -inf where a window was proved to be
 * below the screening level.  The decisions snr - avg > th are taken on these values: the parity suite measures how far they
 * sit from the oracle's (the implementation noise the certified scan's 1e-6 dB margin has to cover).
 * Batches of more than 128 streams per internal lane keep no table (the scan kernel computes the values in LDS):
 * GSMCAL_E_UNSUPPORTED unless the context was created under GSMCAL_SNR_INLINE_KEEP=1. */
int gsmcal_last_batch_snr(gsmcal_ctx* ctx, int stream, double* snr, long cap, long* n_table, long* n_moving);

#ifdef __cplusplus
}
#endif
#endif /* GSMCAL_H */
