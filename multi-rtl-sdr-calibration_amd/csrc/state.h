// state.h -- per-stream device state shared by every kernel of the calibration chain.
//
// One StreamState per stream (dongle capture) lives in HBM.  The chain is a fixed sequence of
// kernel launches; every data-dependent decision of the reference (how many hits, where the windows
// are, which sentinel to return) is taken on the device by the `decide` kernels and stored here, so
// the host never synchronises in the middle of the chain.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gsmcal.h"

// Development build only (hipcc -DGSMCAL_DEVTIMING): per-phase timestamps inside the kernels.  The product library is
// compiled without it and contains none of this.
#ifdef GSMCAL_DEVTIMING
__device__ unsigned long long* g_stamps = nullptr;   // [kernel id][DEV_STAMP_BLOCKS][16], 100 MHz wall clock
#define DEV_STAMP_BLOCKS 1024
// -DGSMCAL_DEVTIMING_LIGHT=<kernel id>: only the stamps of that kernel in GSMCAL_DEVTIMING_MASK (default 0, 9, 10) are compiled in (when do its workgroups start
// and reach their last stage?) -- the full set changes the register allocation of the kernel it measures.
#ifdef GSMCAL_DEVTIMING_LIGHT
#ifndef GSMCAL_DEVTIMING_MASK
#define GSMCAL_DEVTIMING_MASK 0x601          /* stamps 0, 9, 10 */
#endif
#define DEV_STAMP_ON(kid, i) ((kid) == GSMCAL_DEVTIMING_LIGHT && ((GSMCAL_DEVTIMING_MASK >> (i)) & 1))
#else
#define DEV_STAMP_ON(kid, i) true
#endif
#define DEV_STAMP(kid, blk, i)                                                                          \
    do {                                                                                                \
        if (DEV_STAMP_ON(kid, i) && g_stamps && threadIdx.x == 0 && (blk) < DEV_STAMP_BLOCKS)          \
            g_stamps[((size_t)(kid) * DEV_STAMP_BLOCKS + (blk)) * 16 + (i)] = wall_clock64();          \
    } while (0)
#else
#define DEV_STAMP(kid, blk, i) do { } while (0)
#endif
enum { KID_COARSE_SNR = 0, KID_COARSE_SCAN, KID_GATHER, KID_CERT, KID_CHUNK, KID_VERIFY, KID_BT1, KID_SCH, KID_BT0, KID_FRONT, KID_N };

#define MAXH GSMCAL_MAX_HITS
#define MAXROWS GSMCAL_MAX_POS_ROWS

// A stream is evaluated lazily through a chain of up to 4 single-operation levels on top of level 0.
//   level 0 : the filtered stream  f[i] = sum_k coef[k]*(raw[i-k]-mean)   (SRC_RAW)
//             or a complex-double array handed in through the API          (SRC_ARR)
//   OP_LERP : L[k] = lerp(L_prev, k*f)        interp1(...,'linear') of FCCH_fine_correction.m:123-125
//                                             and SCH_corr_rate_correction.m:126-127
//   OP_MIX  : L[k] = L_prev[k]*exp(1i*k*phi)  FCCH_fine_correction.m:165, carrier_correct_post_SCH.m:83
//   OP_COPY : L[k] = L_prev[k]                (SCH resample skipped when e == 0, :120)
enum { OP_NONE = 0, OP_LERP = 1, OP_MIX = 2, OP_COPY = 3 };
enum { SRC_ARR = 0, SRC_RAW = 1 };
#define NLEVELS 5  // level 0 + 4 ops

struct LevelOp {
    int type;      // OP_*
    int pad;
    double param;  // OP_LERP: (1+e); OP_MIX: comp_phase_rotate
    long n;        // length of this level's stream
};

struct alignas(16) StreamState {
    // ---- level chain ----
    long n0;                 // length of level 0
    unsigned long long sum_i, sum_q;  // integer byte sums (raw sources)
    double mean_re, mean_im;
    LevelOp op[NLEVELS];     // op[0] unused (level 0), op[1..4]
    // ---- coarse (FCCH_coarse_position) ----
    int n_coarse;            // number of coarse hits (0 = none found)
    int coarse_hit_flag;     // move_fft hit flag
    double coarse_pos[MAXH]; // 1x-symbol units, 1-based (after (p-1)*dec+1)
    double coarse_snr[MAXH];
    double hit_avg_snr;
    double mv_hit_idx, mv_hit_snr;   // raw outputs of move_fft_snr_runtime_avg
    // ---- generic window list consumed by k_gather / k_slide_dft / k_tone / k_sch_corr ----
    int n_win;
    long win_start[MAXH];    // 0-based start index at the window's level
    // ---- fine search ----
    int n_fine_ws;           // windows of the fine search and where they start at level 0: their filtered samples stay in the
    long fine_ws[MAXH];      // lane's window buffer, and later per-burst gathers inside one of them read level 0 from there
    int n_fine;              // last_idx
    double fine_first[MAXH]; // first-round FCCH_pos (8x units, 1-based)
    int n_fcch;              // length(FCCH_pos) returned
    int fcch_is_sentinel;    // FCCH_pos == -1 (scalar sentinel)
    double fcch_pos[MAXH];
    double sampling_ppm1, carrier_ppm1;
    double fo_burst[MAXH], snr_burst[MAXH];
    int prior_bin[MAXH];     // fine search's winning bin per hit: where the burst spectrum's peak is expected
    int r1_kind;             // what FCCH_fine_correction returns as r: 0 = -1, 1 = s, 2 = lerp only, 3 = lerp+mix
    // ---- SCH ----
    int n_sch_first;
    int sch_edge;            // an SCH correlation peak sat on the edge of its search window (:59)
    double sch_first[MAXH];
    int n_sch;
    double sch_pos[MAXH];
    double sampling_ppm2;
    int r2_kind;             // r of SCH_corr_rate_correction: 0 = -1, 1 = s, 2 = resampled (or s when e==0)
    int n_rows;              // rows of pos_info (0 => sentinel)
    int n_sent_rows;         // rows of the all -1 sentinel when n_rows == 0: 1 for pos_info = [-1 -1] (:9,:61), 3*num_fcch_hit
                             // for the -ones(3K,2) pre-allocation returned by the :84 and :106-112 exits (:32)
    double pos_info[2 * MAXROWS];   // column-major, ld = MAXROWS
    // ---- post SCH ----
    double carrier_ppm2;
    int r3_kind;             // r of carrier_correct_post_SCH: 0 = -1, 3 = mixed
    // ---- bookkeeping ----
    int status;              // first non-zero status met
    int stage_status[4];     // per reference function: fine, sch, post, coarse
};

// the tunable thresholds of gsmcal_params as the kernels see them (by value inside StepArgs / CoarseArgs)
struct DevParams {
    double coarse_th, fine_max_ppm, fine_gate_snr, sch_max_ppm, scan_spacing, scan_spacing_idle, scan_tol;
    int min_hits, post_min_bcch, scan_min_hits, pad;
};

struct PeakOut {  // partial result of k_slide_dft for one (window, bin-block)
    double p;     // best |X|^2
    int tie;      // tie-break key: shift index m (fine search) or fftshift-ed bin index (spectrum)
    int k;        // bin index (unshifted)
};

__host__ __device__ inline void set_status(StreamState* s, int stage, int code) {
    if (s->stage_status[stage] == 0) s->stage_status[stage] = code;
    if (s->status == 0) s->status = code;
}
