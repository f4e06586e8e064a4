// kernels_detect.h -- FCCH coarse detector and the fine (sliding-DFT) peak search.
//
//   k_coarse_snr / k_coarse_scan
//                 FCCH_coarse_position.m:5-94 with move_fft_snr_runtime_avg.m:5-50 and
//                 specific_fft_snr_fix_avg.m:5-34 inside: all sliding-window SNRs in parallel, then one
//                 workgroup per stream replays the reference's serial moving-average recurrence (bit for
//                 bit the same update order) and walks the hops, all on the device.
//   k_fine_cert / k_fine_chunk / k_fine_verify (body)
//                 FCCH_fine_correction.m:48-52: max over 1025 window starts of max_k |FFT_1184|^2 --
//                 exact certificate on the tone's bins, packed-fp32 sweep of the chunks it leaves open,
//                 exact fp64 on what survives (see the comment block above FS_CHUNK).
//   k_fft_burst   1184-point spectra (37 x 32 Cooley-Tukey in LDS); fft37_* are its building blocks, shared
//                 with k_fine_chunk and k_burst_tone.
//   k_fine_search the plain all-bin fp64 sliding DFT (GSMCAL_PRESCREEN=0): cross-check of the scheme above.
#pragma once
#include "state.h"
#include "kernels_frontend.h"

// ---- per-window SNR (move_fft_snr_runtime_avg.m:18-27) ------------------------------------------
// 16-point FFT, radix-2 decimation in frequency, in place and fully unrolled in registers (fp64); the trivial
// twiddles (1, -i) cost no multiplies.  The output is in BIT-REVERSED order: X[k] ends up in x[fft16_rev(k)].
__device__ __forceinline__ constexpr int fft16_rev(int k) { return ((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | ((k & 8) >> 3); }
__device__ __forceinline__ void fft16(cplx* x) {
    const double c1 = 0.92387953251128673848, s1 = 0.38268343236508978178;  // cos/sin(pi/8)
    const double r2 = 0.70710678118654752440;
    const cplx w16[8] = {{1.0, 0.0}, {c1, -s1}, {r2, -r2}, {s1, -c1}, {0.0, -1.0}, {-s1, -c1}, {-r2, -r2}, {-c1, -s1}};
#pragma unroll
    for (int len = 16; len >= 2; len >>= 1) {
        const int half = len >> 1, step = 16 / len;
#pragma unroll
        for (int i = 0; i < 16; i += len) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const cplx u = x[i + j], v = x[i + j + half];
                x[i + j] = make_double2(u.x + v.x, u.y + v.y);
                const cplx d = make_double2(u.x - v.x, u.y - v.y);
                const int t = j * step;
                if (t == 0) x[i + j + half] = d;
                else if (t == 4) x[i + j + half] = make_double2(d.y, -d.x);      // * (-i)
                else x[i + j + half] = cmul(d, w16[t]);
            }
        }
    }
}

// SNR from the power spectrum P[0..L) (move_fft_snr_runtime_avg.m:22-27): first max, 3 circular bins
// around it vs the rest, in dB.  L is a compile-time constant so P stays in registers.
// floor2 >= 0 (batch path, DecView::floor2): the DC term was removed by linearity, which leaves up to ~2^-46 |mean * sum(coef)|
// of rounding residue on every sample.  A window whose whole power is below that residue is EXACTLY zero in the reference's
// arithmetic (it subtracts the mean from the integer samples first: raw2iq.m:8), where 0/0 makes the SNR NaN
// (move_fft_snr_runtime_avg.m:26-27) -- a constant or zero-filled stretch of a capture; NaN is returned for it here too
// instead of the ratio of two residues.  floor2 < 0: never (plain arrays, exact operation order).
template <int L>
__device__ __forceinline__ double snr_from_power(const double (&P)[L], double floor2 = -1.0) {
    // first max (strict >) and its two circular neighbours, carried along in one pass
    double pm = P[L - 1], pc = P[0], pp = P[1 % L];
#pragma unroll
    for (int k = 1; k < L; ++k) {
        const bool h = P[k] > pc;
        pm = h ? P[k - 1] : pm;
        pp = h ? P[(k + 1) % L] : pp;
        pc = h ? P[k] : pc;
    }
    double sig = pm + pc;
    sig = sig + pp;
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < L; ++k) tot += P[k];
    const double noise = tot - sig;
    if (tot <= floor2) return __longlong_as_double(0x7ff8000000000000LL);
    return 10.0 * log10(sig / noise);
}

// The decimated detector input.  In the batch path the fused front end (k_front_fused) stores the FIR
// of the RAW bytes, y'[j] = sum_k coef[k]*raw[64j-k]; the DC term is removed here on load by linearity:
// y[j] = y'[j] - mean * sum_{valid k} coef[k]  (all taps for j >= 1, tap 0 only for j = 0).
// With corr == 0 the view is a plain array (API paths, exact reference operation order).
struct DecView {
    const cplx* s;
    double cr, ci;      // mean * sum(coef)        (subtracted from every sample j >= n_head)
    double mr, mi;      // the mean itself, for the head rows
    const double* head; // head[j] = sum_{k <= min(ntaps-1, decim*j)} coef[k], j < n_head: the rows whose filter()
    int n_head;         // window still overlaps the zero initial state (row 0 only when ntaps <= decim + 1)
    double floor2;      // power below which a 16-point window is rounding residue of the DC removal (snr_from_power); -1: none
};
// DC removal of a sample that was loaded raw
__device__ __forceinline__ cplx dv_fix(const DecView& v, cplx x, long j) {
    if (j < v.n_head) { const double h = v.head[j]; return make_double2(x.x - v.mr * h, x.y - v.mi * h); }
    return make_double2(x.x - v.cr, x.y - v.ci);
}
__device__ __forceinline__ cplx dv_load(const DecView& v, long j) {
    const cplx x = v.s[j];
    if (j < v.n_head) { const double h = v.head[j]; return make_double2(x.x - v.mr * h, x.y - v.mi * h); }
    return make_double2(x.x - v.cr, x.y - v.ci);
}

// 16 samples from `start`.  The DC term of the batch path (DecView) is a constant over the window, and the DFT of a
// constant lives in bin 0 alone, so it is removed there (X[0] -= 16 c) instead of from every sample.  NOT valid for
// the few windows that still overlap filter()'s zero initial state (start < v.n_head): see window_snr_head.
// x: the 16 raw samples of a window wholly past the head rows (anywhere: global memory or LDS)
__device__ __forceinline__ double window_snr16_from(const DecView& v, const cplx* xin) {
    cplx x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = xin[i];
    fft16(x);
    x[0].x -= 16.0 * v.cr; x[0].y -= 16.0 * v.ci;
    double P[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { const cplx X = x[fft16_rev(k)]; P[k] = X.x * X.x + X.y * X.y; }   // abs(fft(.)).^2
    return snr_from_power<16>(P, v.floor2);
}
__device__ __forceinline__ double window_snr16(const DecView& v, long start) {
    cplx x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = v.s[start + i];
    fft16(x);
    x[0].x -= 16.0 * v.cr; x[0].y -= 16.0 * v.ci;
    double P[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { const cplx X = x[fft16_rev(k)]; P[k] = X.x * X.x + X.y * X.y; }   // abs(fft(.)).^2
    return snr_from_power<16>(P, v.floor2);
}

// generic length (2..64) via direct DFT with tw[m] = exp(-2*pi*i*m/L); nothing is stored: the bins
// around the first max are recomputed (bit-identical) after the scan.
__device__ __forceinline__ double dft_bin_power(const DecView& v, long start, int fft_len, const cplx* tw, int k) {
    double ar = 0.0, ai = 0.0;
    int idx = 0;
    for (int n = 0; n < fft_len; ++n) {
        const cplx w = tw[idx];
        const cplx x = dv_load(v, start + n);
        ar += x.x * w.x - x.y * w.y;
        ai += x.x * w.y + x.y * w.x;
        idx += k;
        if (idx >= fft_len) idx -= fft_len;
    }
    const double m = hypot(ar, ai);
    return m * m;
}

__device__ __noinline__ double window_snr_generic(const DecView& s, long start, int fft_len, const cplx* tw) {
    int mi = 0;
    double mx = -1.0, tot = 0.0;
    for (int k = 0; k < fft_len; ++k) {
        const double p = dft_bin_power(s, start, fft_len, tw, k);
        tot += p;
        if (p > mx) { mx = p; mi = k; }             // first max
    }
    const int km = mi == 0 ? fft_len - 1 : mi - 1, kp = mi == fft_len - 1 ? 0 : mi + 1;
    double sig = dft_bin_power(s, start, fft_len, tw, km) + mx;
    sig = sig + dft_bin_power(s, start, fft_len, tw, kp);
    const double noise = tot - sig;
    return 10.0 * log10(sig / noise);
}

// The windows that overlap filter()'s zero initial state (start < n_head; one per stream for the drivers' filters):
// sample-by-sample DC removal; hx holds the corrected samples of the window.
__device__ __forceinline__ double window_snr16_head(const cplx* hx, double floor2) {
    cplx x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = hx[i];
    fft16(x);
    double P[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { const cplx X = x[fft16_rev(k)]; P[k] = X.x * X.x + X.y * X.y; }
    return snr_from_power<16>(P, floor2);
}

__device__ __forceinline__ double window_snr(const DecView& s, long start, int fft_len, const cplx* tw) {
    return fft_len == 16 ? window_snr16(s, start) : window_snr_generic(s, start, fft_len, tw);
}

// scanner path: where k_coarse_scan puts the acceptance rule's outputs (snr_numhit == nullptr: no acceptance step)
struct ScanAccept {
    double* snr_numhit; double* positions; double* pos_snr; int* counts;
};
struct CoarseArgs {
    const cplx* s; long s_stride; long len;   // decimated streams
    int decimation_ratio;                     // FCCH_coarse_position's 2nd argument
    int mode;      // 0 = FCCH_coarse_position, 1 = move_fft_snr_runtime_avg only, 2 = specific only
    int mv_len, fft_len; double th;           // modes 1/2 (mode 0 derives them like the reference)
    long t_lo, t_hi; double avg_snr;          // mode 2: target_set and fixed average
    double* snr_g; long snr_stride;           // per-window SNRs of the moving search (k_coarse_snr -> k_coarse_scan)
    long snr_nwin;                            // windows per stream in that table: 0 = those of the moving search only (the first
                                              // 23 frames); len-fft_len+1 = every window of the stream, and the hop walk reads it
    ScanAccept accept; DevParams P;           // scanner path: multi_rtl_sdr_gsm_FCCH_scanner.m:168-185 at the end of k_coarse_scan
    double snr_screen_db;                     // ... where entries past the moving search may read -inf: "proven below this level"
    int snr_tile;                             // ... and k_coarse_snr's workgroup j handles later windows [j, j+1) * snr_tile (<= CS_TILE, multiple of 4)
    int fine_setup_ov;                        // > 0: run FCCH_fine_correction's window setup at the end (batch path)
    double csum_all;                          // mean_corr: the input is the FIR of the raw bytes; remove mean*csum on load
    const double* csum_head; int n_head;      //            partial tap sums of the rows that overlap filter()'s zero initial state
    int mean_corr;
    double th0; int min_hits;                 // mode 0: gsmcal_params.coarse_th_db (FCCH_coarse_position.m:21), min_hits for the fine setup
    const unsigned long long* partial;        // mean_corr: per-block byte sums of k_front_fused, [S][npartial][2]
    int npartial; long n0;                    //            and the capture length they divide by
    double snr_gx2;                           // the screening level of k_coarse_snr as g(rho_X)^2 (see there)
    int g_fft_len; long g_n_first;            // mode 0: FCCH_coarse_position.m:15-25 worked out by the host (0: derive them here)
};

// raw2iq.m:8 from the front kernel's per-block partial sums: exact integer totals, one fp64 divide each
// (block-cooperative: wave 0 adds the partials, the totals are shared through LDS; contains a barrier)
__device__ __forceinline__ void stream_mean(const CoarseArgs& a, int stream, double* mr, double* mi,
                                            unsigned long long* tot_i, unsigned long long* tot_q) {
    __shared__ unsigned long long sh_tot[2];
    if (threadIdx.x < 64) {
        unsigned long long si = 0, sq = 0;
        const unsigned long long* p = a.partial + (size_t)stream * a.npartial * 2;
        for (int b = threadIdx.x; b < a.npartial; b += 64) { si += p[2 * b]; sq += p[2 * b + 1]; }
        for (int off = 32; off > 0; off >>= 1) {
            si += __shfl_down(si, off, 64);
            sq += __shfl_down(sq, off, 64);
        }
        if (threadIdx.x == 0) { sh_tot[0] = si; sh_tot[1] = sq; }
    }
    __syncthreads();
    *tot_i = sh_tot[0]; *tot_q = sh_tot[1];
    *mr = (double)sh_tot[0] / (double)a.n0;
    *mi = (double)sh_tot[1] / (double)a.n0;
}

__device__ __forceinline__ DecView dec_view(const CoarseArgs& a, int stream, double mean_re, double mean_im) {
    DecView v;
    v.s = a.s + (size_t)stream * a.s_stride;
    const double m = a.mean_corr ? 1.0 : 0.0;
    v.cr = m * mean_re * a.csum_all;   v.ci = m * mean_im * a.csum_all;
    v.mr = mean_re; v.mi = mean_im;
    v.head = a.csum_head;
    v.n_head = a.mean_corr ? a.n_head : 0;
    // (16 |c| 2^-46)^2: sixteen samples, each within 2^-46 |c| of the value the reference's order of operations gives
    v.floor2 = a.mean_corr ? 256.0 * (v.cr * v.cr + v.ci * v.ci) * 2.019483917365790e-28 : -1.0;
    return v;
}

struct CoarseGeom { int fft_len, mv_len; double th; long n_first, nwin; };

template <bool HOST_VALUES = true>
__device__ __forceinline__ CoarseGeom coarse_geom(const CoarseArgs& a) {
    CoarseGeom g;
    if (a.mode == 0) {
        // FCCH_coarse_position.m:15-25
        // (the host's values when it sent them: a log2, a ceil and two divides per thread are a tenth of k_coarse_snr's instructions)
        g.fft_len = HOST_VALUES && a.g_fft_len > 0 ? a.g_fft_len : 1 << (int)floor(log2(148.0 / (double)a.decimation_ratio));
        g.th = a.th0;
        g.mv_len = 10 * g.fft_len;
        g.n_first = HOST_VALUES && a.g_fft_len > 0 ? a.g_n_first : (long)ceil(23.0 * 1250.0 / (double)a.decimation_ratio);
    } else {
        g.fft_len = a.fft_len; g.mv_len = a.mv_len; g.th = a.th; g.n_first = a.len;
    }
    g.nwin = g.n_first - (g.fft_len - 1);
    return g;
}

__device__ __forceinline__ void coarse_twiddles(cplx* tw, int fft_len, int tid, int nthreads) {
    if (fft_len != 16 && fft_len >= 2 && fft_len <= 64)
        for (int i = tid; i < fft_len; i += nthreads) {
            double sn, cs;
            sincospi(-2.0 * (double)i / (double)fft_len, &sn, &cs);
            tw[i] = make_double2(cs, sn);
        }
}

// ---- k_coarse_snr: every sliding-window SNR of move_fft_snr_runtime_avg.m:17-28, all in parallel ----
// grid (ceil(nwin/256), S), block 256.  FFT16: the reference geometry (16-point windows, radix-2 in registers); the
// other instance serves any window length 2..64 by direct DFTs (a called function: keeping it out of the FFT16
// instance keeps that one free of scratch memory and within 128 registers -- four blocks per CU).
// issue priority by dispatch round (four workgroups per CU; only while the whole grid is resident)
#define CS_PRIO(P0, P1, P2, P3) if (gridDim.x * gridDim.y <= 1024u) { const unsigned rnd_ = (blockIdx.y * gridDim.x + blockIdx.x) >> 8; if (rnd_ == 0) __builtin_amdgcn_s_setprio(P0); else if (rnd_ == 1) __builtin_amdgcn_s_setprio(P1); else if (rnd_ == 2) __builtin_amdgcn_s_setprio(P2); else __builtin_amdgcn_s_setprio(P3); }
#define CS_TILE 1024          /* k_coarse_snr<.,true>: most windows past the moving search one workgroup screens */
#define CS_SNR_THREADS 256
// Screening level test of one window for k_coarse_snr (see there): |r1|^2 < g(rho_X)^2 r0^2 proves SNR < X dB.
// Four consecutive windows per call (b .. b+3 of the staged tile xt), the lag products sliding from one to the next;
// bit k of the result: window b+k has to be computed in full.
__device__ __forceinline__ unsigned screen4(const cplx* xt, int b, double cr, double ci, double gx2) {
    cplx xp = xt[b];
    xp.x -= cr; xp.y -= ci;
    double rr = 0.0, ri = 0.0, e = xp.x * xp.x + xp.y * xp.y;
    cplx xo[4];                                          // x'[b..b+3]: the samples that leave
    double pr[3], pi_[3];                                // p[b..b+2]:  the lag products that leave
    xo[0] = xp;
#pragma unroll
    for (int n = 1; n <= 3; ++n) {
        cplx xn = xt[b + n];
        xn.x -= cr; xn.y -= ci;
        const double qr = xn.x * xp.x + xn.y * xp.y, qi = xn.y * xp.x - xn.x * xp.y;   // x'[n] conj(x'[n-1])
        pr[n - 1] = qr; pi_[n - 1] = qi; xo[n] = xn;
        rr += qr; ri += qi;
        e += xn.x * xn.x + xn.y * xn.y;
        xp = xn;
    }
#pragma unroll 4
    for (int n = 4; n < 16; ++n) {
        cplx xn = xt[b + n];
        xn.x -= cr; xn.y -= ci;
        rr += xn.x * xp.x + xn.y * xp.y; ri += xn.y * xp.x - xn.x * xp.y;
        e += xn.x * xn.x + xn.y * xn.y;
        xp = xn;
    }
    cplx x_last = xp, x_lo = xo[0];                      // x'[b+15], x'[b]
    unsigned need = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k > 0) {                                     // slide by one: sample b+k-1 leaves, b+k+15 enters
            cplx xn = xt[b + k + 15];
            xn.x -= cr; xn.y -= ci;
            rr += xn.x * x_last.x + xn.y * x_last.y - pr[k - 1];
            ri += xn.y * x_last.x - xn.x * x_last.y - pi_[k - 1];
            e += xn.x * xn.x + xn.y * xn.y - (xo[k - 1].x * xo[k - 1].x + xo[k - 1].y * xo[k - 1].y);
            x_last = xn;
            x_lo = xo[k];
        }
        // + the wrap term x'[first] conj(x'[last]) of the circular lag
        const double r1r = rr + (x_lo.x * x_last.x + x_lo.y * x_last.y), r1i = ri + (x_lo.y * x_last.x - x_lo.x * x_last.y);
        if (!(r1r * r1r + r1i * r1i < gx2 * (e * e))) need |= 1u << k;     // (NaN: computed in full)
    }
    return need;
}

// k_coarse_snr: the per-window SNRs (move_fft_snr_runtime_avg.m:17-27) of the moving search, one window per thread.
// grid (blocks, S).  SCREEN (latency path, 16-point windows): the table also covers every later window of the stream so
// that the hop walk of k_coarse_scan is a table look-up; block j does windows [256 j, 256 j + 256) of the moving search
// in full AND the j-th tile of a.snr_tile later windows, of which all but a few percent are ruled out without a spectrum:
//   With P_k the window's 16 powers (DC removed), S its strongest bin with both neighbours and rho = S / (sum - S) the
//   quantity whose 10 log10 is the SNR, the circular lag-1 autocorrelation r1 = sum_n x[(n+1) mod 16] conj(x[n]) =
//   (1/16) sum_k P_k e^(2 pi i k/16) satisfies |r1| >= Re(r1 e^(-2 pi i m/16)) >= (1/16) [S cos(2 pi/16) - (sum - S)]
//   (m the strongest bin), and r0 = sum_n |x[n]|^2 = (1/16) sum_k P_k; so |r1| / r0 >= (cos(pi/8) rho - 1) / (rho + 1) =: g(rho),
//   increasing in rho.  A window with |r1| < g(rho_X) r0 therefore has SNR < X dB = a.snr_screen_db: it gets -inf ("below X")
//   and no FFT.  The hop walk only asks "snr - hit_avg_snr > th" and uses the table only when hit_avg_snr + th > X.
// The survivors are compacted and computed in full after the moving search's windows.
// REFG: FCCH_coarse_position on a stream decimated to an eighth of the symbol rate (the drivers' geometry: 16-point windows,
// 160-window average, 3594 samples of moving search) -- the window geometry as compile-time constants
template <bool FFT16, bool SCREEN = false, bool REFG = false>
__global__ void __launch_bounds__(SCREEN ? CS_SNR_THREADS : 256) __attribute__((amdgpu_waves_per_eu(4, 8))) k_coarse_snr(CoarseArgs a) {
    __shared__ cplx tw[64];
    const CoarseGeom g = REFG ? CoarseGeom{16, 160, a.th0, 3594L, 3594L - 15L} : coarse_geom(a);
    if (g.fft_len > 64 || g.fft_len < 2 || g.n_first > a.len) return;   // the scan kernel reports the index error
    const int tid = threadIdx.x;
    if (!FFT16) coarse_twiddles(tw, g.fft_len, tid, 256);
    __syncthreads();
    CS_PRIO(0, 1, 2, 3)                                  // staging and screening: the youngest workgroup of the CU first (see FC_PRIO in k_fine_cert's header)
    const long i = (long)blockIdx.x * 256 + tid;
    // the stream's mean comes from the front kernel's partial byte sums: requested by the first wave BEFORE the samples below, so
    // that the two round trips to L2 overlap (a tenth of this kernel's time); summed after them
    unsigned long long pvi[4] = {0, 0, 0, 0}, pvq[4] = {0, 0, 0, 0};
    const bool pre_mean = FFT16 && a.mean_corr && a.npartial <= 256;
    if (pre_mean && tid < 64) {
        const unsigned long long* pp = a.partial + (size_t)blockIdx.y * a.npartial * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int b = tid + 64 * k;
            if (b < a.npartial) { pvi[k] = pp[2 * b]; pvq[k] = pp[2 * b + 1]; }
        }
    }
    __shared__ cplx xw[FFT16 ? 256 + 16 : 1];
    __shared__ cplx xt[SCREEN ? CS_TILE + 20 : 1];
    __shared__ unsigned short list[SCREEN ? CS_TILE : 1];
    __shared__ int sh_cnt;
    const int nthr = SCREEN ? CS_SNR_THREADS : 256;
    int nw = 0;
    long w0 = 0;
    if (FFT16) {
        // the block's 256 windows overlap in all but one sample each: stage the 271 samples they cover in LDS once
        // instead of 16 loads per window through the vector cache (raw: the DC term is removed in the spectrum)
        const cplx* sp = a.s + (size_t)blockIdx.y * a.s_stride;
        const long b0 = (long)blockIdx.x * 256;
        for (int j = tid; j < 256 + 15; j += nthr) xw[j] = b0 + j < a.len ? sp[b0 + j] : make_double2(0.0, 0.0);
        if (SCREEN) {
            w0 = g.nwin + (long)blockIdx.x * a.snr_tile;
            const long left = a.snr_nwin - w0;
            nw = left <= 0 ? 0 : (left < a.snr_tile ? (int)left : a.snr_tile);
            for (int j = tid; j < nw + 15; j += nthr) xt[j] = sp[w0 + j];
            if (tid == 0) sh_cnt = 0;
        }
    }
    double mr = 0.0, mi = 0.0;
    if (pre_mean) {
        __shared__ unsigned long long sh_tot2[2];
        if (tid < 64) {
            unsigned long long si = (pvi[0] + pvi[1]) + (pvi[2] + pvi[3]), sq = (pvq[0] + pvq[1]) + (pvq[2] + pvq[3]);   // (integers: any order)
            for (int off = 32; off > 0; off >>= 1) {
                si += __shfl_down(si, off, 64);
                sq += __shfl_down(sq, off, 64);
            }
            if (tid == 0) { sh_tot2[0] = si; sh_tot2[1] = sq; }
        }
        __syncthreads();                                 // (also: xw / xt complete)
        mr = (double)sh_tot2[0] / (double)a.n0;          // raw2iq.m:8, as in stream_mean
        mi = (double)sh_tot2[1] / (double)a.n0;
    } else {
        if (a.mean_corr) { unsigned long long ti, tq; stream_mean(a, blockIdx.y, &mr, &mi, &ti, &tq); }   // (whole block: wave 0 adds, barrier)
        __syncthreads();
    }
    const DecView s = dec_view(a, blockIdx.y, mr, mi);
    DEV_STAMP(KID_COARSE_SNR, blockIdx.y * gridDim.x + blockIdx.x, 0);
    double* tab = a.snr_g + (size_t)blockIdx.y * a.snr_stride;
    if (FFT16) {
        if (SCREEN && nw > 0) {
            const double gx2 = a.snr_gx2;                // g(rho_X)^2 with its margin, from the host (an exp10 and a divide per thread otherwise)
            for (int b = 4 * tid; b < nw; b += 4 * nthr) {
                const unsigned need = screen4(xt, b, s.cr, s.ci, gx2);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool in = b + k < nw;
                    const bool pass = in && ((need >> k) & 1u);
                    const unsigned long long m = __ballot(pass);
                    int base = 0;
                    if ((tid & 63) == 0 && m) base = atomicAdd(&sh_cnt, __popcll(m));
                    base = __shfl(base, 0, 64);
                    if (pass) list[base + __popcll(m & ((1ull << (tid & 63)) - 1ull))] = (unsigned short)(b + k);
                    else if (in) tab[w0 + b + k] = -INFINITY;
                }
            }
        }
        if (SCREEN) __syncthreads();
        CS_PRIO(3, 2, 1, 0)                              // the spectra: oldest first (16.7 -> 16.0 us at 64 streams)
        // FFT passes: the moving search's window of each thread, then the survivors (~45 per tile: the first wave)
        const int cnt = SCREEN ? sh_cnt : 0;
        for (int k0 = -256; k0 < cnt; k0 += 256) {
            const cplx* src = nullptr;
            int dst = 0;                                 // (a stream's table stays far below 2^31 entries)
            if (k0 < 0) {
                if (i < g.nwin) { src = xw + tid; dst = (int)i; }
            } else if (k0 + tid < cnt) {
                const int wi = list[k0 + tid];
                src = xt + wi; dst = (int)w0 + wi;
            }
            if (src) tab[dst] = window_snr16_from(s, src);
        }
    } else {
        if (i < g.nwin) tab[i] = window_snr_generic(s, i, g.fft_len, tw);
    }
    if (FFT16 && blockIdx.x == 0 && s.n_head > 0) {   // (block-uniform) redo the head windows with per-sample DC removal
        __shared__ cplx hx[64 + 8];
        const int nh = s.n_head < 8 ? s.n_head : 8;
        if (tid < nh + g.fft_len - 1 && tid < g.n_first) hx[tid] = dv_load(s, tid);
        __syncthreads();
        if (tid < nh && tid < g.nwin) tab[tid] = window_snr16_head(hx + tid, s.floor2);
    }
    DEV_STAMP(KID_COARSE_SNR, blockIdx.y * gridDim.x + blockIdx.x, 1);
}

__device__ void d_fine_setup(StreamState* st, int ov, int lvl, int min_hits, int lane);   // kernels_estim.h
__device__ void d_scan_accept(const StreamState* st, int s, double* snr_numhit, double* positions,
                              double* pos_snr, int* counts, const DevParams& P, int lane);   // kernels_estim.h

// ---- k_coarse_scan: first hit + hop walk of FCCH_coarse_position, one workgroup per stream ----
// grid S, block 256.  LDS: state copy | 64 twiddles | 4 x 36 hop samples | 2 x 11 x 17 hop powers | snr[mv_len + nwin + 64].
//
// (1) First hit, move_fft_snr_runtime_avg.m:30-42.  The loop is serial only through the ROUNDING of sum_snr (two
// dependent fp64 adds per window); its decisions (snr - sum_snr/mv_len > th) almost never depend on that rounding.
// Each thread takes a run of windows, sums the mv_len entries its first window sees directly and slides from there.
// These sums differ from the reference's serially rounded ones by at most
//     (2 nwin + mv_len + 2 per) * 2^-53 * mv_len * V     (every fp64 add errs by <= 2^-53 |result|, |result| <= mv_len V,
//                                                         V = 1000 >= |snr| checked per window, the seed is 999)
// i.e. <= 1e-9 on the average for the reference geometry.  A window whose margin to the threshold exceeds `delta`
// (1e-6 + 4x that bound) is decided for good, and the first decided hit stands if no undecided window precedes it.
// Otherwise (a margin inside delta, a non-finite SNR) wave 0 replays the reference's running-sum updates in the
// reference's order, 64 windows at a time: every lane advances the same wave-uniform sum from broadcast LDS reads,
// lane j keeps the sum window j sees, then the 64 lanes evaluate snr - sum/mv_len > th with the true divide of :30
// and a ballot picks the first hit (~18 ns per window: the latency of two dependent v_add_f64).  mode 1 -- the API's
// move_fft_snr_runtime_avg, which REPORTS hit_avg_snr -- always takes the exact replay.
// (2) Hop walk, FCCH_coarse_position.m:32-86 + specific_fft_snr_fix_avg.m:10-25, by wave 0.  A hop looks at the 11
// windows around +10 frames and, if those all miss, the 11 around +11 frames.  All 22 spectra of a hop come from one
// pass: lane (group, half, bin k) takes bin k of its half's first window by a direct 16-point DFT and slides it
// through the next windows (X_k(t+1) = (X_k(t) + x[t+16] - x[t]) e^{+2 pi i k/16}); the powers go through LDS to
// 22 lanes that form the SNRs (first max, 3 bins, log10) and a ballot applies the reference's first-hit rule.  The
// samples the NEXT hop can ask for are known one hop early (4 rows of 36 for the four (branch, group) pairs), so
// every hop also fetches those and the following hop finds its windows in LDS.  With a certified average
// (1) the hop decisions carry the same delta check; a decision inside delta repeats the stream with the exact replay.
// WAVES: minimum waves per SIMD the register allocator must leave room for (2: small batches, 4: four workgroups per CU).
#define CS_ROW 36
__host__ __device__ inline size_t coarse_scan_lds_fixed() {
    return ((sizeof(StreamState) + 15) & ~(size_t)15) + 64 * sizeof(cplx) + 4 * CS_ROW * sizeof(cplx) + (2 * 11 * 17 + 2 * MAXH) * sizeof(double);
}

// INL (throughput batches, 16-point windows): the moving search's window SNRs are computed HERE, straight into the LDS copy the
// scan reads -- k_coarse_snr<true,false>'s arithmetic, value for value -- so the table never makes the trip through HBM and the
// k_coarse_snr launch in front of this kernel is gone (a.snr_g non-null: it is still written out, for gsmcal_last_batch_snr).
template <int WAVES, bool FFT16, bool REFG = false, bool INL = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) k_coarse_scan(StreamState* __restrict__ sts, CoarseArgs a_in) {
    CoarseArgs a = a_in;
    if (REFG) { a.mode = 0; a.decimation_ratio = 8; }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int sh_hit;     // first hit window (0-based) or INT_MAX
    __shared__ double sh_avg;  // sum/mv_len seen by the hit window
    __shared__ int sh_pred;    // first window the certificate decided as a hit, INT_MAX if none
    __shared__ int sh_unc;     // first window the certificate could not decide, INT_MAX if none
    __shared__ int sh_exact;   // 1: run the exact serial replay
    __shared__ int sh_redo;    // a hop decision sat inside delta: repeat with the exact replay
    StreamState* st = (StreamState*)smem;                 // LDS copy of the stream state
    cplx* tw = (cplx*)(smem + ((sizeof(StreamState) + 15) & ~(size_t)15));
    cplx* rows = tw + 64;                                 // 4 x CS_ROW samples of the current hop
    double* Pb = (double*)(rows + 4 * CS_ROW);            // [group][window][17]: powers of the hop's 22 windows
    double* hop_sig_base = Pb + 2 * 11 * 17;             // (signal, noise) of the hit each hop settled on
    double* snr_s = hop_sig_base + 2 * MAXH;
    StreamState* st_g = sts + blockIdx.x;
    const long len = a.len;
    const CoarseGeom g = REFG ? CoarseGeom{16, 160, a.th0, 3594L, 3594L - 15L} : coarse_geom<false>(a);   // (derived here, under the latency of the loads below: with the host's values this kernel ran 1.2 us SLOWER)
    const int fft_len = g.fft_len, mv_len = g.mv_len;
    const double th = g.th;
    const int tid = threadIdx.x;
#define CS_STAMP(i) DEV_STAMP(KID_COARSE_SCAN, blockIdx.x, i)
    CS_STAMP(0);
    const bool bad = fft_len > 64 || fft_len < 2 || g.n_first > len || (FFT16 != (fft_len == 16));   // s(1:n_first): MATLAB index error
    // padded copy S = [999 x mv_len | snr[0..nwin) | 0 x 64]: S[q] is the SNR evicted by window q
    // (999 while the history still holds its seed, move_fft_snr_runtime_avg.m:11), S[mv_len+q] is
    // window q's own SNR; every load of the unrolled recurrence below is unconditional.  The global loads go out
    // first, 16 per lane in flight, and everything else of the prologue runs under their latency.
    // the front kernel's partial byte sums (the stream's mean): requested by the first wave BEFORE the table loads below, so
    // that the two round trips to L2 overlap; summed after them
    unsigned long long pvi[4] = {0, 0, 0, 0}, pvq[4] = {0, 0, 0, 0};
    const bool pre_mean = a.mean_corr && a.npartial <= 256;
    if (pre_mean && tid < 64) {
        const unsigned long long* pp = a.partial + (size_t)blockIdx.x * a.npartial * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int b = tid + 64 * k;
            if (b < a.npartial) { pvi[k] = pp[2 * b]; pvq[k] = pp[2 * b + 1]; }
        }
    }
    if (!INL && !bad && a.mode != 2) {
        const double* sg = a.snr_g + (size_t)blockIdx.x * a.snr_stride;
        const long tot = mv_len + g.nwin + 64;
        for (long i0 = 0; i0 < tot; i0 += 16 * 256) {
            double rv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const long i = i0 + tid + 256 * u;
                rv[u] = (i >= mv_len && i < mv_len + g.nwin) ? sg[i - mv_len] : (i < mv_len ? 999.0 : 0.0);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const long i = i0 + tid + 256 * u;
                if (i < tot) snr_s[i] = rv[u];
            }
        }
    }
    double mr0 = 0.0, mi0 = 0.0;
    unsigned long long ti0 = 0, tq0 = 0;
    if (pre_mean) {                                                           // batch path: means from the front kernel
        __shared__ unsigned long long sh_tot2[2];
        if (tid < 64) {
            unsigned long long si = (pvi[0] + pvi[1]) + (pvi[2] + pvi[3]), sq = (pvq[0] + pvq[1]) + (pvq[2] + pvq[3]);   // (integers: any order)
            for (int off = 32; off > 0; off >>= 1) {
                si += __shfl_down(si, off, 64);
                sq += __shfl_down(sq, off, 64);
            }
            if (tid == 0) { sh_tot2[0] = si; sh_tot2[1] = sq; }
        }
        __syncthreads();
        ti0 = sh_tot2[0]; tq0 = sh_tot2[1];
        mr0 = (double)ti0 / (double)a.n0;
        mi0 = (double)tq0 / (double)a.n0;
    } else if (a.mean_corr) stream_mean(a, blockIdx.x, &mr0, &mi0, &ti0, &tq0);
    const DecView s = dec_view(a, blockIdx.x, a.mean_corr ? mr0 : st_g->mean_re, a.mean_corr ? mi0 : st_g->mean_im);
    if (a.mean_corr) {
        // batch path: this kernel is the first to touch the stream's state -- build it from scratch in LDS (all zero,
        // then the sentinels the reference functions start from), no memset / finish-mean launches needed
        uint4* dst = (uint4*)st;
        for (int i = tid; i < (int)(sizeof(StreamState) / 16); i += 256) dst[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        if (tid == 0) {
            st->n0 = a.n0;
            st->sum_i = ti0; st->sum_q = tq0;
            st->mean_re = mr0; st->mean_im = mi0;
            st->sampling_ppm1 = INFINITY; st->carrier_ppm1 = INFINITY;
            st->sampling_ppm2 = INFINITY; st->carrier_ppm2 = INFINITY;
            st->fcch_is_sentinel = 1;
        }
    } else {
        const uint4* src = (const uint4*)st_g;
        uint4* dst = (uint4*)st;
        for (int i = tid; i < (int)(sizeof(StreamState) / 16); i += 256) dst[i] = src[i];
    }
    // twiddles of the window DFTs: the batch path with the full SNR table needs them only if its hop walk falls back to
    // its own spectra (built there); sincospi on the way to the first barrier costs the whole block ~1 us
    const bool tw_lazy = FFT16 && a.mode == 0 && a.snr_nwin > 0;
    auto make_tw = [&]() {
        if (fft_len >= 2 && fft_len <= 64)
            for (int i = tid; i < fft_len; i += 256) {
                double sn, cs;
                sincospi(-2.0 * (double)i / (double)fft_len, &sn, &cs);
                tw[i] = make_double2(cs, sn);
            }
    };
    if (!tw_lazy) make_tw();
    if (INL && FFT16 && !bad && a.mode != 2) {
        // k_coarse_snr<true, false>: 256 windows per round from the 271 raw samples they cover (the DC term is removed in the
        // spectrum), the next round's samples requested before this round's spectra; then the head windows again with per-sample
        // DC removal.  S = [999 x mv_len | snr | 0 x 64] as above.
        const long nwin = g.nwin;
        for (int i = tid; i < mv_len; i += 256) snr_s[i] = 999.0;
        for (int i = tid; i < 64; i += 256) snr_s[mv_len + nwin + i] = 0.0;
        cplx* xw = rows;                                  // 271 samples: the hop buffers (rows | Pb, 5.3 KB) are not in use yet
        const cplx* sp = s.s;
        const cplx z = make_double2(0.0, 0.0);
        cplx c0 = tid < len ? sp[tid] : z, c1 = (tid < 15 && 256 + tid < len) ? sp[256 + tid] : z;
        for (long b0 = 0; b0 < nwin; b0 += 256) {
            xw[tid] = c0;
            if (tid < 15) xw[256 + tid] = c1;
            __syncthreads();
            const long nb = b0 + 256;
            if (nb < nwin) {
                c0 = nb + tid < len ? sp[nb + tid] : z;
                if (tid < 15) c1 = nb + 256 + tid < len ? sp[nb + 256 + tid] : z;
            }
            if (b0 + tid < nwin) snr_s[mv_len + b0 + tid] = window_snr16_from(s, xw + tid);
            __syncthreads();
        }
        if (s.n_head > 0) {                               // (block-uniform)
            cplx* hx = rows;
            const int nh = s.n_head < 8 ? s.n_head : 8;
            if (tid < nh + fft_len - 1 && tid < g.n_first) hx[tid] = dv_load(s, tid);
            __syncthreads();
            if (tid < nh && tid < nwin) snr_s[mv_len + tid] = window_snr16_head(hx + tid, s.floor2);
        }
        if (a.snr_g) {
            __syncthreads();
            double* tab = a.snr_g + (size_t)blockIdx.x * a.snr_stride;
            for (long i = tid; i < nwin; i += 256) tab[i] = snr_s[mv_len + i];
        }
    }
    __syncthreads();
    const bool want_cert = !bad && a.mode == 0;
    if (tid == 0) {
        st->n_coarse = 0;
        st->coarse_hit_flag = 0;
        st->hit_avg_snr = INFINITY;
        st->mv_hit_idx = -1.0;
        st->mv_hit_snr = INFINITY;
        sh_hit = 0x7fffffff;
        sh_pred = 0x7fffffff;
        sh_unc = 0x7fffffff;
        sh_exact = want_cert ? 0 : 1;
        sh_redo = 0;
        if (bad) set_status(st, 3, GSMCAL_E_INDEX);
    }
    __syncthreads();
    int n = 0;
    CS_STAMP(1);
    const double cert_V = 1000.0;
    const int sc_total = mv_len + (int)g.nwin;            // entries of S that windows can see
    const int sc_per = (sc_total + 255) / 256;            // entries (and windows) per thread
    const double cert_delta = 1e-6 + 4.0 * 1.1102230246251565e-16 * cert_V *
                              ((2.0 * (double)g.nwin + (double)mv_len + 64.0) +
                               2.0 * ((double)sc_per + 32.0) * (double)sc_total / (double)mv_len);
    if (want_cert) {
        // window sums from a block-wide prefix scan of S: C[p] = sum_{i<p} S[i] at every thread's chunk start (Cb) plus
        // a partial chunk sum; window j sees C[j+mv_len] - C[j].  (<= sc_per + 32 fp64 adds of magnitude <= sc_total V per
        // C value: the second term of cert_delta.)
        __shared__ double sh_ws[4];
        double* Cb = (double*)rows;                        // 257 doubles; the hop buffers are not in use yet
        const int lane = tid & 63, wave = tid >> 6;
        const int p0 = tid * sc_per;
        double loc = 0.0;
        for (int i = 0; i < sc_per; ++i) loc += (p0 + i < sc_total) ? snr_s[p0 + i] : 0.0;
        const double inc = wave_scan_incl(loc);
        if (lane == 63) sh_ws[wave] = inc;
        __syncthreads();
        double base = 0.0;
        for (int i = 0; i < wave; ++i) base += sh_ws[i];
        Cb[tid] = base + (inc - loc);
        if (tid == 255) Cb[256] = base + inc;
        __syncthreads();
        const int j0 = p0, j1 = j0 + sc_per < (int)g.nwin ? j0 + sc_per : (int)g.nwin;
        if (j0 < j1) {
            const int idx = j0 + mv_len, bq = idx / sc_per, rq = idx - bq * sc_per;
            double cx = Cb[bq];
            for (int i = 0; i < rq; ++i) cx += snr_s[bq * sc_per + i];
            double sm = cx - Cb[tid];
            const double inv = 1.0 / (double)mv_len;
            for (int j = j0; j < j1; ++j) {
                const double v = snr_s[mv_len + j];
                const double mg = (v - sm * inv) - th;
                if (!(fabs(v) <= cert_V) || !(fabs(mg) > cert_delta)) { atomicMin(&sh_unc, j); break; }   // (NaN lands here)
                if (mg > 0.0) { atomicMin(&sh_pred, j); break; }
                sm += v - snr_s[j];
            }
        }
        __syncthreads();
        if (tid < 64) {
            // the average the decided hit sees (64 lanes + a reduction tree: within the same bound)
            const int pred = sh_pred;
            const bool ok = !(sh_unc < pred);
            double sm = 0.0;
            if (ok && pred != 0x7fffffff) {
                for (int q = tid; q < mv_len; q += 64) sm += snr_s[pred + q];
            }
            sm = wave_sum(sm);
            if (tid == 0) {
                if (!ok) sh_exact = 1;
                else if (pred != 0x7fffffff) { sh_hit = pred; sh_avg = sm / (double)mv_len; }
            }
        }
        __syncthreads();
    }
    CS_STAMP(2);
    if (!bad && a.mode != 2) {
      for (int pass = 0; pass < 2; ++pass) {               // pass 1 only after a hop decision fell inside delta
        // ---- move_fft_snr_runtime_avg ----
        const long nwin = g.nwin;
        const double dmv = (double)mv_len;
        const bool exact = sh_exact != 0;                  // block-uniform
        if (tid < 64 && exact) {
            const int lane = tid;
            // :11-12 store = 999*ones(1,mv_len); sum_snr = sum(store): sequential sum of 999s (wave-uniform)
            double sum_snr = 0.0;
            for (int i = 0; i < mv_len; ++i) sum_snr += 999.0;
            for (int base = 0; base < (int)nwin; base += 64) {
                const int i = base + lane;
                const double nw = snr_s[mv_len + i];                  // (in-bounds by the zero suffix)
                double mine = 0.0;
                const double* So = snr_s + base;                     // evicted values
                const double* Sn = snr_s + mv_len + base;            // new values
#pragma unroll
                for (int j0 = 0; j0 < 64; j0 += 8) {
                    double o[8], v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {                    // wave-uniform (broadcast) LDS reads
                        v[u] = Sn[j0 + u];
                        o[u] = So[j0 + u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        mine = lane == j0 + u ? sum_snr : mine;      // the sum window j sees
                        sum_snr = sum_snr - o[u];                    // :37
                        sum_snr = sum_snr + v[u];                    // :38
                    }
                }                                                    // (updates past nwin are never used)
                const double avg = mine / dmv;
                const bool h = i < nwin && (nw - avg > th);          // :30-32 strict >
                const unsigned long long m = __ballot(h);
                if (m) {
                    const int f = __ffsll((long long)m) - 1;
                    if (lane == f) { sh_hit = (int)base + f; sh_avg = avg; }
                    break;
                }
            }
        }
        __syncthreads();
        CS_STAMP(3);
        const int hit = sh_hit;
        if (hit != 0x7fffffff && tid == 0) {
            const double h_snr = snr_s[mv_len + hit];
            const double h_pta = h_snr - sh_avg;
            st->coarse_hit_flag = 1;
            st->mv_hit_idx = (double)(hit + 1);
            st->mv_hit_snr = h_snr;
            st->hit_avg_snr = h_snr - h_pta;                         // :48
        }
        __syncthreads();
        if (a.mode == 0) {
            if (hit == 0x7fffffff) {
                if (tid == 0) set_status(st, 3, GSMCAL_S_NO_FCCH);
            } else if (FFT16 && a.snr_nwin >= len - (fft_len - 1) &&
                       st->hit_avg_snr + th > a.snr_screen_db + 1e-6 + (exact ? 0.0 : cert_delta)) {
                // ---- hop loop of FCCH_coarse_position.m:32-86 on the SNR table of the whole stream (k_coarse_snr computed every
                // window, not only those of the moving search): specific_fft_snr_fix_avg.m:10-25 is then 11 table entries and
                // a ballot.  One wave; lanes 0..10 hold the +10-frame candidates, lanes 16..26 the +11-frame ones of the same
                // hop, fetched together.  With a certified average (1) a decision inside cert_delta repeats the stream with
                // the exact replay, as in the moving search.
                __shared__ int sh_n;
                if (tid < 64) {
                    const int lane = tid;
                    const double* tab = a.snr_g + (size_t)blockIdx.x * a.snr_stride;
                    const double hit_avg_snr = st->hit_avg_snr;
                    const int dec = a.decimation_ratio;
                    const long d0 = (long)round(12500.0 / (double)dec);     // :35 round() half away from zero
                    const long d1 = (long)round(13750.0 / (double)dec);     // :36
                    const int max_offset = 5, nt = 2 * max_offset + 1;
                    const long limit = (len - (fft_len - 1)) - max_offset;
                    const unsigned gmask = (1u << nt) - 1u;
                    long cur = hit + 1;
                    int nn = 1;
                    bool redo = false;
                    if (lane == 0) {
                        st->coarse_pos[0] = (double)((cur - 1) * dec + 1);  // :91
                        st->coarse_snr[0] = st->mv_hit_snr;
                    }
                    const int grp = lane >> 4, ci = lane & 15;
                    while (nn < MAXH) {
                        const long nx0 = cur + d0, nx1 = cur + d1;
                        if (nx0 > limit) break;                              // :49
                        const bool g1ok = nx1 <= limit;                      // :67
                        const bool cand = ci < nt && (grp == 0 || (grp == 1 && g1ok));
                        const long w = (grp ? nx1 : nx0) - max_offset + ci;  // 1-based window
                        const double v = cand ? tab[w - 1] : -INFINITY;
                        const bool is_hit = cand && (v - hit_avg_snr > th);  // NaN compares false
                        const bool is_unc = !exact && cand && !(fabs((v - hit_avg_snr) - th) > cert_delta);
                        const unsigned long long hb = __ballot(is_hit), ub = __ballot(is_unc);
                        const unsigned h0 = (unsigned)hb & gmask, u0 = (unsigned)ub & gmask;
                        const unsigned h1 = (unsigned)(hb >> 16) & gmask, u1 = (unsigned)(ub >> 16) & gmask;
                        // every candidate the reference looks at (up to the first hit, else all of the group) must be decided for good
                        if (u0 & (h0 ? (2u << (__ffs(h0) - 1)) - 1u : gmask)) { redo = true; break; }
                        int src;
                        long ncur;
                        if (h0) {
                            src = __ffs(h0) - 1;
                            ncur = nx0 - max_offset + src;
                        } else if (g1ok) {                                   // :65 try across the idle frame
                            if (u1 & (h1 ? (2u << (__ffs(h1) - 1)) - 1u : gmask)) { redo = true; break; }
                            if (!h1) break;                                  // both miss
                            src = __ffs(h1) - 1;
                            ncur = nx1 - max_offset + src;
                            src += 16;
                        } else break;                                        // :67
                        if (lane == src) {
                            st->coarse_pos[nn] = (double)((ncur - 1) * dec + 1);
                            st->coarse_snr[nn] = v;
                        }
                        cur = ncur;
                        ++nn;
                    }
                    if (lane == 0) {
                        sh_n = nn;
                        if (redo) sh_redo = 1;
                        st->n_coarse = nn;
                    }
                }
                __syncthreads();
                n = sh_n;
            } else {
                // ---- hop loop of FCCH_coarse_position.m:32-86 (positions 1-based, decimated units), whole block ----
                if (tw_lazy) make_tw();                                  // (block-uniform; the barrier below the set-up covers it)
                __shared__ long sh_cur;
                __shared__ int sh_stop, sh_need1, sh_took1;
                const int lane = tid & 63;
                const double hit_avg_snr = st->hit_avg_snr;
                const int dec = a.decimation_ratio;
                const long d0 = (long)round(12500.0 / (double)dec);     // :35 round() half away from zero
                const long d1 = (long)round(13750.0 / (double)dec);     // :36
                const int max_offset = 5, nt = 2 * max_offset + 1;
                const long limit = (len - (fft_len - 1)) - max_offset;
                long cur = hit + 1;
                n = 1;
                if (tid == 0) {
                    st->coarse_pos[0] = (double)((cur - 1) * dec + 1);  // :91
                    st->coarse_snr[0] = st->mv_hit_snr;
                    sh_stop = 0; sh_need1 = 0; sh_took1 = 0;
                }
                // Certified decisions without the logarithm: 10 log10(sig/noise) - hit_avg_snr > th  <=>  sig/noise > R,
                // R = 10^((hit_avg_snr + th)/10).  The reference's rounded evaluation sits within 1e-12 dB of the true
                // value and hit_avg_snr within cert_delta of the exactly replayed one, so sig > R (1 + eps) noise is a
                // hit and sig < R (1 - eps) noise a miss for good, eps = 0.2303 cert_delta + 1e-12 (d(ratio)/ratio =
                // ln(10)/10 per dB); anything in between sends the stream to the exact replay.  Only the SNR of the
                // hit a hop settles on is reported: those logarithms are taken after the walk, one lane per hop.
                const double R = exp10((hit_avg_snr + th) / 10.0);
                const double r_eps = 0.2303 * cert_delta + 1e-12;
                const double Rhi = R * (1.0 + r_eps), Rlo = R * (1.0 - r_eps);
                double* hop_sig = hop_sig_base;
                constexpr bool fast = FFT16;
                const bool ratio_mode = !exact && fast;
                // loader threads (tid < 4*CS_ROW): one look-ahead sample each; spectrum threads (tid < 11*16): one
                // (window, bin) each; wave 0 lanes < 11 turn a group's powers into decisions
                cplx pf = make_double2(0.0, 0.0);
                long pg = -1;
                long pbase0 = 0, pbase1 = 0;                               // nx0 - 11, nx1 - 11 of the hop that fetched the rows
                bool have_pf = false;
                const cplx* sraw = s.s;
                const int dw = tid >> 4, dk = tid & 15;                    // spectrum role
                __syncthreads();
#ifdef GSMCAL_DEVTIMING
                unsigned long long cyc[5] = {0, 0, 0, 0, 0}, c0 = 0;
#define HOP_CYC(i) do { const unsigned long long c1 = __builtin_readcyclecounter(); cyc[i] += c1 - c0; c0 = c1; } while (0)
#else
#define HOP_CYC(i) do { } while (0)
#endif
                while (n < MAXH) {                                         // (cur, n: block-uniform)
                    const long nx0 = cur + d0, nx1 = cur + d1;
                    if (nx0 > limit) break;                              // :49
#ifdef GSMCAL_DEVTIMING
                    c0 = __builtin_readcyclecounter();
#endif
                    const bool g1ok = nx1 <= limit;
                    const bool took1 = sh_took1 != 0;
                    int r0 = 0, r1 = 1, off0 = 0, off1 = 0;              // row and offset of each group's first window (start nx - 6)
                    if (fast) {
                        if (have_pf) {
                            // DC removal (DecView) on the way into LDS: the loads themselves were issued a hop ago
                            if (tid < 4 * CS_ROW) rows[tid] = pg < 0 ? make_double2(0.0, 0.0) : dv_fix(s, pf, pg);
                            r0 = took1 ? 2 : 0; r1 = r0 + 1;
                            off0 = (int)((nx0 - 6) - ((took1 ? pbase1 : pbase0) + d0));
                            off1 = (int)((nx1 - 6) - ((took1 ? pbase1 : pbase0) + d1));
                        } else if (tid < 64) {                           // first hop: fetch its own 2 x 26 samples now
                            const int gq = tid >> 5, i = tid & 31;
                            if (i < 26) {
                                const long gi = (gq ? nx1 : nx0) - 6 + i;
                                rows[gq * CS_ROW + i] = (gi >= 0 && gi < len && (gq == 0 || g1ok)) ? dv_load(s, gi) : make_double2(0.0, 0.0);
                            }
                        }
                        if (tid < 4 * CS_ROW) {
                            // fetch what the next hop can ask for (plain loads, nothing waits on them in this hop):
                            // rows (branch b, group g) start at nx_b + d_g - 11
                            const int rr = tid / CS_ROW, i = tid - rr * CS_ROW;
                            long gi = ((rr & 2) ? nx1 : nx0) - 11 + ((rr & 1) ? d1 : d0) + i;
                            if (gi < 0 || gi >= len || ((rr & 2) && !g1ok)) gi = -1;
                            pf = sraw[gi < 0 ? 0 : gi];
                            pg = gi;
                        }
                        pbase0 = nx0 - 11; pbase1 = nx1 - 11;
                        have_pf = true;
                    }
                    HOP_CYC(0);
                    __syncthreads();                                     // rows of this hop are in LDS
                    HOP_CYC(1);
                    double v = -INFINITY, c_sig = 0.0, c_noise = 0.0;
                    int found = -1;                                      // (wave 0) candidate the hop settles on, -1: none
                    bool stop = false;
                    for (int grp = 0; grp < 2; ++grp) {                  // +10-frame candidates, then (rarely) the +11-frame ones
                        if (grp == 1 && !(sh_need1 && g1ok)) break;      // block-uniform
                        if (fast) {
                            if (tid < nt * 16) {                         // bin dk of window dw: direct 16-point DFT, four partial sums
                                const cplx* x = rows + (grp ? r1 : r0) * CS_ROW + (grp ? off1 : off0) + dw;
                                double pr_[4] = {0.0, 0.0, 0.0, 0.0}, pi_[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                                for (int m = 0; m < 16; ++m) {
                                    const cplx t = tw[(dk * m) & 15], xv = x[m];
                                    pr_[m & 3] = fma(xv.x, t.x, pr_[m & 3]); pi_[m & 3] = fma(xv.x, t.y, pi_[m & 3]);
                                    pr_[m & 3] = fma(-xv.y, t.y, pr_[m & 3]); pi_[m & 3] = fma(xv.y, t.x, pi_[m & 3]);
                                }
                                const double xr = (pr_[0] + pr_[1]) + (pr_[2] + pr_[3]), xi = (pi_[0] + pi_[1]) + (pi_[2] + pi_[3]);
                                Pb[dw * 17 + dk] = xr * xr + xi * xi;
                            }
                            HOP_CYC(2);
                            __syncthreads();
                        }
                        if (tid < 64) {
                            const bool cand = lane < nt;
                            bool residue = false;
                            if (fast) {
                                if (cand) {                              // candidate `lane` (move_fft_snr_runtime_avg.m:22-27)
                                    double P[16];
                                    const double* pr = Pb + lane * 17;
#pragma unroll
                                    for (int k = 0; k < 16; ++k) P[k] = pr[k];
                                    // first maximum by a tree (ties: the lower index), total by a tree
                                    double mv[8]; int mi[8];
#pragma unroll
                                    for (int k = 0; k < 8; ++k) { const bool r = P[2 * k + 1] > P[2 * k]; mv[k] = r ? P[2 * k + 1] : P[2 * k]; mi[k] = 2 * k + (r ? 1 : 0); }
#pragma unroll
                                    for (int w = 4; w >= 1; w >>= 1)
#pragma unroll
                                        for (int k = 0; k < w; ++k) { const bool r = mv[2 * k + 1] > mv[2 * k]; mv[k] = r ? mv[2 * k + 1] : mv[2 * k]; mi[k] = r ? mi[2 * k + 1] : mi[2 * k]; }
                                    const double pc = mv[0], pm = pr[(mi[0] + 15) & 15], pp = pr[(mi[0] + 1) & 15];
                                    double sig = pm + pc;
                                    sig = sig + pp;
                                    if (exact) {
                                        double tot = 0.0;                // the reference's sequential sum(chn_tmp)
#pragma unroll
                                        for (int k = 0; k < 16; ++k) tot += P[k];
                                        c_sig = sig; c_noise = tot - sig;
                                        v = 10.0 * log10(c_sig / c_noise);
                                        if (tot <= s.floor2) v = __longlong_as_double(0x7ff8000000000000LL);   // rounding residue of the DC removal: 0/0 in the reference (snr_from_power)
                                    } else {                             // (the tree total differs from the sequential one by ~1e-16: inside eps)
                                        double t8[8];
#pragma unroll
                                        for (int k = 0; k < 8; ++k) t8[k] = P[2 * k] + P[2 * k + 1];
                                        const double tot = ((t8[0] + t8[1]) + (t8[2] + t8[3])) + ((t8[4] + t8[5]) + (t8[6] + t8[7]));
                                        c_sig = sig; c_noise = tot - sig;
                                        residue = tot <= s.floor2;       // ... a NaN SNR there: a definite miss
                                    }
                                }
                            } else if (cand) {
                                if (!FFT16) v = window_snr_generic(s, (grp ? nx1 : nx0) - max_offset - 1 + lane, fft_len, tw);
                            }
                            bool is_hit, is_unc;
                            if (ratio_mode) {
                                const bool okn = c_noise > 0.0 && c_noise < INFINITY && c_sig < INFINITY;
                                is_hit = cand && !residue && okn && c_sig > Rhi * c_noise;
                                is_unc = cand && !residue && !(is_hit || (okn && c_sig < Rlo * c_noise));
                            } else {
                                is_hit = cand && (v - hit_avg_snr > th);     // NaN compares false
                                is_unc = !exact && cand && !(fabs((v - hit_avg_snr) - th) > cert_delta);
                            }
                            const unsigned long long hm = __ballot(is_hit);
                            if (!exact) {
                                // every candidate the reference looks at (up to the first hit, else all of the group) must be
                                // decided for good
                                const unsigned long long unc = __ballot(is_unc);
                                const unsigned long long looked = hm ? (2ull << (__ffsll((long long)hm) - 1)) - 1 : (1ull << nt) - 1;
                                if (unc & looked) { if (lane == 0) { sh_redo = 1; sh_stop = 1; } stop = true; }
                            }
                            if (!stop) {
                                if (hm) {
                                    found = __ffsll((long long)hm) - 1;
                                    const long nxt = grp ? nx1 : nx0;
                                    const long ncur = nxt - max_offset + found;
                                    if (lane == found) {
                                        st->coarse_pos[n] = (double)((ncur - 1) * dec + 1);
                                        if (ratio_mode) { hop_sig[2 * n] = c_sig; hop_sig[2 * n + 1] = c_noise; }
                                        else st->coarse_snr[n] = v;
                                        sh_cur = ncur; sh_took1 = grp; sh_need1 = 0;
                                    }
                                } else if (grp == 0 && g1ok) {           // :65 try across the idle frame
                                    if (lane == 0) sh_need1 = 1;
                                } else if (lane == 0) sh_stop = 1;       // :67 / both miss
                            }
                        }
                        HOP_CYC(3);
                        __syncthreads();
                        if (sh_stop || !sh_need1) break;                 // block-uniform
                    }
                    if (sh_stop) break;
                    cur = sh_cur;
                    ++n;
                    __syncthreads();                                     // (sh_* are rewritten by the next hop)
                    HOP_CYC(4);
                }
#ifdef GSMCAL_DEVTIMING
                if (g_stamps && tid == 0 && blockIdx.x < DEV_STAMP_BLOCKS) {
                    unsigned long long* r = g_stamps + ((size_t)KID_COARSE_SCAN * DEV_STAMP_BLOCKS + blockIdx.x) * 16;
                    unsigned long long acc = r[0];
                    r[8] = acc;
                    for (int i = 0; i < 5; ++i) { acc += cyc[i]; r[9 + i] = acc; }   // cumulative cycles per hop phase (report shows /100)
                    r[14] = acc + 100ull * (unsigned long long)n;
                }
#endif
                __syncthreads();
                if (ratio_mode && tid >= 1 && tid < n)                   // the reported SNRs of hops 1..n-1 (:25)
                    st->coarse_snr[tid] = 10.0 * log10(hop_sig[2 * tid] / hop_sig[2 * tid + 1]);
                if (tid == 0) st->n_coarse = n;
            }
        }
        __syncthreads();
        const int redo = sh_redo;
        __syncthreads();
        if (!redo) break;                                  // block-uniform
        if (tid == 0) {                                    // start over with the exact replay
            sh_exact = 1; sh_redo = 0; sh_hit = 0x7fffffff;
            st->n_coarse = 0; st->coarse_hit_flag = 0; st->hit_avg_snr = INFINITY; st->mv_hit_idx = -1.0; st->mv_hit_snr = INFINITY;
        }
        n = 0;
        __syncthreads();
      }
      CS_STAMP(4);
    } else if (!bad) {
        // ---- specific_fft_snr_fix_avg stand-alone ----
        const long lo = a.t_lo, hi = a.t_hi;
        if (hi >= lo) {
            if (lo < 1) {
                if (tid == 0) set_status(st, 3, GSMCAL_E_INDEX);
            } else {
                // the reference's loop (specific_fft_snr_fix_avg.m:10-11) indexes window by window: a hit in a window that fits
                // returns before a later window would run past the end of s; only a miss up to there is MATLAB's index error
                const long hi_fit = hi + fft_len - 1 > len ? len - (fft_len - 1) : hi;
                const long cnt = hi_fit >= lo ? hi_fit - lo + 1 : 0;
                for (long i = tid; i < cnt; i += 256) snr_s[i] = FFT16 ? window_snr16(s, lo - 1 + i) : window_snr_generic(s, lo - 1 + i, fft_len, tw);
                __syncthreads();
                if (tid == 0) {
                    bool hit = false;
                    for (long i = 0; i < cnt && !hit; ++i)
                        if (snr_s[i] - a.avg_snr > th) {
                            st->coarse_hit_flag = 1;
                            st->mv_hit_idx = (double)(lo + i);
                            st->mv_hit_snr = snr_s[i];
                            hit = true;
                        }
                    if (!hit && hi_fit < hi) set_status(st, 3, GSMCAL_E_INDEX);
                }
            }
        }
    }
    __syncthreads();
    if (tid < 64 && a.fine_setup_ov > 0) d_fine_setup(st, a.fine_setup_ov, 0, a.min_hits, tid);
    if (tid < 64 && a.accept.snr_numhit)
        d_scan_accept(st, blockIdx.x, a.accept.snr_numhit, a.accept.positions, a.accept.pos_snr, a.accept.counts, a.P, tid);
    __syncthreads();
    CS_STAMP(5);
    {
        const uint4* src = (const uint4*)st;
        uint4* dst = (uint4*)st_g;
        for (int i = tid; i < (int)(sizeof(StreamState) / 16); i += 256) dst[i] = src[i];
    }
    CS_STAMP(6);
}

// ------------------------------------------------------------------------------------------------
// 1184-point spectra.  nfft = 148*ov = 37 * N2 with N2 = 4*ov: Cooley-Tukey split n = N2*n1 + n2,
// k = k1 + 37*k2, both factors as direct DFTs in LDS with exact table twiddles
// tw[m] = exp(-2*pi*i*m/nfft) (k_make_twiddles, sincospi).  ~70 complex MACs per output instead of
// 1184: used for the burst spectra (FCCH_fine_correction.m:148-150, carrier_correct_post_SCH.m:63-65)
// and to start the sliding DFT of the fine search at its first window.
//   MODE 0: argmax of |X|^2 in fftshift order -> PeakOut (one per window)
//   MODE 1: X of the window's first nfft samples -> global x0[s][w][k]
// grid (H, S), block 512.  LDS: x[nfft] | B[37][N2+1] | w37 | wN2.
// ------------------------------------------------------------------------------------------------
// ---- 37 x N2 Cooley-Tukey building blocks (N2 = 4*ov; 32 for the reference's 8x oversampling) ----
// tables in LDS: w37[m] = exp(-2 pi i m/37), wN2[m] = exp(-2 pi i m/N2); the inter-stage twiddle
// exp(-2 pi i n2 k1 / nfft) comes from the global table tw_g (one read per output).
// (taken from the global table: W_37^m = W_nfft^(N2 m) and W_N2^m = W_nfft^(37 m) are the very same correctly rounded
// values sincospi would give, without 69 lanes of fp64 sincospi at the head of every workgroup)
__device__ __forceinline__ void fft37_tables(cplx* w37, cplx* wN2, int N2, int tid, const cplx* __restrict__ tw_g) {
    if (tid < 37) w37[tid] = tw_g[N2 * tid];
    else if (tid >= 64 && tid < 64 + N2) wN2[tid - 64] = tw_g[37 * (tid - 64)];
}

// step 1: B[k1][n2] = W_nfft^(n2*k1) * sum_n1 x[N2*n1+n2] * W_37^(n1*k1)
__device__ __forceinline__ void fft37_step1(const cplx* xs, cplx* B, const cplx* w37, const cplx* __restrict__ tw_g,
                                            int nfft, int N2, int ldb, int tid, int nthreads) {
    for (int o = tid; o < nfft; o += nthreads) {
        const int k1 = o / N2, n2 = o - k1 * N2;
        double ar = 0.0, ai = 0.0;
        int idx = 0;
#pragma unroll 4                     // (the fallback of a fallback: unrolled completely it costs its callers 60 registers)
        for (int n1 = 0; n1 < 37; ++n1) {
            const cplx v = xs[N2 * n1 + n2], t = w37[idx];
            ar = fma(v.x, t.x, fma(-v.y, t.y, ar));
            ai = fma(v.x, t.y, fma(v.y, t.x, ai));
            idx += k1;
            idx = idx >= 37 ? idx - 37 : idx;
        }
        const cplx t = tw_g[n2 * k1];                   // n2*k1 < N2*37 = nfft
        B[k1 * ldb + n2] = make_double2(ar * t.x - ai * t.y, ar * t.y + ai * t.x);
    }
}

// step 1, four times cheaper, DESTROYING xs: the 37-point DFT of a column a[n1] = x[N2*n1+n2] by its symmetries,
//   Y[p]    = a[0] + sum_{n1=1..18} (a[n1]+a[37-n1]) cos(2 pi n1 p/37)  -  i sum_{n1=1..18} (a[n1]-a[37-n1]) sin(2 pi n1 p/37)
//   Y[37-p] = the same two sums with +i:  one pass yields both outputs, with 2 real MACs where the direct form has 8.
// Pass A rewrites xs in place (rows 1..18 <- sums, rows 36..19 <- differences), pass B takes one (n2, p) per thread.
// Same mathematics as fft37_step1, different summation order (fp64: ~1e-16 relative).  Contains a barrier.
// pre(n, v): applied to sample n as it is first read (pass A; row 0, which pass A does not touch, is rewritten too) --
// a rotation the caller would otherwise spend a pass over xs and a barrier on.
struct Fft37NoPre { __device__ __forceinline__ cplx operator()(int, const cplx& v) const { return v; } };
template <bool PRE = false, class F = Fft37NoPre>
__device__ __forceinline__ void fft37_step1_sym(cplx* xs, cplx* B, const cplx* w37, const cplx* __restrict__ tw_g,
                                                int nfft, int N2, int ldb, int tid, int nthreads, F pre = F()) {
    for (int o = tid; o < (PRE ? 19 : 18) * N2; o += nthreads) {
        if (PRE && o >= 18 * N2) { const int n2 = o - 18 * N2; xs[n2] = pre(n2, xs[n2]); continue; }
        const int n1 = 1 + o / N2, n2 = o - (n1 - 1) * N2;
        const int ia = N2 * n1 + n2, ib = N2 * (37 - n1) + n2;
        const cplx u = PRE ? pre(ia, xs[ia]) : xs[ia], v = PRE ? pre(ib, xs[ib]) : xs[ib];
        xs[ia] = make_double2(u.x + v.x, u.y + v.y);
        xs[ib] = make_double2(u.x - v.x, u.y - v.y);
    }
    __syncthreads();
    for (int o = tid; o < 19 * N2; o += nthreads) {
        const int p = o / N2, n2 = o - p * N2;               // p = 0: Y[0]; p = 1..18: Y[p] and Y[37-p]
        const cplx a0 = xs[n2];
        double pr = a0.x, pi = a0.y, qr = 0.0, qi = 0.0;
        int idx = 0;
#pragma unroll 6
        for (int n1 = 1; n1 <= 18; ++n1) {
            idx += p;
            idx = idx >= 37 ? idx - 37 : idx;
            const cplx sm = xs[N2 * n1 + n2], df = xs[N2 * (37 - n1) + n2];
            const cplx t = w37[idx];                         // (cos, -sin) of 2 pi n1 p / 37
            pr = fma(sm.x, t.x, pr); pi = fma(sm.y, t.x, pi);
            qr = fma(df.x, t.y, qr); qi = fma(df.y, t.y, qi);   // q = -sum d*sin
        }
        // Y[p] = P - i*Qs with Qs = -q:  P + i*q;   Y[37-p] = P - i*q
        {
            const double yr = pr - qi, yi = pi + qr;
            const cplx t = tw_g[n2 * p];
            B[p * ldb + n2] = make_double2(yr * t.x - yi * t.y, yr * t.y + yi * t.x);
        }
        if (p > 0) {
            const int k1 = 37 - p;
            const double yr = pr + qi, yi = pi - qr;
            const cplx t = tw_g[n2 * k1];
            B[k1 * ldb + n2] = make_double2(yr * t.x - yi * t.y, yr * t.y + yi * t.x);
        }
    }
}

// step 2 for one output bin k = k1 + 37*k2: X[k] = sum_n2 B[k1][n2] * W_N2^(n2*k2)
__device__ __forceinline__ cplx fft37_step2_bin(const cplx* B, const cplx* wN2, int N2, int ldb, int k) {
    const int k2 = k / 37, k1 = k - k2 * 37;
    double ar = 0.0, ai = 0.0;
    int idx = 0;
    const cplx* row = B + k1 * ldb;
#pragma unroll 8
    for (int n2 = 0; n2 < N2; ++n2) {
        const cplx v = row[n2], t = wN2[idx];
        ar = fma(v.x, t.x, fma(-v.y, t.y, ar));
        ai = fma(v.x, t.y, fma(v.y, t.x, ai));
        idx += k2;
        idx = idx >= N2 ? idx - N2 : idx;
    }
    return make_double2(ar, ai);
}

// step 2 for a MIRROR PAIR of bins sharing row k1: k = k1 + 37*j and k' = k1 + 37*(N2-j), 1 <= j < N2/2 --
//   X[k] = P - iQ, X[k'] = P + iQ,  P = B[0] + (-1)^j B[N2/2] + sum_{n2=1..N2/2-1} (B[n2]+B[N2-n2]) cos(2 pi n2 j/N2),
//                                   Q = sum_{n2=1..N2/2-1} (B[n2]-B[N2-n2]) sin(2 pi n2 j/N2)
// -- and for j = 0 the pair (k1, k1 + 37*N2/2): sums of the even / odd columns.  A quarter of the MACs of two
// fft37_step2_bin calls (N2 is a multiple of 4).
__device__ __forceinline__ void fft37_step2_pair(const cplx* B, const cplx* wN2, int N2, int ldb, int k1, int j,
                                                 cplx* Xa, cplx* Xb) {
    const cplx* row = B + k1 * ldb;
    const int h = N2 >> 1;
    if (j == 0) {
        double er = 0.0, ei = 0.0, orr = 0.0, oi = 0.0;
#pragma unroll 4
        for (int n2 = 0; n2 < N2; n2 += 2) {
            const cplx e = row[n2], o = row[n2 + 1];
            er += e.x; ei += e.y; orr += o.x; oi += o.y;
        }
        *Xa = make_double2(er + orr, ei + oi);
        *Xb = make_double2(er - orr, ei - oi);
        return;
    }
    const cplx b0 = row[0], bh = row[h];
    const double sg = (j & 1) ? -1.0 : 1.0;
    double pr = b0.x + sg * bh.x, pi = b0.y + sg * bh.y, qr = 0.0, qi = 0.0;
    int idx = 0;
#pragma unroll 5
    for (int n2 = 1; n2 < h; ++n2) {
        idx += j;
        idx = idx >= N2 ? idx - N2 : idx;
        const cplx u = row[n2], v = row[N2 - n2];
        const cplx t = wN2[idx];                             // (cos, -sin) of 2 pi n2 j / N2
        pr = fma(u.x + v.x, t.x, pr); pi = fma(u.y + v.y, t.x, pi);
        qr = fma(u.x - v.x, t.y, qr); qi = fma(u.y - v.y, t.y, qi);   // q = -Q
    }
    *Xa = make_double2(pr - qi, pi + qr);                    // P - iQ = P + i q
    *Xb = make_double2(pr + qi, pi - qr);
}

__global__ void k_make_twiddles(cplx* tw, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double sn, cs;
    sincospi(-2.0 * (double)i / (double)n, &sn, &cs);
    tw[i] = make_double2(cs, sn);
}

#define FFT_THREADS 512
#define FC_NB 8        /* k_fine_cert: candidate bins around the tone */
template <int MODE>
__global__ void __launch_bounds__(FFT_THREADS) k_fft_burst(const StreamState* __restrict__ sts,
                                                           const cplx* __restrict__ win, long win_stream_stride,
                                                           long win_stride, int nfft,
                                                           const cplx* __restrict__ tw_g, PeakOut* __restrict__ peaks,
                                                           cplx* __restrict__ x0, int H) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int N2 = nfft / 37;
    const int ldb = N2 + 1;                     // padded row: conflict-free column reads in step 2
    cplx* xs = (cplx*)smem;                     // nfft
    cplx* B = xs + nfft;                        // 37 * ldb
    cplx* w37 = B + 37 * ldb;                   // 37 (+3 pad)
    cplx* wN2 = w37 + 40;                       // N2
    __shared__ double red_p[FFT_THREADS / 64];
    __shared__ int red_t[FFT_THREADS / 64], red_k[FFT_THREADS / 64];
    const int s = blockIdx.y, w = blockIdx.x;
    if (w >= sts[s].n_win) return;
    const int tid = threadIdx.x;
    const cplx* x = win + (size_t)s * win_stream_stride + (size_t)w * win_stride;
    for (int i = tid; i < nfft; i += FFT_THREADS) xs[i] = x[i];
    fft37_tables(w37, wN2, N2, tid, tw_g);
    __syncthreads();
    fft37_step1_sym(xs, B, w37, tw_g, nfft, N2, ldb, tid, FFT_THREADS);
    __syncthreads();
    double best = -1.0;
    int key = 0x7fffffff, kk = 0;
    for (int k = tid; k < nfft; k += FFT_THREADS) {
        const cplx X = fft37_step2_bin(B, wN2, N2, ldb, k);
        const double ar = X.x, ai = X.y;
        if (MODE == 1) {
            x0[((size_t)s * H + w) * nfft + k] = make_double2(ar, ai);
        } else {
            const double p = ar * ar + ai * ai;
            const int sk = (k + nfft / 2) % nfft;            // position after fftshift
            if (p > best || (p == best && sk < key)) { best = p; key = sk; kk = k; }
        }
    }
    if (MODE == 1) return;
    for (int off = 32; off > 0; off >>= 1) {
        const double op = __shfl_down(best, off, 64);
        const int ok = __shfl_down(key, off, 64);
        const int okk = __shfl_down(kk, off, 64);
        if (op > best || (op == best && ok < key)) { best = op; key = ok; kk = okk; }
    }
    const int wv = tid >> 6;
    if ((tid & 63) == 0) { red_p[wv] = best; red_t[wv] = key; red_k[wv] = kk; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < FFT_THREADS / 64; ++i)
            if (red_p[i] > best || (red_p[i] == best && red_t[i] < key)) { best = red_p[i]; key = red_t[i]; kk = red_k[i]; }
        PeakOut o; o.p = best; o.tie = key; o.k = kk;
        peaks[(size_t)s * H + w] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// Fine search, FCCH_fine_correction.m:48-52: argmax over the nshift = 128*ov+1 window starts of
// max_k |FFT_nfft(window)|^2 (first maximum), evaluated as a sliding DFT
//   X_k(m+1) = (X_k(m) + x[m+nfft] - x[m]) * exp(+2*pi*i*k/nfft)
// in three kernels over 64-shift chunks (chunk c = shifts 64c+1 .. 64c+64; shift 0 rides with chunk 0):
//  1. k_fine_cert  (fp64, exact): the FC_NB bins around the tone at ALL shifts -> (P*, t*, k*), plus a
//     certificate (Parseval + recurrence slack) that no other bin can reach P* on the shifts [a, b]; what is
//     left open is a short prefix of chunks (nch), usually none.
//  2. k_fine_chunk (packed fp32): one block per OPEN (window, chunk): all-bin spectrum at the chunk's first
//     shift by the 37 x N2 FFT (fp64), then 64 steps of the recurrence in single precision for every bin,
//     two bins per lane (v_pk_*_f32).  It keeps each bin's maximum power over the chunk and hands on only the
//     bins that could still matter:  sqrt(p32) + E_c >= max(sqrt(P*), chunk maximum - E_c), where E_c bounds
//     the fp32 error of the chunk:  |X32 - X| <= (2^-24 + 64 steps * 16 ulp32) * (sum|x window| + sum|d|)
//     < 2^-13 * (...)  (triangle inequality on the unrolled recurrence; |re|+|im| bounds |z|).
//  3. k_fine_verify (fp64, exact): every surviving (bin, chunk) whose fp32 amplitude + E_c reaches
//     L = max(sqrt(P*), max_c(chunk maximum - E_c)) is re-evaluated with an fp64 anchor and the fp64
//     recurrence; the first maximum of those and (P*, t*, k*) is the result.  The pair holding the true fp64
//     maximum always passes (its fp32 amplitude is within E_c of the truth, and L never exceeds the truth),
//     so the result equals that of running fp64 on every (bin, shift).
// GSMCAL_CERT=0 opens every chunk of every window; GSMCAL_PRESCREEN=0 runs k_fine_search, the plain all-bin
// fp64 sweep, as the cross-check of the whole scheme.
// ------------------------------------------------------------------------------------------------
#define FS_CHUNK 64
#define FS_ERR_SCALE_CHUNK 0.0001220703125   /* 2^-13 */

// X_k at one shift by a direct fp64 DFT over the nfft samples xw[0..nfft), executed by one wave: lane l sums
// the terms n = l, l+64, ... (fixed order), then a shuffle tree adds the 64 partials.  The 19 table loads of a
// pass are independent and issued together (the twiddle table lives in L2).  Returns the value in every lane.
// NB: table loads in flight per pass (the order of the additions does not depend on it).  19 = the whole 8x-oversampled
// window in one pass, 76 registers; k_post_chain_r, at 80 registers for everything, takes 6.
template <int NB = 19>
__device__ __forceinline__ cplx anchor_dft(const cplx* xw, int k, const cplx* __restrict__ tw_g, int nfft, int lane) {
    double ar = 0.0, ai = 0.0;
    const int stp = (int)(((unsigned)k * 64u) % (unsigned)nfft);          // (k < nfft <= 2^24: no overflow)
    int idx = (int)(((unsigned)k * (unsigned)lane) % (unsigned)nfft);
    for (int n0 = 0; n0 < nfft; n0 += NB * 64) {
        cplx t[NB];
        int id = idx;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            t[u] = tw_g[id];
            id += stp;
            if (id >= nfft) id -= nfft;
        }
        idx = id;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int n = n0 + lane + 64 * u;
            if (n < nfft) {
                const cplx v = xw[n];
                ar = fma(v.x, t[u].x, fma(-v.y, t[u].y, ar));
                ai = fma(v.x, t[u].y, fma(v.y, t[u].x, ai));
            }
        }
    }
    return make_double2(wave_sum(ar), wave_sum(ai));
}

// result of k_fine_cert for one window (see below)
// nch open chunks: the first nch - nsuf of the window (its uncertified prefix of shifts) and the last nsuf (its uncertified suffix)
struct FineCert { double p; int t, k, a, b; int nch; int nsuf; };
// chunk id of the j-th open chunk of a window
__device__ __forceinline__ int fc_open_chunk(int j, int nch, int nsuf, int nchunk) { return j < nch - nsuf ? j : nchunk - nch + j; }

typedef float v2f __attribute__((ext_vector_type(2)));

// what k_fine_chunk hands to k_fine_verify for one (window, chunk): 256 bytes
#define FK_CAP 30
struct ChunkCand { int k; float p; };
struct ChunkRec {
    float amax;                 // largest fp32 power of the chunk (any bin)
    int count;                  // candidates listed; -1: more than FK_CAP (verify then tries every bin of the chunk)
    double E;                   // fp32 amplitude error bound of the chunk
    ChunkCand cand[FK_CAP];
};

// ------------------------------------------------------------------------------------------------
// k_fine_chunk: persistent workgroups (3 per CU, block FK_THREADS) over the list of open (window, chunk) items that
// k_fine_cert (or k_fine_openall) wrote.
// LDS: xs[nfft+64] | B[37][N2+1] | w37 (40) | wN2 | d[64] (float2).
// ------------------------------------------------------------------------------------------------
#define FK_THREADS 640
__host__ inline size_t fk_lds_bytes(int nfft) {
    const int N2 = nfft / 37;
    return ((size_t)nfft + FS_CHUNK + (size_t)37 * (N2 + 1) + 40 + N2) * sizeof(cplx) + FS_CHUNK * sizeof(float2);
}
// (Not specialised for the reference geometry like the kernels around it: with compile-time bounds the compiler unrolls
// this one into 116 B of scratch per lane and 13.3 us instead of 11.5.)
__global__ void __launch_bounds__(FK_THREADS) __attribute__((amdgpu_waves_per_eu(6, 8))) k_fine_chunk(const cplx* __restrict__ win, long win_stream_stride,
                                                          long win_stride, int nshift, int nfft,
                                                          const cplx* __restrict__ tw_g, const FineCert* __restrict__ cert,
                                                          ChunkRec* __restrict__ rec, int H,
                                                          const int* __restrict__ items, const int* __restrict__ n_items) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int N2 = nfft / 37, ldb = N2 + 1;
    cplx* xs = (cplx*)smem;                     // nfft + FS_CHUNK samples from the chunk's first shift
    cplx* B = xs + nfft + FS_CHUNK;
    cplx* w37 = B + 37 * ldb;
    cplx* wN2 = w37 + 40;
    float2* d = (float2*)(wN2 + N2);
    __shared__ double sh_e[FK_THREADS / 64];
    __shared__ float sh_m[FK_THREADS / 64];
    __shared__ int sh_cnt;
    __shared__ double sh_D;                     // sum of |d_q| over the chunk's steps: how far any bin's amplitude can move
    const int tid = threadIdx.x;
    const int nstep = nshift - 1;
    const int nchunk = (nstep + FS_CHUNK - 1) / FS_CHUNK;
    const int total = *n_items;
    fft37_tables(w37, wN2, N2, tid, tw_g);
    DEV_STAMP(KID_CHUNK, blockIdx.x, 0);
    for (int it = blockIdx.x; it < total; it += gridDim.x) {   // open (window, chunk) items, block-uniform
    const int item = items[it];
    const int c = item & 0xFF, widx = item >> 8, s = widx / H, w = widx - s * H;
    const double pstar = cert && cert[widx].p > 0.0 ? cert[widx].p : 0.0;
    const int t0 = c * FS_CHUNK;
    const int lim = nstep - t0 < FS_CHUNK ? nstep - t0 : FS_CHUNK;
    const cplx* x = win + (size_t)s * win_stream_stride + (size_t)w * win_stride + t0;
    __syncthreads();                            // (LDS of the previous item is free)
    for (int i = tid; i < nfft + lim; i += FK_THREADS) xs[i] = x[i];
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    double e = 0.0;                             // error-bound sum: |re|+|im| of the anchor window and of the steps
    for (int i = tid; i < nfft; i += FK_THREADS) e += fabs(xs[i].x) + fabs(xs[i].y);
    if (tid < FS_CHUNK) {
        float2 v = make_float2(0.0f, 0.0f);     // (zero-padded steps only rotate X: |X| unchanged)
        double dabs = 0.0;
        if (tid < lim) {
            const cplx a1 = xs[tid + nfft], b1 = xs[tid];
            const double dr = a1.x - b1.x, di = a1.y - b1.y;
            v = make_float2((float)dr, (float)di);
            dabs = fabs(dr) + fabs(di);         // >= |d|
            e += dabs;
        }
        d[tid] = v;
        dabs = wave_sum(dabs);                  // (FS_CHUNK = 64: exactly the first wave)
        if (tid == 0) sh_D = dabs;
    }
    for (int off = 32; off > 0; off >>= 1) e += __shfl_down(e, off, 64);
    if ((tid & 63) == 0) sh_e[tid >> 6] = e;
    __syncthreads();                            // (xs is rewritten in place from here on)
    fft37_step1_sym(xs, B, w37, tw_g, nfft, N2, ldb, tid, FK_THREADS);
    __syncthreads();
    double es = 0.0;
    for (int i = 0; i < FK_THREADS / 64; ++i) es += sh_e[i];
    const double E = FS_ERR_SCALE_CHUNK * es;
    // ---- sweep: two bins per lane (one pass whenever nfft <= 2*FK_THREADS) ----
    const int npair = nfft >> 1;                             // nfft = 148*ov is even
    const double sp = sqrt(pstar);
    ChunkRec* r = rec + ((size_t)s * H + w) * nchunk + c;
    float m = 0.0f;
    for (int g0 = 0; g0 < npair; g0 += FK_THREADS) {         // block-uniform trip count
        const int g = g0 + tid;
        const bool act = g < npair;
        // lane g owns a mirror pair of bins (any pairing serves the sweep; this one halves the second FFT stage)
        const int jj = g / 37, kr = g - jj * 37;
        const int k0 = kr + 37 * jj, k1 = jj == 0 ? kr + 37 * (N2 >> 1) : kr + 37 * (N2 - jj);
        v2f best = {0.0f, 0.0f};
        if (act) {
            cplx X0, X1;
            fft37_step2_pair(B, wN2, N2, ldb, kr, jj, &X0, &X1);
            const cplx t0w = tw_g[k0], t1w = tw_g[k1];       // exp(-2 pi i k/nfft): the recurrence turns by the conjugate
            const v2f wr = {(float)t0w.x, (float)t1w.x}, wi = {(float)-t0w.y, (float)-t1w.y};
            v2f xr = {(float)X0.x, (float)X1.x}, xi = {(float)X0.y, (float)X1.y};
            if (c == 0) best = xr * xr + xi * xi;            // the start window m = 0 rides with chunk 0
            // |X_k(t+1)| = |X_k(t) + d_t| <= |X_k(t)| + |d_t|: a bin that starts the chunk more than D = sum |d_q| below
            // sqrt(P*) stays below it on all 64 shifts -- it cannot hold the window's maximum (P* is attained) and is not
            // swept.  In the edge chunks the certificate leaves open that is every bin but the few around the tone, so
            // whole waves skip the loop; an unswept bin counts as 0 in the chunk maximum, which only lowers the bars
            // (thr below, L in k_fine_verify) -- the safe direction.
            const double D = sh_D * (1.0 + 1e-9);
            const bool sweep = !(pstar > 0.0) || !((sqrt(X0.x * X0.x + X0.y * X0.y) + D) * (1.0 + 1e-9) < sp) ||
                               !((sqrt(X1.x * X1.x + X1.y * X1.y) + D) * (1.0 + 1e-9) < sp);
            if (sweep) {
#pragma unroll 8
            for (int t = 0; t < FS_CHUNK; ++t) {
                const float2 dv = d[t];
                const v2f dr = {dv.x, dv.x}, di = {dv.y, dv.y};
                const v2f ar = xr + dr, ai = xi + di;
                xr = ar * wr - ai * wi;
                xi = ar * wi + ai * wr;
                const v2f p = xr * xr + xi * xi;
                best.x = fmaxf(best.x, p.x);
                best.y = fmaxf(best.y, p.y);
            }
            }
            if (k1 == k0) best.y = 0.0f;
            m = fmaxf(m, fmaxf(best.x, best.y));
        }
        // the largest power seen by the block so far: the listing test below needs it (a later pass can only raise
        // it, so testing against the running value is a weaker, still necessary, condition)
        float mm = m;
        for (int off = 32; off > 0; off >>= 1) mm = fmaxf(mm, __shfl_xor(mm, off, 64));
        __syncthreads();
        if ((tid & 63) == 0) sh_m[tid >> 6] = mm;
        __syncthreads();
        float bm = sh_m[0];
        for (int i = 1; i < FK_THREADS / 64; ++i) bm = fmaxf(bm, sh_m[i]);
        // a pair can only matter if sqrt(p) + E >= max(sqrt(P*), sqrt(bm) - E)
        double thr = sqrt((double)bm) - E;
        if (sp > thr) thr = sp;
        thr = (thr - E) * (1.0 - 1e-6);
        if (act && sqrt((double)best.x) >= thr) {
            const int i = atomicAdd(&sh_cnt, 1);
            if (i < FK_CAP) { r->cand[i].k = k0; r->cand[i].p = best.x; }
        }
        if (act && k1 != k0 && sqrt((double)best.y) >= thr) {
            const int i = atomicAdd(&sh_cnt, 1);
            if (i < FK_CAP) { r->cand[i].k = k1; r->cand[i].p = best.y; }
        }
    }
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((tid & 63) == 0) sh_m[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        float bm = sh_m[0];
        for (int i = 1; i < FK_THREADS / 64; ++i) bm = fmaxf(bm, sh_m[i]);
        r->amax = bm;
        r->E = E;
        r->count = sh_cnt <= FK_CAP ? sh_cnt : -1;
    }
    DEV_STAMP(KID_CHUNK, blockIdx.x, 1 + (it / gridDim.x < 6 ? it / gridDim.x : 6));
    }
}

// GSMCAL_CERT=0 (or a geometry the certificate does not cover): every chunk of every window is open.
// grid (H, S), block 64.
__global__ void k_fine_openall(const StreamState* __restrict__ sts, int nchunk, int H, int* __restrict__ items,
                               int* __restrict__ n_items) {
    const int s = blockIdx.y, w = blockIdx.x;
    if (w >= sts[s].n_win) return;
    __shared__ int base;
    if (threadIdx.x == 0) base = atomicAdd(n_items, nchunk);
    __syncthreads();
    for (int c = threadIdx.x; c < nchunk; c += blockDim.x) items[base + c] = ((s * H + w) << 8) | c;
}

// ------------------------------------------------------------------------------------------------
// k_fine_verify: the exact (fp64) last word of the fine search.  grid (H, S), block 256.
// Work items = the (bin, chunk) pairs k_fine_chunk listed and that reach the window-wide bar L.  For each item the
// spectrum value at the chunk's first shift is anchored by a direct fp64 DFT (all 64 lanes of a wave, fixed
// summation order, table twiddles), then one lane slides through the chunk's 64 shifts with the fp64 recurrence
// tracking (max power, first shift).  The block reduces with MATLAB's first-max rule (larger power, then smaller
// shift, then smaller bin), the certificate's (P*, t*, k*) competing, into one PeakOut per window.
// LDS: the window | item anchors | item list.
// ------------------------------------------------------------------------------------------------
#define FV_MAX_ITEMS 512
#define FV_THREADS 256
// (agent-scope stores: the workgroup that runs the stream's decision step in this same kernel reads them)
__device__ __forceinline__ void peak_store(PeakOut* dst, const PeakOut& o) {
    static_assert(sizeof(PeakOut) == 16, "two 64-bit words");
    unsigned long long* d = (unsigned long long*)dst;
    __hip_atomic_store(d, (unsigned long long)__double_as_longlong(o.p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d + 1, (unsigned long long)(unsigned)o.tie | ((unsigned long long)(unsigned)o.k << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
template <int FVT, int ANB = 19>
__device__ __forceinline__ void fine_verify_body(const StreamState* __restrict__ sts,
                                                 const cplx* __restrict__ win, long win_stream_stride,
                                                 long win_stride, int nshift, int nfft,
                                                 const cplx* __restrict__ tw_g, const ChunkRec* __restrict__ rec,
                                                 PeakOut* __restrict__ out, int H,
                                                 const FineCert* __restrict__ cert, int* __restrict__ n_open,
                                                 unsigned char* smem, PeakOut* res = nullptr) {   // res: the window's peak instead of out[]
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *n_open = 0;   // k_fine_chunk is done with the list: clear it for the next batch
    const int wlen = nshift - 1 + nfft;
    cplx* xs = (cplx*)smem;                               // window
    cplx* anchor = xs + wlen;                             // X_k at the chunk's first shift, per item
    int* items = (int*)(anchor + FV_MAX_ITEMS);           // (k << 8) | chunk
    __shared__ int n_items, n_over;
    __shared__ double red_p[FVT / 64];
    __shared__ int red_t[FVT / 64], red_k[FVT / 64];
    const int s = blockIdx.y, w = blockIdx.x;
    if (w >= sts[s].n_win) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nstep = nshift - 1;
    const int nchunk = (nstep + FS_CHUNK - 1) / FS_CHUNK;
    FineCert fc;
    fc.p = -1.0; fc.t = 0x7fffffff; fc.k = 0x7fffffff; fc.nch = nchunk; fc.nsuf = 0;
    if (cert) {
        fc = cert[(size_t)s * H + w];
        if (fc.nch == 0) {                                // fully certified window: the certificate IS the answer
            if (tid == 0) { PeakOut o2; o2.p = fc.p; o2.tie = fc.t; o2.k = fc.k; if (res) *res = o2; else peak_store(&out[(size_t)s * H + w], o2); }
            return;
        }
    }
    const int nopen = fc.nch, nsuf = fc.nsuf;             // open chunk j has id fc_open_chunk(j, nopen, nsuf, nchunk); its record sits in slot j
    // the open chunks' records (256 B each) come into LDS with one round of coalesced loads; the window's place is free
    ChunkRec* r = (ChunkRec*)smem;
    {
        const uint4* src = (const uint4*)(rec + ((size_t)s * H + w) * nchunk);
        uint4* dst = (uint4*)r;
        constexpr int Q = (int)(sizeof(ChunkRec) / 16);
        for (int i = tid; i < nopen * Q; i += FVT) { const int j = i / Q; dst[i] = src[fc_open_chunk(j, nopen, nsuf, nchunk) * Q + (i - j * Q)]; }
    }
    if (tid == 0) { n_items = 0; n_over = 0; }
    __syncthreads();
    // the bar a pair must reach: the certificate's exact maximum, or the best guaranteed amplitude of any open chunk
    double L = fc.p > 0.0 ? sqrt(fc.p) : 0.0;
    for (int c = 0; c < nopen; ++c) {
        const double v = sqrt((double)r[c].amax) - r[c].E;
        if (v > L) L = v;
    }
    L *= 1.0 - 1e-9;
    for (int i = tid; i < nopen * FK_CAP; i += FVT) {     // listed candidates of all open chunks
        const int c = i / FK_CAP, q = i - c * FK_CAP;
        const int cnt = r[c].count;
        if (cnt < 0) { if (q == 0) atomicAdd(&n_over, 1); continue; }
        if (q >= cnt) continue;
        const ChunkCand cc = r[c].cand[q];
        if (sqrt((double)cc.p) + r[c].E >= L) {
            const int idx = atomicAdd(&n_items, 1);
            if (idx < FV_MAX_ITEMS) items[idx] = (cc.k << 8) | fc_open_chunk(c, nopen, nsuf, nchunk);
        }
    }
    __syncthreads();
    const bool slow = n_over > 0 || n_items > FV_MAX_ITEMS;   // block-uniform; pathological inputs only (e.g. all zeros)
    DEV_STAMP(KID_VERIFY, blockIdx.y * gridDim.x + blockIdx.x, 3);
    if (n_items == 0 && !slow && fc.p > 0.0) {            // (block-uniform) nothing reaches the bar: the certificate's answer stands
        if (tid == 0) { PeakOut o2; o2.p = fc.p; o2.tie = fc.t; o2.k = fc.k; if (res) *res = o2; else peak_store(&out[(size_t)s * H + w], o2); }
        return;
    }
    const cplx* x = win + (size_t)s * win_stream_stride + (size_t)w * win_stride;
    double best = -1.0;
    int bt = 0x7fffffff, bk = 0x7fffffff;
    if (tid == 0 && fc.p > 0.0) { best = fc.p; bt = fc.t; bk = fc.k; }   // the certificate's bins compete with the rest
    if (n_items > 0 || slow) {
        for (int i = tid; i < wlen; i += FVT) xs[i] = x[i];
        __syncthreads();
        // normal case: one round over the listed items.  slow case: every (bin, open chunk) pair, FV_MAX_ITEMS at a time
        const long total = slow ? (long)nopen * nfft : (long)n_items;
        for (long base = 0; base < total; base += FV_MAX_ITEMS) {
            const int ni = (int)(total - base < FV_MAX_ITEMS ? total - base : FV_MAX_ITEMS);
            if (slow) {
                __syncthreads();
                for (int i = tid; i < ni; i += FVT) {
                    const long g = base + i;
                    items[i] = ((int)(g % nfft) << 8) | fc_open_chunk((int)(g / nfft), nopen, nsuf, nchunk);
                }
                __syncthreads();
            }
            // ---- one wave per item: anchor X_k at the chunk's first shift by a wave-parallel direct DFT, then all 64
            // shifts of the chunk at once.  Unrolling the recurrence, X_k(t0+m) = W^(-km) [X_k(t0) + S_m] with
            // S_m = sum_{q<m} W^(kq) d_q, d_q = x[t0+q+nfft] - x[t0+q] (W = exp(-2 pi i/nfft)): a wave-wide prefix sum, and
            // the rotation drops out of the power |X_k(t0) + S_m|^2.
            for (int i = wave; i < ni; i += FVT / 64) {
                const int k = items[i] >> 8, c = items[i] & 0xFF;
                const int t0 = c * FS_CHUNK;
                const cplx av = anchor_dft<ANB>(xs + t0, k, tw_g, nfft, lane);
                const double a0r = av.x, a0i = av.y;
                const int lim = nstep - t0 < FS_CHUNK ? nstep - t0 : FS_CHUNK;
                double sr = 0.0, si = 0.0;
                if (lane < lim) {
                    const cplx xa = xs[t0 + lane + nfft], xb = xs[t0 + lane];
                    const cplx t = tw_g[((unsigned)k * (unsigned)lane) % (unsigned)nfft];
                    const double dr = xa.x - xb.x, di = xa.y - xb.y;
                    sr = dr * t.x - di * t.y;
                    si = dr * t.y + di * t.x;
                }
                sr = wave_scan_incl(sr);                                 // inclusive scan: lane m ends with S_(m+1)
                si = wave_scan_incl(si);
                if (lane < lim) {
                    const double xr = a0r + sr, xi = a0i + si;
                    const double p = xr * xr + xi * xi;
                    const int m = t0 + lane + 1;
                    if (p > best || (p == best && (m < bt || (m == bt && k < bk)))) { best = p; bt = m; bk = k; }
                }
                if (c == 0 && lane == 0) {                               // the start window m = 0 belongs to chunk 0
                    const double p = a0r * a0r + a0i * a0i;
                    if (p > best || (p == best && (0 < bt || (0 == bt && k < bk)))) { best = p; bt = 0; bk = k; }
                }
            }
        }
    }
    DEV_STAMP(KID_VERIFY, blockIdx.y * gridDim.x + blockIdx.x, 5);
#ifdef GSMCAL_DEVTIMING
    if (g_stamps && tid == 0 && blockIdx.y * gridDim.x + blockIdx.x < DEV_STAMP_BLOCKS) g_stamps[((size_t)KID_VERIFY * DEV_STAMP_BLOCKS + blockIdx.y * gridDim.x + blockIdx.x) * 16 + 15] = (unsigned long long)n_items * 100ull + (unsigned long long)nopen * 10000000ull;
#endif
    for (int off = 32; off > 0; off >>= 1) {
        const double op = __shfl_down(best, off, 64);
        const int ot = __shfl_down(bt, off, 64);
        const int ok = __shfl_down(bk, off, 64);
        if (op > best || (op == best && (ot < bt || (ot == bt && ok < bk)))) { best = op; bt = ot; bk = ok; }
    }
    if (lane == 0) { red_p[wave] = best; red_t[wave] = bt; red_k[wave] = bk; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < FVT / 64; ++i)
            if (red_p[i] > best || (red_p[i] == best && (red_t[i] < bt || (red_t[i] == bt && red_k[i] < bk)))) {
                best = red_p[i]; bt = red_t[i]; bk = red_k[i];
            }
        PeakOut o2; o2.p = best; o2.tie = bt; o2.k = bk;
        if (res) *res = o2; else peak_store(&out[(size_t)s * H + w], o2);
    }
}

// ------------------------------------------------------------------------------------------------
// k_fine_cert: Parseval certificate for the fine search.  grid (H, S), block 8*nchunk rounded up to whole
// waves (nchunk = nstep/64).
// An FCCH window is dominated by one tone, so the maximum over (shift, bin) is almost always in the few
// bins around the tone.  With S = the FC_NB bins around the strongest bin of the start spectrum x0:
//   1. every bin of S is evaluated exactly (fp64) at ALL shifts: lane (c, j) anchors X_k(64c) and slides
//      through the 64 shifts of chunk c with the recurrence, tracking (max power, first shift) -> P*, t*, k*;
//   2. every OTHER bin at shift t is bounded by Parseval:  P_other(t) <= R(t) = N*E(t) - sum_{k in S} P(t,k),
//      E(t) = energy of the window starting at t;
//   3. shift t is "certified" when R(t) (plus the rounding margin of the fp32 group sums) < P*.
// The certified shifts form a range [a, b] around t*; what is left for the all-bin search is the prefix
// [0, a) -- and, rarely, a suffix, in which case everything is searched.  Output: (P*, t*, k*), a, b and
// nch = number of leading 64-shift chunks k_fine_chunk still has to sweep (0: the window is settled).
// Anchors by a two-level regrouping of the DFT sum (exact algebra, fp64): with B = gcd(64, nfft),
//   S_k(m) = sum_{b<B} x[B*m+b] W^(k*b),   X_k(64c) = sum_{a<nfft/B} W^(k*B*a) * S_k(64c/B + a),   W = exp(-2 pi i/nfft)
// i.e. nfft/B + (wlen/B)*B/nchunk MACs per anchor instead of nfft.
// LDS: window | S partials (later E[nstep+1]) | twiddle tables (later the fp32 group sums).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fc_group8_sum(float v) {     // sum over the 8 lanes of a bin group, in every lane
    int t;
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v += __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v += __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true);  // row_half_mirror: the other quad
    v += __int_as_float(t);
    return v;
}

__host__ __device__ inline int fc_gcd64(int n) { int b = 64; while (n % b) b >>= 1; return b; }
// threads of k_fine_cert: 8 per chunk, at least 256.  (320 would save the nearly empty third round of the level-1 phase --
// 69 blocks handed out 32 at a time -- but five-wave workgroups no longer sit three to a CU: measured 77 us against 59.)
__host__ __device__ constexpr int fc_threads(int nshift) {
    return ((nshift - 1) / FS_CHUNK * FC_NB + 63) / 64 * 64 < 256 ? 256 : ((nshift - 1) / FS_CHUNK * FC_NB + 63) / 64 * 64;
}
// One element of padding after every 128: lane groups that work 64 shifts apart (the chunks of the slide phase) would
// otherwise all hit the same LDS banks (8-way; the SQ counters showed 4 bank-conflict cycles per LDS instruction in this
// kernel); this leaves them 2-way.  (Padding every 64 would clear them entirely but costs the third workgroup per CU: the
// kernel sits 100 bytes under the 53 760-byte line.)
#define FC_XP(p) ((p) + ((p) >> 7))
// k_fine_cert building its window from the raw bytes itself (raw == nullptr: read it from `win`)
struct FusedGather {
    const uint8_t* raw; long raw_stride;
    const double* coef;
    cplx* win_out;
    int ntaps, per;           // per: outputs per staging pass (multiple of 4)
    int sym47, pad;           // sym47: 47 exactly mirrored taps (the drivers' fir1(46)): taps in registers, every sample read once (fir4_sym47)
};
// staging bytes of one pass: padded complex input | coefficients
__host__ inline size_t fc_stage_bytes(int per, int ntaps) {
    const int span8 = per + ntaps + 8;
    return (size_t)(span8 + span8 / 4 + 1) * sizeof(cplx) + (size_t)fir_taps_padded(ntaps) * sizeof(double);
}
__host__ inline size_t fc_lds_bytes(int nshift, int nfft) {
    const int nstep = nshift - 1, wlen = nstep + nfft, B = fc_gcd64(nfft);
    size_t r1 = (size_t)FC_NB * (wlen / B) * sizeof(cplx), e1 = (size_t)(nstep + 2) * sizeof(double);
    size_t r2 = (size_t)FC_NB * (B + 2 + nfft / B) * sizeof(cplx), e2 = (size_t)2 * (size_t)(FC_XP(nstep + 2) + 1) * sizeof(float) + 16 * sizeof(double);
    return (size_t)FC_XP(wlen) * sizeof(cplx) + (r1 > e1 ? r1 : e1) + (r2 > e2 ? r2 : e2);
}

// Issue priority of this wave by the dispatch round of its workgroup (0..255 / 256..511 / 512..: first, second, third on its
// CU).  The SIMDs issue oldest wave first: left alone, the first workgroup of a CU finishes at ~35 us, the third at 55, alone on
// the CU for its last 10 us -- and the kernel ends with it.  Turning the order round between phases keeps the three closer
// together and the CU full to the end: 55.3 -> 49.7 us (schedules tried, by phase group build | S + anchors | E(t) + slides +
// certificate, E = oldest first, L = youngest first: ELE 54.1, LLL 50.7, ELL 50.9, LNL 50.0, LELEL 49.6, LEL 49.5).
// (Only while the whole grid is resident, three workgroups per CU: a longer grid is dispatched continuously, there is no "round",
// and the same switches cost the 1 024-stream step 1.3 %.)
#define FC_PRIO(P0, P1, P2) if (gridDim.x * gridDim.y <= 768u) { const unsigned rnd_ = (blockIdx.y * gridDim.x + blockIdx.x) >> 8; if (rnd_ == 0) __builtin_amdgcn_s_setprio(P0); else if (rnd_ == 1) __builtin_amdgcn_s_setprio(P1); else __builtin_amdgcn_s_setprio(P2); }
#define FC_PF 8   /* slide steps whose samples are fetched from LDS ahead of the arithmetic */
// OV > 0: the reference geometry as compile-time constants (128*OV+1 shifts, 148*OV-point windows, NTAPS filter taps);
// OV = 0: from the arguments.
template <int OV, int NTAPS>
__global__ void __launch_bounds__(512) k_fine_cert(const StreamState* __restrict__ sts,
                                                   const cplx* __restrict__ win, long win_stream_stride,
                                                   long win_stride, int nshift_rt, int nfft_rt,
                                                   const cplx* __restrict__ tw_g, FineCert* __restrict__ cert, int H,
                                                   int* __restrict__ items, int* __restrict__ n_items, FusedGather fg_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nshift = OV > 0 ? 128 * OV + 1 : nshift_rt, nfft = OV > 0 ? 148 * OV : nfft_rt;
    FusedGather fg = fg_in;
    if (OV > 0) fg.ntaps = NTAPS;
    const int nstep = nshift - 1, wlen = nstep + nfft;
    const int B = fc_gcd64(nfft), nA = nfft / B, nM = wlen / B, nchunk = nstep / FS_CHUNK;
    const int nthr = blockDim.x;                          // >= 8 * nchunk, whole waves (as a compile-time constant: 1.2 us SLOWER)
    cplx* xs = (cplx*)smem;                               // window
    const size_t r1 = (size_t)FC_NB * nM * sizeof(cplx), e1 = (size_t)(nstep + 2) * sizeof(double);
    cplx* Sp = xs + FC_XP(wlen);                          // S partials [j][m]
    double* Et = (double*)Sp;                             // ... later E[t], t = 0..nstep
    unsigned char* reg2 = (unsigned char*)Sp + (r1 > e1 ? r1 : e1);
    const int ld1 = B + 2;                                // plane stride of tw1: spreads the 8 bins over the banks
    cplx* tw1 = (cplx*)reg2;                              // [j][b]  W^(k_j b)
    cplx* tw2 = tw1 + FC_NB * ld1;                        // [a][j]  W^(k_j B a)
    float* sumS = (float*)reg2;                           // ... later sum over S of P(t,k), t = 0..nstep
    float* Cs = sumS + FC_XP(nstep + 2);                  // ... and C(t) = sum_{q<t} |d_q|, the slack of the recurrence
    double* sh_scan = (double*)(Cs + FC_XP(nstep + 2) + (FC_XP(nstep + 2) & 1));   // 2 x 8 scan partials behind them (same phases)
    double* sh_scan2 = sh_scan + 8;
    __shared__ double red_p[8];
    __shared__ int red_t[8], red_k[8];
    __shared__ int sh_a, sh_b, sh_lo;
    const int s = blockIdx.y, w = blockIdx.x;
    if (w >= sts[s].n_win) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
    DEV_STAMP(KID_CERT, blockIdx.y * gridDim.x + blockIdx.x, 0);
    FC_PRIO(0, 1, 2)                                      // window build: youngest workgroup of the CU first (see FC_PRIO)
    if (fg.raw) {
        // The window straight from the raw bytes (raw2iq.m:6-8 + filter(coef,1,.), gather_core's level 0 with the same
        // order of operations), in passes of fg.per outputs staged through regions 1/2 (free until the tone estimate):
        // the window never makes the trip through HBM before the first use, and the k_gather launch in front of this
        // kernel is gone.  It is still written out for k_fine_chunk / k_fine_verify.
        const StreamState* st = sts + s;
        const long n0 = st->n0, ws = st->win_start[w];
        const double mr = st->mean_re, mi = st->mean_im;
        const int ntp = fg.ntaps, per = fg.per;
        cplx* xq = (cplx*)Sp;
        double* c_s = (double*)(xq + xs_pad(per + ntp + 8) + 1);
        const unsigned short* base = (const unsigned short*)(fg.raw + (size_t)s * fg.raw_stride);
        cplx* wout = fg.win_out + (size_t)s * win_stream_stride + (size_t)w * win_stride;
        fir_stage_taps(c_s, fg.coef, ntp, tid, nthr);
        const bool sym47 = OV > 0 && NTAPS == 47 && fg.sym47 != 0;      // (block-uniform)
        double cf[24];                                                // the 24 distinct taps (uniform addresses: scalar loads)
        if (OV > 0 && NTAPS == 47) {
#pragma unroll
            for (int k = 0; k < 24; ++k) cf[k] = fg.coef[k];
        }
        // raw chunk `tid` of a pass (8 samples, 16-byte aligned), fetched one pass ahead so the loads overlap the FIR of
        // the previous one; samples outside the stream read as 0 and are zeroed again after the mean is subtracted
        const long ao = (long)(((uintptr_t)base >> 1) & 7);
        auto raw_chunk = [&](long first, int span, long& first_al) -> uint4 {
            long m = (first + ao) % 8;
            if (m < 0) m += 8;
            first_al = first - m;
            uint4 v = make_uint4(0, 0, 0, 0);
            const long g0 = first_al + 8L * tid;
            if (g0 < first + span) {
                if (g0 >= 0 && g0 + 8 <= n0) {
                    v = *(const uint4*)(base + g0);
                } else {
                    unsigned short t[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) t[i] = (g0 + i >= 0 && g0 + i < n0) ? base[g0 + i] : (unsigned short)0;
                    v.x = t[0] | ((unsigned)t[1] << 16); v.y = t[2] | ((unsigned)t[3] << 16);
                    v.z = t[4] | ((unsigned)t[5] << 16); v.w = t[6] | ((unsigned)t[7] << 16);
                }
            }
            return v;
        };
        long first_al;
        uint4 pre = raw_chunk(ws - (ntp - 1), (per < wlen ? per : wlen) + ntp - 1, first_al);
        for (int o0 = 0; o0 < wlen; o0 += per) {
            const int cnt = wlen - o0 < per ? wlen - o0 : per;
            const long first = ws + o0 - (ntp - 1);
            const int span = cnt + ntp - 1;
            {   // raw2iq.m:6-8 from the registers: staged sample i = 8*tid + u - (first - first_al); entries span .. span+7 are 0
                const int i_base = 8 * tid - (int)(first - first_al);
                const unsigned wv[4] = {pre.x, pre.y, pre.z, pre.w};
                const long g_base = first + i_base;
                if (i_base >= 0 && i_base + 8 <= span && g_base >= 0 && g_base + 8 <= n0) {
                    // the usual chunk: all eight samples inside the span and the stream -- no per-sample tests, and two
                    // 16-byte-aligned groups of four padded slots (i_base is a multiple of 8 here only when first is aligned, so
                    // the stores stay per sample; the tests were most of this loop)
                    // (xs_pad(i_base + u) = xs_pad(i_base) + u + ((r + u) >> 2), r = i_base & 3 the same in every lane: one lane
                    // address plus scalar offsets, as in k_stream_tile_s47)
                    const int r = __builtin_amdgcn_readfirstlane(i_base & 3);
                    cplx* xp = xq + xs_pad(i_base);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const unsigned q = wv[u >> 1] >> (16 * (u & 1));
                        xp[u + ((r + u) >> 2)] = make_double2((double)(q & 0xFFu) - mr, (double)((q >> 8) & 0xFFu) - mi);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i_base + u;
                        const long g = first + i;
                        const unsigned q = (wv[u >> 1] >> (16 * (u & 1))) & 0xFFFFu;
                        cplx v = make_double2(0.0, 0.0);
                        if (i < span && g >= 0 && g < n0) v = make_double2((double)(q & 0xFF) - mr, (double)(q >> 8) - mi);
                        if (i >= 0 && i < span + 8) xq[xs_pad(i)] = v;
                    }
                }
            }
            __syncthreads();
            if (o0 == 0) DEV_STAMP(KID_CERT, blockIdx.y * gridDim.x + blockIdx.x, 8);
            if (o0 + per < wlen) {
                const int ncnt = wlen - (o0 + per) < per ? wlen - (o0 + per) : per;
                pre = raw_chunk(first + per, ncnt + ntp - 1, first_al);
            }
            for (int i0 = 4 * tid; i0 < cnt; i0 += 4 * nthr) {
                cplx y0, y1, y2, y3;              // gather_core's loop: every accumulator takes its taps oldest first
                if (OV > 0 && NTAPS == 47 && sym47) {   // same sums, bit for bit: 50 LDS reads per four outputs instead of 72
                    cplx y[4];
                    fir4_sym47(xq, cf, i0, y);
                    y0 = y[0]; y1 = y[1]; y2 = y[2]; y3 = y[3];
                } else if (ntp == 47) fir4_lds<47>(xq, c_s, i0, ntp, &y0, &y1, &y2, &y3);
                else fir4_lds<0>(xq, c_s, i0, ntp, &y0, &y1, &y2, &y3);
                const int o = o0 + i0;
#ifdef GSMCAL_AB_LAZY_WIN      /* A/B build of tools/ab_traffic.sh only: the window goes out at the end, and only if the certificate left chunks open */
                xs[FC_XP(o)] = y0;
                if (i0 + 1 < cnt) xs[FC_XP(o + 1)] = y1;
                if (i0 + 2 < cnt) xs[FC_XP(o + 2)] = y2;
                if (i0 + 3 < cnt) xs[FC_XP(o + 3)] = y3;
#else
                xs[FC_XP(o)] = y0; wout[o] = y0;
                if (i0 + 1 < cnt) { xs[FC_XP(o + 1)] = y1; wout[o + 1] = y1; }
                if (i0 + 2 < cnt) { xs[FC_XP(o + 2)] = y2; wout[o + 2] = y2; }
                if (i0 + 3 < cnt) { xs[FC_XP(o + 3)] = y3; wout[o + 3] = y3; }
#endif
            }
            __syncthreads();
            DEV_STAMP(KID_CERT, blockIdx.y * gridDim.x + blockIdx.x, 9 + (o0 > 0) + (o0 > per));
        }
        if (tid == 0) { sh_a = 0; sh_b = nstep; }
        __syncthreads();
    } else {
        const cplx* x = win + (size_t)s * win_stream_stride + (size_t)w * win_stride;
        for (int i = tid; i < wlen; i += nthr) xs[FC_XP(i)] = x[i];
        if (tid == 0) { sh_a = 0; sh_b = nstep; }
        __syncthreads();
    }
    DEV_STAMP(KID_CERT, blockIdx.y * gridDim.x + blockIdx.x, 1);
    FC_PRIO(2, 1, 0)                                      // choice of S and the anchors: oldest first
    // ---- S: the tone's bin from a 148-point spectrum of the window's middle nfft samples summed in groups of
    // ov (nfft = 148*ov, so the two frequency grids coincide; the channel filter keeps the signal inside the
    // decimated band).  Only the choice of S depends on this estimate, never the result: a poor choice just
    // certifies less.
    {
        const int D = nfft / 148, n0 = nstep / 2;
        cplx* yd = Sp;                                    // region 1 is free until level 1
        cplx* t148 = Sp + 148;
        double* pw = (double*)(Sp + 296);
        for (int m = tid; m < 148; m += nthr) {
            double sr = 0.0, si = 0.0;
            for (int i = 0; i < D; ++i) { const cplx v = xs[FC_XP(n0 + D * m + i)]; sr += v.x; si += v.y; }
            yd[m] = make_double2(sr, si);
            t148[m] = tw_g[D * m];
        }
        __syncthreads();
        // 148 = 4 x 37: Y[k1 + 37 k2] = sum_{n2<4} W148^(n2 (k1+37 k2)) * sum_{n1<37} y[4 n1 + n2] W37^(n1 k1)
        cplx* A = (cplx*)(pw + 148);                      // [n2][k1], 148 entries
        for (int o = tid; o < 148; o += nthr) {
            const int n2 = o / 37, k1 = o - n2 * 37;
            double ar = 0.0, ai = 0.0;
            int idx = 0;
#pragma unroll 4
            for (int n1 = 0; n1 < 37; ++n1) {
                const cplx v = yd[4 * n1 + n2], t = t148[idx];   // W37^m = W148^(4m)
                ar = fma(v.x, t.x, fma(-v.y, t.y, ar));
                ai = fma(v.x, t.y, fma(v.y, t.x, ai));
                idx += 4 * k1;
                idx = idx >= 148 ? idx - 148 : idx;
            }
            A[o] = make_double2(ar, ai);
        }
        __syncthreads();
        for (int q = tid; q < 148; q += nthr) {
            const int k1 = q % 37;
            double ar = 0.0, ai = 0.0;
            int idx = 0;
#pragma unroll
            for (int n2 = 0; n2 < 4; ++n2) {
                const cplx v = A[n2 * 37 + k1], t = t148[idx];
                ar = fma(v.x, t.x, fma(-v.y, t.y, ar));
                ai = fma(v.x, t.y, fma(v.y, t.x, ai));
                idx += q;
                idx = idx >= 148 ? idx - 148 : idx;
            }
            pw[q] = ar * ar + ai * ai;
        }
        __syncthreads();
        if (wave == 0) {
            double bp = -1.0;
            int bq = 0;
            for (int q = lane; q < 148; q += 64)
                if (pw[q] > bp) { bp = pw[q]; bq = q; }
            for (int off = 32; off > 0; off >>= 1) {
                const double op = __shfl_down(bp, off, 64);
                const int oq = __shfl_down(bq, off, 64);
                if (op > bp || (op == bp && oq < bq)) { bp = op; bq = oq; }
            }
            if (lane == 0) {
                const bool up = pw[(bq + 1) % 148] > pw[(bq + 147) % 148];
                const int kpk = bq < 74 ? bq : nfft - 148 + bq;
                sh_lo = kpk - (up ? FC_NB / 2 - 1 : FC_NB / 2);   // bins lo .. lo+FC_NB-1 (mod nfft), leaning to the stronger side
            }
        }
        __syncthreads();
    }
    const int lo = sh_lo;
    DEV_STAMP(KID_CERT, blockIdx.y * gridDim.x + blockIdx.x, 2);
    const int j = tid & (FC_NB - 1), c = tid / FC_NB;
    const bool act = c < nchunk;                          // lanes beyond the last chunk only help with the shared phases
    const int k = ((lo + j) % nfft + nfft) % nfft;
    for (int i = tid; i < FC_NB * (B + nA); i += nthr) {  // twiddle tables of the FC_NB bins
        if (i < FC_NB * B) {
            const int jj = i / B, q = i - jj * B;
            const int kk = ((lo + jj) % nfft + nfft) % nfft;
            tw1[jj * ld1 + q] = tw_g[(kk * q) % nfft];
        } else {
            const int ii = i - FC_NB * B;
            const int q = ii / FC_NB, jj = ii - q * FC_NB;
            const int kk = ((lo + jj) % nfft + nfft) % nfft;
            tw2[q * FC_NB + jj] = tw_g[(((kk * B) % nfft) * q) % nfft];
        }
    }
    __syncthreads();
    // ---- level 1: S_k(m) for all nM blocks of the window.  Lane (c, j) starts its block at sample c mod B so
    // that the groups of a wave read different LDS banks (the 8 lanes of a group share the sample: broadcast).
    for (int m = c; m < nM; m += nthr / FC_NB) {
        const int xb0 = B * m;
        // B divides 128, so the block's B samples sit contiguously behind FC_XP(xb0); sample and twiddle share one running byte
        // offset that wraps inside the block (two integer instructions per step instead of the index -> padded index -> address
        // chain for each of the two reads)
        const unsigned char* xp = (const unsigned char*)(xs + FC_XP(xb0));
        const unsigned char* tp = (const unsigned char*)(tw1 + j * ld1);
        const unsigned wrap = (unsigned)B * (unsigned)sizeof(cplx) - 1u;
        unsigned o = (unsigned)(c & (B - 1)) * (unsigned)sizeof(cplx);
        double ar = 0.0, ai = 0.0;
#pragma unroll 4
        for (int i = 0; i < B; ++i) {
            const cplx v = *(const cplx*)(xp + o), t = *(const cplx*)(tp + o);
            ar = fma(v.x, t.x, fma(-v.y, t.y, ar));
            ai = fma(v.x, t.y, fma(v.y, t.x, ai));
            o = (o + (unsigned)sizeof(cplx)) & wrap;
        }
        Sp[j * nM + m] = make_double2(ar, ai);
    }
    __syncthreads();
    DEV_STAMP(KID_CERT, blockIdx.y * gridDim.x + blockIdx.x, 3);
    // ---- level 2: anchor X_k(64c) ----
    double xr = 0.0, xi = 0.0;
    if (act) {
        const cplx* sb = Sp + j * nM + (FS_CHUNK / B) * c;
        const cplx* tb = tw2 + j;
#pragma unroll 4
        for (int a = 0; a < nA; ++a) {
            const cplx v = sb[a], t = tb[a * FC_NB];
            xr = fma(v.x, t.x, fma(-v.y, t.y, xr));
            xi = fma(v.x, t.y, fma(v.y, t.x, xi));
        }
    }
    __syncthreads();                                      // S and the tables are dead: E and sumS take their place
    DEV_STAMP(KID_CERT, blockIdx.y * gridDim.x + blockIdx.x, 4);
    FC_PRIO(0, 1, 2)                                      // E(t), slides, certificate: youngest first
    // ---- E(t): E(0) by a block reduction, then a block scan of g[q] = |x[q+nfft]|^2 - |x[q]|^2 ----
    {
        double e0 = 0.0;
        for (int i = tid; i < nfft; i += nthr) { const cplx v = xs[FC_XP(i)]; e0 += v.x * v.x + v.y * v.y; }
        for (int off = 32; off > 0; off >>= 1) e0 += __shfl_down(e0, off, 64);
        if (lane == 0) red_p[wave] = e0;
        const int per = (nstep + nthr - 1) / nthr;
        const int i0 = tid * per < nstep ? tid * per : nstep, i1 = i0 + per < nstep ? i0 + per : nstep;
        // |d_q| in fp32 (v_sqrt_f32 is one instruction, the fp64 square root a dozen): C(t) only has to be an UPPER bound of
        // the slack, each term is scaled up by 1 + 2^-20 > the rounding of the fp32 conversion and square root (< 2^-22)
        double loc = 0.0, locd = 0.0;
        for (int q = i0; q < i1; ++q) {
            const cplx a1 = xs[FC_XP(q + nfft)], b1 = xs[FC_XP(q)];
            loc += (a1.x * a1.x + a1.y * a1.y) - (b1.x * b1.x + b1.y * b1.y);
            const double dr = a1.x - b1.x, di = a1.y - b1.y;
            locd += (double)(__fsqrt_rn((float)(dr * dr + di * di)) * 1.00000095367431640625f);
        }
        double inc = loc, incd = locd;                    // inclusive scans across the wave
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(inc, off, 64), od = __shfl_up(incd, off, 64);
            if (lane >= off) { inc += o; incd += od; }
        }
        if (lane == 63) { sh_scan[wave] = inc; sh_scan2[wave] = incd; }
        __syncthreads();
        double run = inc - loc, rund = incd - locd;
        for (int i = 0; i < wave; ++i) { run += sh_scan[i]; rund += sh_scan2[i]; }
        for (int i = 0; i < nwave; ++i) run += red_p[i];  // + E(0)
        for (int q = i0; q < i1; ++q) {
            Et[q] = run;
            Cs[FC_XP(q)] = (float)rund;
            const cplx a1 = xs[FC_XP(q + nfft)], b1 = xs[FC_XP(q)];
            run += (a1.x * a1.x + a1.y * a1.y) - (b1.x * b1.x + b1.y * b1.y);
            const double dr = a1.x - b1.x, di = a1.y - b1.y;
            rund += (double)(__fsqrt_rn((float)(dr * dr + di * di)) * 1.00000095367431640625f);
        }
        if (i1 == nstep && i0 < nstep) { Et[nstep] = run; Cs[FC_XP(nstep)] = (float)rund; }
    }
    DEV_STAMP(KID_CERT, blockIdx.y * gridDim.x + blockIdx.x, 5);
    // ---- slides: lane (c, j) walks chunk c of bin k; the 8 lanes of a group share the shift ----
    double best = -1.0;
    int bt = 0x7fffffff;
    if (act) {                                            // (uniform per 8-lane group)
        const cplx wk = tw_g[k];                          // exp(-2 pi i k/nfft): the recurrence turns by its conjugate
        const double wr = wk.x, wi = -wk.y;
        const int t0 = c * FS_CHUNK;
        if (c == 0) {                                     // the start window m = 0
            const double p = xr * xr + xi * xi;
            best = p; bt = 0;
            sumS[FC_XP(0)] = fc_group8_sum((float)p);
        }
        const int xa0 = t0 + nfft, xb0 = t0;
        for (int q0 = 0; q0 < FS_CHUNK; q0 += FC_PF) {
            double dr[FC_PF], di[FC_PF];
#pragma unroll
            for (int u = 0; u < FC_PF; ++u) {             // all LDS reads of the group first: one wait, not FC_PF
                const cplx a1 = xs[FC_XP(xa0 + q0 + u)], b1 = xs[FC_XP(xb0 + q0 + u)];
                dr[u] = a1.x - b1.x;
                di[u] = a1.y - b1.y;
            }
            float ps[FC_PF];
#pragma unroll
            for (int u = 0; u < FC_PF; ++u) {
                const double ar = xr + dr[u], ai = xi + di[u];
                xr = ar * wr - ai * wi;
                xi = ar * wi + ai * wr;
                const double p = xr * xr + xi * xi;
                if (p > best) { best = p; bt = t0 + q0 + u + 1; }   // shifts ascend: strict > keeps the first maximum
                ps[u] = fc_group8_sum((float)p);
            }
#pragma unroll
            for (int u = 0; u < FC_PF; ++u) sumS[FC_XP(t0 + q0 + u + 1)] = ps[u];
        }
    }
    int bk = k;
    for (int off = 32; off > 0; off >>= 1) {
        const double op = __shfl_down(best, off, 64);
        const int ot = __shfl_down(bt, off, 64);
        const int ok = __shfl_down(bk, off, 64);
        if (op > best || (op == best && (ot < bt || (ot == bt && ok < bk)))) { best = op; bt = ot; bk = ok; }
    }
    __syncthreads();                                      // red_p was E(0) until here
    DEV_STAMP(KID_CERT, blockIdx.y * gridDim.x + blockIdx.x, 6);
    if (lane == 0) { red_p[wave] = best; red_t[wave] = bt; red_k[wave] = bk; }
    __syncthreads();
    best = red_p[0]; bt = red_t[0]; bk = red_k[0];
    for (int i = 1; i < nwave; ++i)
        if (red_p[i] > best || (red_p[i] == best && (red_t[i] < bt || (red_t[i] == bt && red_k[i] < bk)))) {
            best = red_p[i]; bt = red_t[i]; bk = red_k[i];
        }
    // ---- certificate per shift.  Two bounds on every bin outside S:
    //   Parseval at the shift itself:            |X_k(t)| <= s(t) = sqrt(R(t))
    //   the recurrence from any other shift t':  |X_k(t)| <= s(t') + |C(t) - C(t')|   (each step adds at most |d_q|)
    // so U(t) = min( C(t) + min_{t'<=t}(s(t') - C(t')),  -C(t) + min_{t'>=t}(s(t') + C(t')) ) and shift t is certified
    // when U(t) < sqrt(P*).  fp32 group sums: each of the 8 terms and 7 additions errs by <= 2^-24 relative, all
    // terms are <= N*E(t), so |sumS - exact| < 2e-6 * N*E(t); the margins cover that, the fp32 storage of s and C
    // and the fp64 rounding of E and P.
    {
        const double INF = __longlong_as_double(0x7ff0000000000000LL);
        const double sb = best > 0.0 ? sqrt(best) * (1.0 - 1e-9) : -1.0;
        const int per = (nstep + nthr) / nthr;            // >= ceil((nstep+1)/nthr)
        const int u0 = tid * per < nstep + 1 ? tid * per : nstep + 1, u1 = u0 + per < nstep + 1 ? u0 + per : nstep + 1;
        double mf = INF, mb = INF;
        for (int t = u0; t < u1; ++t) {
            const double NE = (double)nfft * Et[t];
            const float ss = sumS[FC_XP(t)];
            const double R = NE - (double)ss + 4e-6 * NE;
            // (fp32 square root: s(t) is stored in fp32 anyway; the 1e-6 margin covers both roundings, < 2^-22 together)
            float sf = ss <= 3.0e38f ? __fsqrt_rn((float)(R > 0.0 ? R : 0.0)) * 1.000001f : __int_as_float(0x7f800000);   // (an overflowed fp32 sum bounds nothing)
            if (!(sf >= 0.0f)) sf = __int_as_float(0x7f800000);                  // NaN input
            const double sv = (double)sf;
            (void)sv;
            sumS[FC_XP(t)] = sf;                          // own range only: s(t) replaces the group sum
            const double cv = (double)Cs[FC_XP(t)];
            mf = fmin(mf, (double)sf - cv);
            mb = fmin(mb, (double)sf + cv);
        }
        double pf = mf, pb = mb;                          // inclusive prefix-min / suffix-min across the wave
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(pf, off, 64), ob = __shfl_down(pb, off, 64);
            if (lane >= off) pf = fmin(pf, o);
            if (lane + off < 64) pb = fmin(pb, ob);
        }
        __syncthreads();                                  // sh_scan / sh_scan2 are free again
        if (lane == 63) sh_scan[wave] = pf;
        if (lane == 0) sh_scan2[wave] = pb;
        double ef = __shfl_up(pf, 1, 64), eb = __shfl_down(pb, 1, 64);   // exclusive parts inside the wave
        if (lane == 0) ef = INF;
        if (lane == 63) eb = INF;
        __syncthreads();
        for (int i = 0; i < wave; ++i) ef = fmin(ef, sh_scan[i]);
        for (int i = wave + 1; i < nwave; ++i) eb = fmin(eb, sh_scan2[i]);
        const double ctot = (double)Cs[FC_XP(nstep)] * 1e-6;     // covers the fp32 rounding of the C(t) differences
        unsigned okm = 0;                                 // per <= 32 for every supported geometry
        for (int t = u0; t < u1; ++t) {
            const double cv = (double)Cs[FC_XP(t)];
            ef = fmin(ef, (double)sumS[FC_XP(t)] - cv);
            if (cv + ef + ctot < sb) okm |= 1u << (t - u0);
        }
        for (int t = u1 - 1; t >= u0; --t) {
            const double cv = (double)Cs[FC_XP(t)];
            eb = fmin(eb, (double)sumS[FC_XP(t)] + cv);
            const bool okc = ((okm >> (t - u0)) & 1u) || (eb - cv + ctot < sb);
            if (!okc) {
                if (t <= bt) atomicMax(&sh_a, t + 1);     // uncertified prefix [0, a)
                if (t >= bt) atomicMin(&sh_b, t - 1);     // uncertified suffix (b, nstep]
            }
        }
    }
    __syncthreads();
#ifdef GSMCAL_AB_LAZY_WIN
    if (fg.raw && OV > 0 && NTAPS == 47 && fg.sym47 != 0 && (sh_a != 0 || sh_b < nstep)) {      // (block-uniform) chunks stay open: k_fine_chunk / the exact pass read the window
        cplx* wout = fg.win_out + (size_t)s * win_stream_stride + (size_t)w * win_stride;
        for (int i = tid; i < wlen; i += nthr) wout[i] = xs[FC_XP(i)];
    }
#endif
    if (tid == 0) {
        FineCert o;
        o.p = best; o.t = bt; o.k = bk; o.a = sh_a; o.b = sh_b;
        // every uncertified shift lies in the prefix [0, a) or in the suffix (b, nstep] (the certificate's loop above): only
        // the chunks that hold those are swept.  Chunk c holds shifts 64c+1 .. 64c+64 (+ shift 0 in chunk 0).  (Round 3: a
        // window with any suffix open used to be swept whole -- 560 chunks instead of 165 on the 64 mixed streams of bench.py.)
        const int a = sh_a, b = sh_b;
        int npre = a == 0 ? 0 : (a == 1 ? 1 : (a - 2) / FS_CHUNK + 1);
        int nsuf = b < nstep ? nchunk - b / FS_CHUNK : 0;  // shift b+1 is the first of the suffix: chunk b/64
        if (npre + nsuf > nchunk) { npre = nchunk; nsuf = 0; }
        o.nch = npre + nsuf; o.nsuf = nsuf;
        cert[(size_t)s * H + w] = o;
        if (o.nch > 0) {                                  // work list of k_fine_chunk
            const int base = atomicAdd(n_items, o.nch);
            for (int i = 0; i < o.nch; ++i) items[base + i] = ((s * H + w) << 8) | fc_open_chunk(i, o.nch, nsuf, nchunk);
        }
    }
    DEV_STAMP(KID_CERT, blockIdx.y * gridDim.x + blockIdx.x, 7);
}

// The all-bins fp64 search (GSMCAL_PRESCREEN=0): kept as the ablation / cross-check of the two-pass scheme.
__global__ void __launch_bounds__(256) k_fine_search(const StreamState* __restrict__ sts,
                                                     const cplx* __restrict__ win, long win_stream_stride,
                                                     long win_stride, int nshift, int nfft,
                                                     const cplx* __restrict__ x0, PeakOut* __restrict__ out,
                                                     int H, int NB) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cplx* d = (cplx*)smem;                      // nshift-1 differences x[t+nfft]-x[t] (+ FS_CHUNK zero pad)
    __shared__ double red_p[4];
    __shared__ int red_t[4], red_k[4];
    const int s = blockIdx.z, w = blockIdx.y;
    if (w >= sts[s].n_win) return;
    const cplx* x = win + (size_t)s * win_stream_stride + (size_t)w * win_stride;
    const int tid = threadIdx.x;
    const int nstep = nshift - 1;
    const int nstep_pad = (nstep + FS_CHUNK - 1) / FS_CHUNK * FS_CHUNK;
    const int k = blockIdx.x * 256 + tid;
    const bool cand = k < nfft;
    for (int t = tid; t < nstep_pad; t += 256) {
        cplx v = make_double2(0.0, 0.0);
        if (t < nstep) {
            const cplx a = x[t + nfft], b = x[t];
            v = make_double2(a.x - b.x, a.y - b.y);
        }
        d[t] = v;
    }
    __syncthreads();
    double best = -1.0;          // running best power of this bin
    int best_c = -1;             // chunk that attained it (-1: the start window m = 0)
    double sxr = 0.0, sxi = 0.0; // state at the start of that chunk
    double wr = 1.0, wi = 0.0;
    if (cand) {
        sincospi(2.0 * (double)k / (double)nfft, &wi, &wr);
        const cplx xi0 = x0[((size_t)s * H + w) * nfft + k];
        double xr = xi0.x, xi = xi0.y;
        best = xr * xr + xi * xi;               // window start m = 0
        const int nchunk = nstep_pad / FS_CHUNK;
        for (int c = 0; c < nchunk; ++c) {
            const double cxr = xr, cxi = xi;
            double cmax = -1.0;
            const cplx* dc = d + c * FS_CHUNK;
            const int lim = nstep - c * FS_CHUNK;   // real steps in this chunk (< FS_CHUNK only in a ragged last chunk)
            if (lim >= FS_CHUNK) {
#pragma unroll 8
                for (int j = 0; j < FS_CHUNK; ++j) {
                    const cplx dv = dc[j];
                    const double ar = xr + dv.x, ai = xi + dv.y;
                    xr = ar * wr - ai * wi;
                    xi = ar * wi + ai * wr;
                    cmax = fmax(cmax, xr * xr + xi * xi);
                }
            } else {
                for (int j = 0; j < lim; ++j) {
                    const cplx dv = dc[j];
                    const double ar = xr + dv.x, ai = xi + dv.y;
                    xr = ar * wr - ai * wi;
                    xi = ar * wi + ai * wr;
                    cmax = fmax(cmax, xr * xr + xi * xi);
                }
            }
            if (cmax > best) { best = cmax; best_c = c; sxr = cxr; sxi = cxi; }
        }
    }
    // block winner: larger power, then earlier chunk, then smaller bin
    double rb = best;
    int rc = cand ? best_c : 0x7ffffff0, rk = k;
    for (int off = 32; off > 0; off >>= 1) {
        const double op = __shfl_down(rb, off, 64);
        const int oc = __shfl_down(rc, off, 64);
        const int ok = __shfl_down(rk, off, 64);
        if (op > rb || (op == rb && (oc < rc || (oc == rc && ok < rk)))) { rb = op; rc = oc; rk = ok; }
    }
    const int wv = tid >> 6;
    if ((tid & 63) == 0) { red_p[wv] = rb; red_t[wv] = rc; red_k[wv] = rk; }
    __syncthreads();
    rb = red_p[0]; rc = red_t[0]; rk = red_k[0];
    for (int i = 1; i < 4; ++i)
        if (red_p[i] > rb || (red_p[i] == rb && (red_t[i] < rc || (red_t[i] == rc && red_k[i] < rk)))) {
            rb = red_p[i]; rc = red_t[i]; rk = red_k[i];
        }
    if (k == rk) {                              // the winning lane locates the first step inside its chunk
        int m = 0;
        if (best_c >= 0) {
            double xr = sxr, xi = sxi;
            const cplx* dc = d + best_c * FS_CHUNK;
            const int lim = nstep - best_c * FS_CHUNK < FS_CHUNK ? nstep - best_c * FS_CHUNK : FS_CHUNK;
            for (int j = 0; j < lim; ++j) {
                const cplx dv = dc[j];
                const double ar = xr + dv.x, ai = xi + dv.y;
                xr = ar * wr - ai * wi;
                xi = ar * wi + ai * wr;
                const double p = xr * xr + xi * xi;
                if (p == best) { m = best_c * FS_CHUNK + j + 1; break; }
            }
        }
        PeakOut o; o.p = best; o.tie = m; o.k = k;
        out[((size_t)s * H + w) * NB + blockIdx.x] = o;
    }
}
