// kernels_detect.h -- FCCH coarse detector and the sliding-DFT peak search.
//
//   k_coarse      FCCH_coarse_position.m:5-94 with move_fft_snr_runtime_avg.m:5-50 and
//                 specific_fft_snr_fix_avg.m:5-34 inside: one workgroup per stream; all sliding
//                 window SNRs in parallel, then the reference's serial moving-average recurrence
//                 (bit-for-bit the same update order) and the hop loop, all on the device.
//   k_slide_dft   FCCH_fine_correction.m:48-52: max over 1025 window starts of max_k |FFT_1184|^2,
//                 as an exact sliding DFT (one bin per lane, fp64 state); also used with one shift
//                 for the burst spectra of :148-150 / carrier_correct_post_SCH.m:63-65.
#pragma once
#include "state.h"
#include "kernels_frontend.h"

// ---- per-window SNR (move_fft_snr_runtime_avg.m:18-27) ------------------------------------------
// 16-point FFT, radix-2 DIT, fully unrolled in registers (fp64).
__device__ __forceinline__ void fft16(cplx* x) {
    // bit reversal
    const int rev[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
    cplx y[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) y[i] = x[rev[i]];
    const double c1 = 0.92387953251128673848, s1 = 0.38268343236508978178;  // cos/sin(pi/8)
    const double r2 = 0.70710678118654752440;
    const cplx w16[8] = {{1.0, 0.0}, {c1, -s1}, {r2, -r2}, {s1, -c1}, {0.0, -1.0}, {-s1, -c1}, {-r2, -r2}, {-c1, -s1}};
#pragma unroll
    for (int len = 2; len <= 16; len <<= 1) {
        const int half = len >> 1, step = 16 / len;
#pragma unroll
        for (int i = 0; i < 16; i += len) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const cplx w = w16[j * step];
                const cplx u = y[i + j];
                const cplx v = cmul(y[i + j + half], w);
                y[i + j] = make_double2(u.x + v.x, u.y + v.y);
                y[i + j + half] = make_double2(u.x - v.x, u.y - v.y);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = y[i];
}

// SNR of one window; generic length via direct DFT (tw = exp(-2*pi*i*m/L) table), 16 via fft16.
__device__ __forceinline__ double window_snr(const cplx* __restrict__ s, int fft_len, const cplx* tw) {
    double P[64];
    if (fft_len == 16) {
        cplx x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = s[i];
        fft16(x);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double m = hypot(x[i].x, x[i].y);  // abs(fft(.)).^2
            P[i] = m * m;
        }
    } else {
        for (int k = 0; k < fft_len; ++k) {
            double ar = 0.0, ai = 0.0;
            int idx = 0;
            for (int n = 0; n < fft_len; ++n) {
                const cplx w = tw[idx];
                ar += s[n].x * w.x - s[n].y * w.y;
                ai += s[n].x * w.y + s[n].y * w.x;
                idx += k;
                if (idx >= fft_len) idx -= fft_len;
            }
            const double m = hypot(ar, ai);
            P[k] = m * m;
        }
    }
    int mi = 0;
    double mx = P[0];
    for (int k = 1; k < fft_len; ++k)
        if (P[k] > mx) { mx = P[k]; mi = k; }       // first max
    const int km = mi == 0 ? fft_len - 1 : mi - 1, kp = mi == fft_len - 1 ? 0 : mi + 1;
    double sig = P[km] + P[mi];
    sig = sig + P[kp];
    double tot = 0.0;
    for (int k = 0; k < fft_len; ++k) tot += P[k];
    const double noise = tot - sig;
    return 10.0 * log10(sig / noise);
}

struct CoarseArgs {
    const cplx* s; long s_stride; long len;   // decimated streams
    int decimation_ratio;                     // FCCH_coarse_position's 2nd argument
    int mode;      // 0 = FCCH_coarse_position, 1 = move_fft_snr_runtime_avg only, 2 = specific only
    int mv_len, fft_len; double th;           // modes 1/2 (mode 0 derives them like the reference)
    long t_lo, t_hi; double avg_snr;          // mode 2: target_set and fixed average
};

// grid S, block 256.  LDS: snr[nwin] doubles + twiddles.
__global__ void __launch_bounds__(256) k_coarse(StreamState* __restrict__ sts, CoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int sh_hit;     // hit window (0-based) or -1
    __shared__ int sh_i[4];
    __shared__ double sh_d[4];
    StreamState* st = sts + blockIdx.x;
    const cplx* s = a.s + (size_t)blockIdx.x * a.s_stride;
    const long len = a.len;
    int fft_len, mv_len;
    double th;
    long n_first;
    if (a.mode == 0) {
        // FCCH_coarse_position.m:15-25
        fft_len = 1 << (int)floor(log2(148.0 / (double)a.decimation_ratio));
        th = 10.0;
        mv_len = 10 * fft_len;
        n_first = (long)ceil(23.0 * 1250.0 / (double)a.decimation_ratio);
    } else {
        fft_len = a.fft_len; mv_len = a.mv_len; th = a.th; n_first = len;
    }
    cplx* tw = (cplx*)smem;                       // fft_len twiddles
    double* snr_s = (double*)(tw + fft_len);      // nwin SNRs
    const int tid = threadIdx.x;
    if (tid == 0) {
        st->n_coarse = 0;
        st->coarse_hit_flag = 0;
        st->hit_avg_snr = INFINITY;
        st->mv_hit_idx = -1.0;
        st->mv_hit_snr = INFINITY;
    }
    if (fft_len != 16)
        for (int i = tid; i < fft_len; i += 256) {
            double sn, cs;
            sincospi(-2.0 * (double)i / (double)fft_len, &sn, &cs);
            tw[i] = make_double2(cs, sn);
        }
    __syncthreads();
    if (fft_len > 64 || fft_len < 2 || n_first > len) {  // s(1:n_first) would be a MATLAB index error
        if (tid == 0) set_status(st, 3, GSMCAL_E_INDEX);
        return;
    }
    double hit_avg_snr = a.avg_snr;
    if (a.mode != 2) {
        // ---- move_fft_snr_runtime_avg ----
        const long nwin = n_first - (fft_len - 1);
        for (long i = tid; i < nwin; i += 256) snr_s[i] = window_snr(s + i, fft_len, tw);
        __syncthreads();
        if (tid == 0) {
            // :11-12 store = 999*ones(1,mv_len); sum_snr = sum(store): sequential sum of 999s
            double sum_snr = 0.0;
            for (int i = 0; i < mv_len; ++i) sum_snr += 999.0;
            int hit = -1;
            double h_snr = 0.0, h_pta = 0.0;
            const double dmv = (double)mv_len;
            for (long i = 0; i < nwin; ++i) {
                const double snr = snr_s[i];
                const double pta = snr - (sum_snr / dmv);            // :30
                if (pta > th) { hit = (int)i; h_snr = snr; h_pta = pta; break; }
                const double oldest = i >= mv_len ? snr_s[i - mv_len] : 999.0;
                sum_snr = sum_snr - oldest;                          // :37
                sum_snr = sum_snr + snr;                             // :38
            }
            sh_hit = hit;
            if (hit >= 0) {
                st->coarse_hit_flag = 1;
                st->mv_hit_idx = (double)(hit + 1);
                st->mv_hit_snr = h_snr;
                st->hit_avg_snr = h_snr - h_pta;                     // :48
            }
        }
        __syncthreads();
        if (a.mode == 1) return;
        if (sh_hit < 0) {
            if (tid == 0) set_status(st, 3, GSMCAL_S_NO_FCCH);
            return;
        }
        hit_avg_snr = st->hit_avg_snr;
    }
    if (a.mode == 2) {
        // ---- specific_fft_snr_fix_avg stand-alone ----
        const long lo = a.t_lo, hi = a.t_hi;
        if (lo < 1 || hi + fft_len - 1 > len) {
            if (tid == 0 && hi >= lo) set_status(st, 3, GSMCAL_E_INDEX);
            if (hi >= lo) return;
        }
        const long cnt = hi - lo + 1;
        for (long i = tid; i < cnt; i += 256) snr_s[i] = window_snr(s + (lo - 1 + i), fft_len, tw);
        __syncthreads();
        if (tid == 0) {
            for (long i = 0; i < cnt; ++i)
                if (snr_s[i] - hit_avg_snr > th) {
                    st->coarse_hit_flag = 1;
                    st->mv_hit_idx = (double)(lo + i);
                    st->mv_hit_snr = snr_s[i];
                    break;
                }
        }
        return;
    }
    // ---- hop loop of FCCH_coarse_position.m:32-86 (positions 1-based, decimated units) ----
    const int dec = a.decimation_ratio;
    const long d0 = (long)round(12500.0 / (double)dec);     // :35 round() half away from zero
    const long d1 = (long)round(13750.0 / (double)dec);     // :36
    const int max_offset = 5;
    const long limit = (len - (fft_len - 1)) - max_offset;
    long cur = sh_hit + 1;
    int n = 1;
    if (tid == 0) {
        st->coarse_pos[0] = (double)((cur - 1) * dec + 1);  // :91
        st->coarse_snr[0] = st->mv_hit_snr;
    }
    const int nt = 2 * max_offset + 1;
    while (n < MAXH) {
        long nxt = cur + d0;
        if (nxt > limit) break;                              // :49
        int found = -1;
        double fsnr = 0.0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            __syncthreads();
            if (tid < nt) snr_s[tid] = window_snr(s + (nxt - max_offset - 1 + tid), fft_len, tw);
            __syncthreads();
            for (int i = 0; i < nt; ++i)                     // every thread scans: uniform result
                if (snr_s[i] - hit_avg_snr > th) { found = i; fsnr = snr_s[i]; break; }
            if (found >= 0 || attempt == 1) break;
            nxt = cur + d1;                                  // :65 across the idle frame
            if (nxt > limit) break;                          // :67
        }
        if (found < 0) break;
        cur = nxt - max_offset + found;
        if (tid == 0) {
            st->coarse_pos[n] = (double)((cur - 1) * dec + 1);
            st->coarse_snr[n] = fsnr;
        }
        ++n;
    }
    if (tid == 0) st->n_coarse = n;
    (void)sh_i; (void)sh_d;
}

// ------------------------------------------------------------------------------------------------
// Sliding DFT peak search.  grid (NB, H, S), block 256; bin k = blockIdx.x*256 + tid < nfft.
//   X_k <- (X_k + x[t] - x[t-nfft]) * exp(+2*pi*i*k/nfft),  t = 0 .. wlen-1
// After sample t >= nfft-1 has been pushed, |X_k|^2 is the k-th power-spectrum bin of the window
// starting at m = t-(nfft-1).  Each lane keeps its best (power, first m); the block reduces to one
// PeakOut with tie rule "smaller key wins" (key = m, or the fftshift-ed bin when shifted_key != 0),
// which reproduces MATLAB's first-max rule of max(max(|fft|^2)) over windows (:50-52) and of
// max(fftshift-ed spectrum) (:149-150).
// ------------------------------------------------------------------------------------------------
// SHIFTED_KEY = 0: k_slide_dft<0> is the fine search (key = window start); 1: burst spectrum argmax
// in fftshift order (wlen == nfft, a single window).
template <int SHIFTED_KEY>
__global__ void __launch_bounds__(256) k_slide_dft(const StreamState* __restrict__ sts,
                                                   const cplx* __restrict__ win, long win_stream_stride,
                                                   long win_stride, int wlen, int nfft,
                                                   PeakOut* __restrict__ out, int H, int NB) {
    constexpr int shifted_key = SHIFTED_KEY;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cplx* d = (cplx*)smem;                      // wlen differences
    __shared__ double red_p[4];
    __shared__ int red_t[4], red_k[4];
    const int s = blockIdx.z, w = blockIdx.y;
    if (w >= sts[s].n_win) return;
    const cplx* x = win + (size_t)s * win_stream_stride + (size_t)w * win_stride;
    const int tid = threadIdx.x;
    for (int t = tid; t < wlen; t += 256) {
        const cplx a = x[t];
        const cplx b = t >= nfft ? x[t - nfft] : make_double2(0.0, 0.0);
        d[t] = make_double2(a.x - b.x, a.y - b.y);
    }
    __syncthreads();
    const int k = blockIdx.x * 256 + tid;
    double best = -1.0;
    int best_m = 0;
    if (k < nfft) {
        double wi, wr;
        sincospi(2.0 * (double)k / (double)nfft, &wi, &wr);
        double xr = 0.0, xi = 0.0;
        int t = 0;
        for (; t < nfft - 1; ++t) {             // warm-up: window not yet full
            const cplx dv = d[t];
            const double ar = xr + dv.x, ai = xi + dv.y;
            xr = ar * wr - ai * wi;
            xi = ar * wi + ai * wr;
        }
        for (; t < wlen; ++t) {
            const cplx dv = d[t];
            const double ar = xr + dv.x, ai = xi + dv.y;
            xr = ar * wr - ai * wi;
            xi = ar * wi + ai * wr;
            const double p = xr * xr + xi * xi;
            if (p > best) { best = p; best_m = t - (nfft - 1); }
        }
    }
    int key = best_m;
    if (shifted_key) key = k < nfft ? (k + nfft / 2) % nfft : 0x7fffffff;  // position after fftshift
    int kk = k;
    // wave reduce: larger p wins, equal p -> smaller key
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
        for (int i = 1; i < 4; ++i)
            if (red_p[i] > best || (red_p[i] == best && red_t[i] < key)) { best = red_p[i]; key = red_t[i]; kk = red_k[i]; }
        PeakOut o; o.p = best; o.tie = key; o.k = kk;
        out[((size_t)s * H + w) * NB + blockIdx.x] = o;
    }
}
