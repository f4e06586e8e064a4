// kernels_detect.h -- FCCH coarse detector and the sliding-DFT peak search.
//
//   k_coarse      FCCH_coarse_position.m:5-94 with move_fft_snr_runtime_avg.m:5-50 and
//                 specific_fft_snr_fix_avg.m:5-34 inside: one workgroup per stream; all sliding
//                 window SNRs in parallel, then the reference's serial moving-average recurrence
//                 (bit-for-bit the same update order) and the hop loop, all on the device.
//   k_fft_burst   1184-point spectra (37 x 32 Cooley-Tukey in LDS): burst spectrum argmax of
//                 FCCH_fine_correction.m:148-150 / carrier_correct_post_SCH.m:63-65, and the starting
//                 spectrum of the fine search
//   k_fine_search FCCH_fine_correction.m:48-52: max over 1025 window starts of max_k |FFT_1184|^2,
//                 as an exact sliding DFT (one bin per lane, fp64 state)
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

// SNR from the power spectrum P[0..L) (move_fft_snr_runtime_avg.m:22-27): first max, 3 circular bins
// around it vs the rest, in dB.  L is a compile-time constant so P stays in registers.
template <int L>
__device__ __forceinline__ double snr_from_power(const double (&P)[L]) {
    int mi = 0;
    double mx = P[0];
#pragma unroll
    for (int k = 1; k < L; ++k)
        if (P[k] > mx) { mx = P[k]; mi = k; }       // first max
    double pm = P[L - 1], pc = P[0], pp = P[1 % L];
#pragma unroll
    for (int k = 1; k < L; ++k) {
        const bool h = (k == mi);
        pm = h ? P[k - 1] : pm;
        pc = h ? P[k] : pc;
        pp = h ? P[(k + 1) % L] : pp;
    }
    double sig = pm + pc;
    sig = sig + pp;
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < L; ++k) tot += P[k];
    const double noise = tot - sig;
    return 10.0 * log10(sig / noise);
}

__device__ __forceinline__ double window_snr16(const cplx* __restrict__ s) {
    cplx x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = s[i];
    fft16(x);
    double P[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const double m = hypot(x[i].x, x[i].y);  // abs(fft(.)).^2
        P[i] = m * m;
    }
    return snr_from_power<16>(P);
}

// generic length (2..64) via direct DFT with tw[m] = exp(-2*pi*i*m/L); nothing is stored: the bins
// around the first max are recomputed (bit-identical) after the scan.
__device__ __forceinline__ double dft_bin_power(const cplx* __restrict__ s, int fft_len, const cplx* tw, int k) {
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
    return m * m;
}

__device__ __noinline__ double window_snr_generic(const cplx* __restrict__ s, int fft_len, const cplx* tw) {
    int mi = 0;
    double mx = -1.0, tot = 0.0;
    for (int k = 0; k < fft_len; ++k) {
        const double p = dft_bin_power(s, fft_len, tw, k);
        tot += p;
        if (p > mx) { mx = p; mi = k; }             // first max
    }
    const int km = mi == 0 ? fft_len - 1 : mi - 1, kp = mi == fft_len - 1 ? 0 : mi + 1;
    double sig = dft_bin_power(s, fft_len, tw, km) + mx;
    sig = sig + dft_bin_power(s, fft_len, tw, kp);
    const double noise = tot - sig;
    return 10.0 * log10(sig / noise);
}

__device__ __forceinline__ double window_snr(const cplx* __restrict__ s, int fft_len, const cplx* tw) {
    return fft_len == 16 ? window_snr16(s) : window_snr_generic(s, fft_len, tw);
}

struct CoarseArgs {
    const cplx* s; long s_stride; long len;   // decimated streams
    int decimation_ratio;                     // FCCH_coarse_position's 2nd argument
    int mode;      // 0 = FCCH_coarse_position, 1 = move_fft_snr_runtime_avg only, 2 = specific only
    int mv_len, fft_len; double th;           // modes 1/2 (mode 0 derives them like the reference)
    long t_lo, t_hi; double avg_snr;          // mode 2: target_set and fixed average
};

// grid S, block 256.  LDS: 64 twiddles | running sums of one chunk | snr[nwin].
//
// move_fft_snr_runtime_avg's loop is serial only through sum_snr (two dependent fp64 adds per window,
// :37-38).  Per chunk of COARSE_CHUNK windows: (A) all lanes compute the window SNRs, (B) lane 0
// replays the reference's running-sum updates in the reference's order and records the sum each
// window sees, (C) all lanes evaluate snr - sum/mv_len > th (the true fp64 divide of :30) and the
// first hit wins.  Updates past a hit inside a chunk are never used -- the same as the `break`.
#define COARSE_CHUNK 512
__global__ void __launch_bounds__(256) k_coarse(StreamState* __restrict__ sts, CoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int sh_hit;     // first hit window (0-based) or INT_MAX
    __shared__ double sh_sum;  // running sum carried between chunks
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
    const int tid = threadIdx.x;
    cplx* tw = (cplx*)smem;                              // 64 twiddles
    double* sums = (double*)(tw + 64);                   // COARSE_CHUNK running sums
    double* snr_s = sums + COARSE_CHUNK;                 // nwin SNRs
    if (tid == 0) {
        st->n_coarse = 0;
        st->coarse_hit_flag = 0;
        st->hit_avg_snr = INFINITY;
        st->mv_hit_idx = -1.0;
        st->mv_hit_snr = INFINITY;
        sh_hit = 0x7fffffff;
    }
    if (fft_len != 16 && fft_len >= 2 && fft_len <= 64)
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
        const double dmv = (double)mv_len;
        if (tid == 0) {
            // :11-12 store = 999*ones(1,mv_len); sum_snr = sum(store): sequential sum of 999s
            double sum_snr = 0.0;
            for (int i = 0; i < mv_len; ++i) sum_snr += 999.0;
            sh_sum = sum_snr;
        }
        int hit = 0x7fffffff;
        for (long c0 = 0; c0 < nwin; c0 += COARSE_CHUNK) {
            const int cn = (int)(nwin - c0 < COARSE_CHUNK ? nwin - c0 : COARSE_CHUNK);
            for (int j = tid; j < cn; j += 256) snr_s[c0 + j] = window_snr(s + c0 + j, fft_len, tw);   // (A)
            __syncthreads();
            if (tid == 0) {                                                                               // (B)
                double sum_snr = sh_sum;
                for (int j = 0; j < cn; ++j) {
                    const long i = c0 + j;
                    sums[j] = sum_snr;
                    const double oldest = i >= mv_len ? snr_s[i - mv_len] : 999.0;
                    sum_snr = sum_snr - oldest;                          // :37
                    sum_snr = sum_snr + snr_s[i];                        // :38
                }
                sh_sum = sum_snr;
            }
            __syncthreads();
            int first = 0x7fffffff;                                                                       // (C)
            for (int j = tid; j < cn; j += 256) {
                const double pta = snr_s[c0 + j] - (sums[j] / dmv);      // :30
                if (pta > th && j < first) first = j;                    // :32 strict >
            }
            for (int off = 32; off > 0; off >>= 1) {
                const int o = __shfl_down(first, off, 64);
                first = o < first ? o : first;
            }
            if ((tid & 63) == 0 && first != 0x7fffffff) atomicMin(&sh_hit, (int)c0 + first);
            __syncthreads();
            hit = sh_hit;
            if (hit != 0x7fffffff) break;                                // uniform
        }
        if (hit != 0x7fffffff && tid == 0) {
            const double h_snr = snr_s[hit];
            const double h_pta = h_snr - (sums[hit % COARSE_CHUNK] / dmv);
            st->coarse_hit_flag = 1;
            st->mv_hit_idx = (double)(hit + 1);
            st->mv_hit_snr = h_snr;
            st->hit_avg_snr = h_snr - h_pta;                             // :48
        }
        __syncthreads();
        if (a.mode == 1) return;
        if (hit == 0x7fffffff) {
            if (tid == 0) set_status(st, 3, GSMCAL_S_NO_FCCH);
            return;
        }
        hit_avg_snr = st->hit_avg_snr;
    }
    if (a.mode == 2) {
        // ---- specific_fft_snr_fix_avg stand-alone ----
        const long lo = a.t_lo, hi = a.t_hi;
        if (hi < lo) return;
        if (lo < 1 || hi + fft_len - 1 > len) {
            if (tid == 0) set_status(st, 3, GSMCAL_E_INDEX);
            return;
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
    double* hop = sums;                                      // reuse: nt SNRs per attempt
    while (n < MAXH) {
        long nxt = cur + d0;
        if (nxt > limit) break;                              // :49
        int found = -1;
        double fsnr = 0.0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            __syncthreads();
            if (tid < nt) hop[tid] = window_snr(s + (nxt - max_offset - 1 + tid), fft_len, tw);
            __syncthreads();
            for (int i = 0; i < nt; ++i)                     // every thread scans: uniform result
                if (hop[i] - hit_avg_snr > th) { found = i; fsnr = hop[i]; break; }
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
}

// ------------------------------------------------------------------------------------------------
// 1184-point spectra.  nfft = 148*ov = 37 * N2 with N2 = 4*ov: Cooley-Tukey split n = N2*n1 + n2,
// k = k1 + 37*k2, both factors as direct DFTs in LDS with exact table twiddles
// tw[m] = exp(-2*pi*i*m/nfft) (k_make_twiddles, sincospi).  ~70 complex MACs per output instead of
// 1184: used for the burst spectra (FCCH_fine_correction.m:148-150, carrier_correct_post_SCH.m:63-65)
// and to start the sliding DFT of the fine search at its first window.
//   MODE 0: argmax of |X|^2 in fftshift order -> PeakOut (one per window)
//   MODE 1: X of the window's first nfft samples -> global x0[s][w][k]
// grid (H, S), block 256.  LDS: x[nfft] | B[37][N2+1] | tw[nfft].
// ------------------------------------------------------------------------------------------------
__global__ void k_make_twiddles(cplx* tw, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double sn, cs;
    sincospi(-2.0 * (double)i / (double)n, &sn, &cs);
    tw[i] = make_double2(cs, sn);
}

template <int MODE>
__global__ void __launch_bounds__(256) k_fft_burst(const StreamState* __restrict__ sts, const cplx* __restrict__ win,
                                                   long win_stream_stride, long win_stride, int nfft,
                                                   const cplx* __restrict__ tw_g, PeakOut* __restrict__ peaks,
                                                   cplx* __restrict__ x0, int H) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int N1 = 37;
    const int N2 = nfft / N1;
    const int ldb = N2 + 1;                     // padded row: conflict-free column reads in step 2
    cplx* xs = (cplx*)smem;                     // nfft
    cplx* B = xs + nfft;                        // N1 * ldb
    cplx* tw = B + N1 * ldb;                    // nfft
    __shared__ double red_p[4];
    __shared__ int red_t[4], red_k[4];
    const int s = blockIdx.y, w = blockIdx.x;
    if (w >= sts[s].n_win) return;
    const int tid = threadIdx.x;
    const cplx* x = win + (size_t)s * win_stream_stride + (size_t)w * win_stride;
    for (int i = tid; i < nfft; i += 256) { xs[i] = x[i]; tw[i] = tw_g[i]; }
    __syncthreads();
    // step 1: B[k1][n2] = W_nfft^(n2*k1) * sum_n1 x[N2*n1+n2] * W_37^(n1*k1),  W_37^m = tw[N2*m]
    for (int o = tid; o < nfft; o += 256) {
        const int k1 = o / N2, n2 = o - k1 * N2;
        double ar = 0.0, ai = 0.0;
        int idx = 0;
        const int stp = (N2 * k1) % nfft;
        for (int n1 = 0; n1 < N1; ++n1) {
            const cplx v = xs[N2 * n1 + n2], t = tw[idx];
            ar = fma(v.x, t.x, fma(-v.y, t.y, ar));
            ai = fma(v.x, t.y, fma(v.y, t.x, ai));
            idx += stp;
            if (idx >= nfft) idx -= nfft;
        }
        const cplx t = tw[(n2 * k1) % nfft];
        B[k1 * ldb + n2] = make_double2(ar * t.x - ai * t.y, ar * t.y + ai * t.x);
    }
    __syncthreads();
    // step 2: X[k1 + 37*k2] = sum_n2 B[k1][n2] * W_N2^(n2*k2),  W_N2^m = tw[37*m]
    double best = -1.0;
    int key = 0x7fffffff, kk = 0;
    for (int k = tid; k < nfft; k += 256) {
        const int k2 = k / N1, k1 = k - k2 * N1;
        double ar = 0.0, ai = 0.0;
        int idx = 0;
        const int stp = (N1 * k2) % nfft;
        const cplx* row = B + k1 * ldb;
        for (int n2 = 0; n2 < N2; ++n2) {
            const cplx v = row[n2], t = tw[idx];
            ar = fma(v.x, t.x, fma(-v.y, t.y, ar));
            ai = fma(v.x, t.y, fma(v.y, t.x, ai));
            idx += stp;
            if (idx >= nfft) idx -= nfft;
        }
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
        for (int i = 1; i < 4; ++i)
            if (red_p[i] > best || (red_p[i] == best && red_t[i] < key)) { best = red_p[i]; key = red_t[i]; kk = red_k[i]; }
        PeakOut o; o.p = best; o.tie = key; o.k = kk;
        peaks[(size_t)s * H + w] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// Fine search, FCCH_fine_correction.m:48-52: argmax over the nshift = 128*ov+1 window starts of
// max_k |FFT_nfft(window)|^2, as an exact sliding DFT in fp64: one bin per lane,
//   X_k(m+1) = (X_k(m) + x[m+nfft] - x[m]) * exp(+2*pi*i*k/nfft),   X_k(0) from k_fft_burst<1>.
// Each lane keeps its best (power, first m); the block reduces with "larger power, then smaller m",
// which is MATLAB's first-max rule for max(max(|fft|^2,[],1)).
// grid (NB, H, S), block 256; bin k = blockIdx.x*256 + tid.  LDS: nshift-1 differences.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_fine_search(const StreamState* __restrict__ sts,
                                                     const cplx* __restrict__ win, long win_stream_stride,
                                                     long win_stride, int nshift, int nfft,
                                                     const cplx* __restrict__ x0, PeakOut* __restrict__ out,
                                                     int H, int NB) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cplx* d = (cplx*)smem;                      // nshift-1 differences x[t+nfft]-x[t]
    __shared__ double red_p[4];
    __shared__ int red_t[4], red_k[4];
    const int s = blockIdx.z, w = blockIdx.y;
    if (w >= sts[s].n_win) return;
    const cplx* x = win + (size_t)s * win_stream_stride + (size_t)w * win_stride;
    const int tid = threadIdx.x;
    const int nstep = nshift - 1;
    for (int t = tid; t < nstep; t += 256) {
        const cplx a = x[t + nfft], b = x[t];
        d[t] = make_double2(a.x - b.x, a.y - b.y);
    }
    __syncthreads();
    const int k = blockIdx.x * 256 + tid;
    double best = -1.0;
    int best_m = 0;
    if (k < nfft) {
        double wi, wr;
        sincospi(2.0 * (double)k / (double)nfft, &wi, &wr);
        const cplx xi0 = x0[((size_t)s * H + w) * nfft + k];
        double xr = xi0.x, xi = xi0.y;
        best = xr * xr + xi * xi;               // window start m = 0
#pragma unroll 4
        for (int t = 0; t < nstep; ++t) {
            const cplx dv = d[t];
            const double ar = xr + dv.x, ai = xi + dv.y;
            xr = ar * wr - ai * wi;
            xi = ar * wi + ai * wr;
            const double p = xr * xr + xi * xi;
            if (p > best) { best = p; best_m = t + 1; }
        }
    }
    int key = best_m, kk = k;
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
