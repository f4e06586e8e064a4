// kernels_demod.h -- front end of the SCH burst demodulator: per SCH burst, the frequency-domain channel
// estimate and equalisation of SCH_demod.m:53-59,79-90.  (The Viterbi GMSK demodulator behind it is Communications
// Toolbox code whose output the reference discards; it stays out.)
//
//   len_fde_ov = (148 + 2*8 + 30) * ov            = 1552 at 8x: the burst with 8 symbols either side + the traceback depth
//   td_training = zeros(1, L); td_training(sp_t : sp_t+64*ov-1) = training_sequence, sp_t = (8+42)*ov + 1      :56-58
//   fd_training = fft(td_training)                                                                                 :59
//   per burst i: x = s(sch_pos(i) - 8*ov : +L-1)                                                                :79-81
//       fd_chn = fft(x masked to the training span) ./ fd_training                                              :83-86
//       x_eq   = ifft( fft(x) ./ fd_chn.' )                                                                     :88-90
//
// L = 194*ov = 97 * (2*ov): Cooley-Tukey split n = N2*n1 + n2, k = k1 + 97*k2 with both factors as direct DFTs in
// LDS (97 is prime), twiddles from one exact table tw[m] = exp(-2 pi i m/L) (sincospi).  ~113 complex MACs per output
// instead of 1552.
#pragma once
#include "state.h"
#include "kernels_frontend.h"

#define DM_THREADS 512
#define DM_N1 97

// one forward (SIGN = -1) or inverse-without-1/L (SIGN = +1) DFT of in[0..L) -> out[0..L), both in LDS; B: N1 x (N2+1).
// Contains barriers; every thread of the block must call it.
template <int SIGN>
__device__ __forceinline__ void dm_dft(const cplx* in, cplx* out, cplx* B, const cplx* __restrict__ tw, int L, int N2, int tid) {
    const int ldb = N2 + 1;
    for (int o = tid; o < L; o += DM_THREADS) {
        const int k1 = o / N2, n2 = o - k1 * N2;
        double ar0 = 0.0, ai0 = 0.0, ar1 = 0.0, ai1 = 0.0;
        int idx = 0;                                   // (n1*k1 mod 97) * N2: index of W_97^(n1 k1) in the length-L table
        const int stp = k1 * N2;
        int n1 = 0;
        for (; n1 + 2 <= DM_N1; n1 += 2) {
            const cplx v0 = in[N2 * n1 + n2], t0 = tw[idx];
            idx += stp; if (idx >= L) idx -= L;
            const cplx v1 = in[N2 * (n1 + 1) + n2], t1 = tw[idx];
            idx += stp; if (idx >= L) idx -= L;
            const double s0 = SIGN < 0 ? t0.y : -t0.y, s1 = SIGN < 0 ? t1.y : -t1.y;
            ar0 = fma(v0.x, t0.x, fma(-v0.y, s0, ar0)); ai0 = fma(v0.x, s0, fma(v0.y, t0.x, ai0));
            ar1 = fma(v1.x, t1.x, fma(-v1.y, s1, ar1)); ai1 = fma(v1.x, s1, fma(v1.y, t1.x, ai1));
        }
        for (; n1 < DM_N1; ++n1) {
            const cplx v0 = in[N2 * n1 + n2], t0 = tw[idx];
            idx += stp; if (idx >= L) idx -= L;
            const double s0 = SIGN < 0 ? t0.y : -t0.y;
            ar0 = fma(v0.x, t0.x, fma(-v0.y, s0, ar0)); ai0 = fma(v0.x, s0, fma(v0.y, t0.x, ai0));
        }
        const double ar = ar0 + ar1, ai = ai0 + ai1;
        const cplx t = tw[n2 * k1];                    // inter-stage twiddle W_L^(n2 k1), n2*k1 < L
        const double ts = SIGN < 0 ? t.y : -t.y;
        B[k1 * ldb + n2] = make_double2(ar * t.x - ai * ts, ar * ts + ai * t.x);
    }
    __syncthreads();
    for (int k = tid; k < L; k += DM_THREADS) {
        const int k2 = k / DM_N1, k1 = k - k2 * DM_N1;
        double ar = 0.0, ai = 0.0;
        int idx = 0;                                   // (n2*k2 mod N2) * 97
        const int stp = (k2 % N2) * DM_N1;
        const cplx* row = B + k1 * ldb;
        for (int n2 = 0; n2 < N2; ++n2) {
            const cplx v = row[n2], t = tw[idx];
            idx += stp; if (idx >= L) idx -= L;
            const double s = SIGN < 0 ? t.y : -t.y;
            ar = fma(v.x, t.x, fma(-v.y, s, ar)); ai = fma(v.x, s, fma(v.y, t.x, ai));
        }
        out[k] = make_double2(ar, ai);
    }
    __syncthreads();
}

// MATLAB's complex right division a ./ b with scaling against overflow (Smith's algorithm)
__device__ __forceinline__ cplx dm_cdiv(cplx a, cplx b) {
    if (fabs(b.x) >= fabs(b.y)) {
        const double r = b.y / b.x, d = b.x + b.y * r;
        return make_double2((a.x + a.y * r) / d, (a.y - a.x * r) / d);
    }
    const double r = b.x / b.y, d = b.x * r + b.y;
    return make_double2((a.x * r + a.y) / d, (a.y * r - a.x) / d);
}

inline size_t dm_lds_bytes(int L, int N2) { return ((size_t)3 * L + (size_t)DM_N1 * (N2 + 1)) * sizeof(cplx); }

// fd_training (SCH_demod.m:56-59): grid 1, block DM_THREADS.
__global__ void __launch_bounds__(DM_THREADS) k_sch_fd_training(const cplx* __restrict__ ts, int len_ts, int sp_t0, int L, int N2,
                                                                const cplx* __restrict__ tw, cplx* __restrict__ fd_training) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cplx* a = (cplx*)smem;
    cplx* b = a + L;
    cplx* B = b + 2 * L;
    const int tid = threadIdx.x;
    for (int i = tid; i < L; i += DM_THREADS) a[i] = (i >= sp_t0 && i < sp_t0 + len_ts) ? ts[i - sp_t0] : make_double2(0.0, 0.0);
    __syncthreads();
    dm_dft<-1>(a, b, B, tw, L, N2, tid);
    for (int i = tid; i < L; i += DM_THREADS) fd_training[i] = b[i];
}

// one workgroup per SCH burst: x_eq[burst][0..L).  start[burst] = 0-based index of s(sp); status[burst] != 0: the
// reference would stop with an index error (s(sp:ep) outside the stream) and nothing is written.
__global__ void __launch_bounds__(DM_THREADS) k_sch_equalise(const cplx* __restrict__ s, long len, const long* __restrict__ start,
                                                             int len_ts, int sp_t0, int L, int N2, const cplx* __restrict__ tw,
                                                             const cplx* __restrict__ fd_training, cplx* __restrict__ out,
                                                             int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cplx* x = (cplx*)smem;               // the burst, later fd_x ./ fd_chn
    cplx* r = x + L;                     // the burst masked to the training span, later its spectrum
    cplx* f = r + L;                     // fft(x), later the equalised burst
    cplx* B = f + L;
    const int tid = threadIdx.x, w = blockIdx.x;
    const long sp = start[w];
    if (sp < 0 || sp + L > len) {        // block-uniform
        if (tid == 0) status[w] = GSMCAL_E_INDEX;
        return;
    }
    if (tid == 0) status[w] = 0;
    for (int i = tid; i < L; i += DM_THREADS) {
        const cplx v = s[sp + i];
        x[i] = v;
        r[i] = (i >= sp_t0 && i < sp_t0 + len_ts) ? v : make_double2(0.0, 0.0);     // :83-84
    }
    __syncthreads();
    dm_dft<-1>(x, f, B, tw, L, N2, tid);         // fd_x = fft(x)                                      :88
    dm_dft<-1>(r, x, B, tw, L, N2, tid);         // fd_received_training = fft(received_training_ov)    :85   (x is free)
    for (int i = tid; i < L; i += DM_THREADS) {
        const cplx chn = dm_cdiv(x[i], fd_training[i]);                                              // :86
        r[i] = dm_cdiv(f[i], chn);                                                                     // :89
    }
    __syncthreads();
    dm_dft<+1>(r, f, B, tw, L, N2, tid);         // ifft                                                :90
    const double inv = 1.0 / (double)L;
    cplx* o = out + (size_t)w * L;
    for (int i = tid; i < L; i += DM_THREADS) o[i] = make_double2(f[i].x * inv, f[i].y * inv);
}
