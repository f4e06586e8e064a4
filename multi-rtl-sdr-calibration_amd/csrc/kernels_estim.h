// kernels_estim.h -- estimators on gathered windows and the per-stream decision kernels.
//
//   k_burst_tone  gather + spectrum argmax + tone-frequency estimator of FCCH_fine_correction.m:148-155
//                 (+ SNR gate :185-189) and carrier_correct_post_SCH.m:63-72, one workgroup per burst
//   k_window_sch  gather + SCH_corr_rate_correction.m:50-55: |sch_ts' * window|^2 for the 89 offsets
//   d_*_setup / d_*_decide   the integer / ppm logic of the reference functions, one thread per stream on an LDS
//                 copy of its state, writing the next stage's window list; run by stream_tail in the last
//                 per-window workgroup of the stream (or by k_step where a stage has no per-window kernel)
//   k_fine_verify the global wrapper of fine_verify_body (kernels_detect.h) + FINE_DECIDE tail
#pragma once
#include "state.h"
#include "kernels_frontend.h"
#include "kernels_detect.h"

#define GSM_SYMBOL_RATE ((1625.0 / 6.0) * 1e3)
#define TWO_PI_D (2.0 * 3.14159265358979323846)
#define PI_D 3.14159265358979323846

// merge the NB partial peaks of one window: larger p wins, equal p -> smaller tie key
__device__ __forceinline__ PeakOut merge_peaks(const PeakOut* p, int NB) {
    PeakOut b = p[0];
    for (int i = 1; i < NB; ++i)
        if (p[i].p > b.p || (p[i].p == b.p && p[i].tie < b.tie)) b = p[i];
    return b;
}

// Results one workgroup hands to ANOTHER workgroup of the same kernel (stream_tail below) go through agent-scope
// relaxed atomics: write-through past the XCD's L2 on the store side, cache-bypassing on the load side.  The
// MI355X has eight L2s; a device-wide fence per workgroup (write back + invalidate) was measured at ~100 us per
// kernel, these cost nothing.
__device__ __forceinline__ void coherent_store(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long coherent_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ PeakOut coherent_peak(const PeakOut* p) {
    static_assert(sizeof(PeakOut) == 16, "two 64-bit words");
    const unsigned long long a = coherent_load((const unsigned long long*)p), b = coherent_load((const unsigned long long*)p + 1);
    PeakOut o;
    o.p = __longlong_as_double((long long)a);
    o.tie = (int)(b & 0xffffffffu);
    o.k = (int)(b >> 32);
    return o;
}
__device__ __forceinline__ PeakOut merge_peaks_coherent(const PeakOut* p, int NB) {
    PeakOut b = coherent_peak(p);
    for (int i = 1; i < NB; ++i) {
        const PeakOut q = coherent_peak(p + i);
        if (q.p > b.p || (q.p == b.p && q.tie < b.tie)) b = q;
    }
    return b;
}

// ------------------------------------------------------------------------------------------------
// k_burst_tone<GATE>: one workgroup per FCCH burst does the whole per-burst estimate without leaving
// LDS: gather the burst window at level a.level (raw -> raw2iq -> FIR -> lerp [-> mix -> lerp]),
// 1184-point spectrum argmax in fftshift order, integer-bin derotation, unit-phasor phase step
// (FCCH_fine_correction.m:148-155 / carrier_correct_post_SCH.m:63-72) and, for GATE, the SNR gate
// (:185-189).  grid (H, S), block 512.  Writes st->fo_burst[w] and (GATE) st->snr_burst[w].
// LDS: the gather carve (kernels_frontend.h); the window ends up in buf0 or buf1, the other buffer then
// holds B[37][N2+1]; w37 | wN2 | P[2*hnl] follow the carve.  ~48 KB: three workgroups per CU.
// ------------------------------------------------------------------------------------------------
// Issue priority of this wave by the dispatch round of its workgroup (workgroups 0..255 / 256..511 / 512..: the first, second
// and third to arrive on their CU).  The SIMDs issue oldest wave first, so of the three workgroups that share a CU the first
// one dispatched runs ahead: in-kernel stamps show the streams of round 0 through every stage of k_post_chain_r 6-8 us before
// those of round 2 (done at 45 against 57 us), and the launch ends with the slowest stream.  s_setprio overrides the age order.
// Eight schedules tried (first half of the burst stage | its gate | SCH stage | last stage; E = oldest first, L = youngest
// first): none 66.4 us, LLEE 64.8, LELE 64.4, LEEL 67.8, EELL 64.9, ELEL 66.2, ELLL 62.7, ELLE 62.0 -- the one used.
#define PCR_PRIO(P0, P1, P2) { const unsigned rnd_ = (blockIdx.y * gridDim.x + blockIdx.x) >> 8; if (rnd_ == 0) __builtin_amdgcn_s_setprio(P0); else if (rnd_ == 1) __builtin_amdgcn_s_setprio(P1); else __builtin_amdgcn_s_setprio(P2); }
#define BT_THREADS 512

// ------------------------------------------------------------------------------------------------
// burst_gather_fast: gather_core for the case k_post_chain_r's burst stages nearly always see -- the stream's state is
// this workgroup's LDS copy, level 0 of the burst lies inside a window the fine search filtered (a.l0), and the chain
// above it is LERP (level 1) or LERP | MIX | LERP-or-COPY (level 3).  Same arithmetic per sample as gather_core (same
// expressions in the same order: the two give identical bits), organised around what bounds a workgroup here, the
// number of barrier-separated phases and of round trips to L2:
//   * every wave works the plan out itself from LDS (no plan phase);
//   * level 1 is interpolated straight from the window buffer (no LDS copy of level 0), with the loads of the burst
//     stage's nfft-point twiddle table in flight at the same time;
//   * level 3: the MIX rotator is applied to the two level-1 samples of each level-3 output as they are read (no level-2
//     buffer) -- two phases where gather_core has six.
// Returns 1: window in smem (buf0), twiddle table in the other buffer, NO trailing barrier; 0: not this case (the caller
// falls back to gather_core); -1: this workgroup has no window.  All three block-uniform.
// ------------------------------------------------------------------------------------------------
// a wave-uniform value (read from LDS, so in a vector register as far as the compiler knows) moved to scalar registers
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long uni_l(long v) {
    return (long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)v >> 32)) << 32) |
                  (unsigned)__builtin_amdgcn_readfirstlane((int)v));
}
__device__ __forceinline__ double uni_d(double v) { return __longlong_as_double(uni_l(__double_as_longlong(v))); }
template <int NT, int KID = -1>
__device__ __forceinline__ int burst_gather_fast(const StreamState* __restrict__ st, const GatherArgs& a, unsigned char* smem,
                                                 int widx, int s, const cplx* __restrict__ tw_g, int nfft) {
    const int tid = threadIdx.x, level = a.level;
    if (!a.l0 || a.src_kind != SRC_RAW || a.tiles || (level != 1 && level != 3) || nfft > 3 * NT || nfft > a.len + 40) return 0;
    if (widx >= MAXH) return -1;
    // every field of the plan in one batch of LDS reads (the generic pointer is this workgroup's LDS copy: addressed as
    // LDS the reads are ds_read, not flat loads that take the vector-memory path to find that out), decisions afterwards
    typedef const __attribute__((address_space(3))) StreamState* LdsState;
    LdsState sl = (LdsState)st;
    const int lane = tid & 63;
    const int v_nwin = sl->n_win, v_t1 = sl->op[1].type, v_t2 = sl->op[2].type, v_t3 = sl->op[3].type, v_nf = sl->n_fine_ws;
    const double v_f1 = sl->op[1].param, v_c2 = sl->op[2].param, v_f3 = sl->op[3].param;
    const long v_ws = sl->win_start[widx], v_n2 = sl->op[2].n, v_n0 = sl->n0;
    const long fw = lane < MAXH ? sl->fine_ws[lane] : 0;
    if (widx >= uni_i(v_nwin)) return -1;
    if (uni_i(v_t1) != OP_LERP) return 0;
    const double f1 = uni_d(v_f1);
    int t3 = OP_NONE;
    double c2 = 0.0, f3 = 1.0;
    if (level == 3) {
        t3 = uni_i(v_t3);
        if (uni_i(v_t2) != OP_MIX || (t3 != OP_LERP && t3 != OP_COPY)) return 0;
        c2 = uni_d(v_c2);
        f3 = uni_d(v_f3);
    }
    const int L = a.len;
    const long lo3 = uni_l(v_ws), hi3 = lo3 + L - 1;
    long lo1 = lo3, hi1 = hi3;                                   // level 1 (and 2: a MIX keeps the range)
    if (level == 3 && t3 == OP_LERP) {
        lo1 = (long)floor((double)lo3 * f3);
        const long h = (long)floor((double)hi3 * f3) + 1, n2 = uni_l(v_n2);
        hi1 = h > n2 - 1 ? n2 - 1 : h;
    }
    const long lo0 = (long)floor((double)lo1 * f1);
    long hi0 = (long)floor((double)hi1 * f1) + 1;
    {
        const long n0 = uni_l(v_n0);
        hi0 = hi0 > n0 - 1 ? n0 - 1 : hi0;
    }
    const int cnt1 = (int)(hi1 - lo1 + 1);
    const GatherCarve gc = gather_carve(a.len, a.level, a.src_kind, a.ntaps, true, a.pad != 0);
    if (level == 3 && (cnt1 > 32 * GC_ROT_A || cnt1 > (int)gc.bufn || cnt1 <= 0)) return 0;
    // the fine window holding [lo0, hi0] (every wave finds it itself)
    int h0;
    long off0;
    {
        const bool in = lane < uni_i(v_nf) && lane < MAXH;
        const unsigned long long m = __ballot(in && fw <= lo0 && hi0 < fw + a.l0_len);
        if (!m) return 0;
        h0 = __ffsll((long long)m) - 1;
        off0 = lo0 - uni_l(__shfl(fw, h0, 64));
    }
    const cplx* __restrict__ x = a.l0 + (size_t)s * a.l0_stream_stride + (size_t)h0 * a.l0_win_stride + off0;   // x[i]: level-0 sample lo0+i
    cplx* buf0 = (cplx*)smem;
    cplx* buf1 = (cplx*)(smem + gc.off_region1);
    auto lerp1 = [&](int i) {                                    // level-1 sample lo1+i
        const long k = lo1 + i;
        const double xq = (double)k * f1;                        // interp_seq = (0:max_len-1)'.*(1+e)
        const long i0 = (long)floor(xq);
        const long i1 = i0 + 1 > hi0 ? hi0 : i0 + 1;             // beyond the last sample the weight is 0
        const double t = xq - (double)i0;
        const cplx v0 = x[i0 - lo0], v1 = x[i1 - lo0];
        return make_double2(v0.x + t * (v1.x - v0.x), v0.y + t * (v1.y - v0.y));
    };
    DEV_STAMP(KID, blockIdx.y * gridDim.x + blockIdx.x, 5);
    if (level == 1) {
        for (int n = tid; n < nfft; n += NT) buf1[n] = tw_g[n];
        DEV_STAMP(KID, blockIdx.y * gridDim.x + blockIdx.x, 6);
        for (int i = tid; i < L; i += NT) buf0[i] = lerp1(i);
        return 1;
    }
    // level 3
    cplx* T = (cplx*)(smem + gc.off_rot);                        // S | A[] | B[] of the level-2 rotator (gather_core's table)
    const int na = (cnt1 + 31) >> 5;
    for (int i = tid; i < cnt1; i += NT) buf1[i] = lerp1(i);
    for (int i = tid; i < 1 + na + 32; i += NT) {
        const double arg = i == 0 ? (double)lo1 * c2 : (i <= na ? (double)(32 * (i - 1)) * c2 : (double)(i - 1 - na) * c2);
        double sn, cs;
        sincos_large(arg, &sn, &cs);
        T[i == 0 ? 0 : (i <= na ? i : 1 + GC_ROT_A + (i - 1 - na))] = make_double2(cs, sn);
    }
    DEV_STAMP(KID, blockIdx.y * gridDim.x + blockIdx.x, 6);
    __syncthreads();
    DEV_STAMP(KID, blockIdx.y * gridDim.x + blockIdx.x, 7);
    // the burst stage's twiddle table goes where level 1 is now: loaded here, stored after the barrier below.  (Clamped
    // unconditional loads: with the three loads predicated the register allocator spills 150 vector registers to scratch,
    // 880 B/lane, and a frame that size halves the speed of every kernel on the queue.)
    const cplx tw0 = tw_g[min(tid, nfft - 1)], tw1 = tw_g[min(tid + NT, nfft - 1)], tw2 = tw_g[min(tid + 2 * NT, nfft - 1)];
    auto lvl2 = [&](int i) {                                     // level-2 sample lo1+i: exp(1i*k*comp_phase_rotate) from the table
        return cmul(buf1[i], cmul(cmul(T[0], T[1 + (i >> 5)]), T[1 + GC_ROT_A + (i & 31)]));
    };
    if (t3 == OP_LERP) {
#pragma unroll 1
        for (int i = tid; i < L; i += NT) {
            const long k = lo3 + i;
            const double xq = (double)k * f3;
            const long i0 = (long)floor(xq);
            const long i1 = i0 + 1 > hi1 ? hi1 : i0 + 1;
            const double t = xq - (double)i0;
            const cplx v0 = lvl2((int)(i0 - lo1)), v1 = lvl2((int)(i1 - lo1));
            buf0[i] = make_double2(v0.x + t * (v1.x - v0.x), v0.y + t * (v1.y - v0.y));
        }
    } else {
        for (int i = tid; i < L; i += NT) buf0[i] = lvl2(i);
    }
    __syncthreads();                                             // level 1 is dead
    if (tid < nfft) buf1[tid] = tw0;
    if (tid + NT < nfft) buf1[tid + NT] = tw1;
    if (tid + 2 * NT < nfft) buf1[tid + 2 * NT] = tw2;
    return 1;
}
#define BT_STAMP(i) DEV_STAMP(GATE ? KID_BT1 : KID_BT0, blockIdx.y * gridDim.x + blockIdx.x, i)
template <int GATE, int FIR_UNR = 1>
__device__ __forceinline__ void burst_tone_body(StreamState* __restrict__ sts, const GatherArgs& a, int nfft,
                                                const cplx* __restrict__ tw_g, int ov, int prior_mode,
                                                unsigned char* smem, double* res = nullptr,    // res: {fo, snr} instead of the stores into the state
                                                bool state_in_lds = false) {                   // sts + blockIdx.y is this workgroup's LDS copy
    __shared__ double red_p[BT_THREADS / 64];
    __shared__ int red_t[BT_THREADS / 64];
    __shared__ double red[2 * (BT_THREADS / 64)];
    __shared__ double sh_phase;
    __shared__ int sh_key;
    const int s = blockIdx.y, w = blockIdx.x, tid = threadIdx.x;
    BT_STAMP(0);
    const int fast = state_in_lds ? burst_gather_fast<BT_THREADS, GATE ? KID_BT1 : KID_BT0>(sts + s, a, smem, w, s, tw_g, nfft) : 0;
    if (fast < 0) return;                                           // block-uniform, like the two below
    cplx* xs = (cplx*)smem;                                         // nfft samples of the burst, in LDS
    if (!fast) {
        xs = gather_core<BT_THREADS, GATE ? KID_BT1 : KID_BT0, FIR_UNR>(sts, a, smem, w, s, true);
        if (!xs) return;
        __syncthreads();
    }
    BT_STAMP(1);
    StreamState* st = sts + s;
    const int N2 = nfft / 37, ldb = N2 + 1;
    // the window sits in one of the two gather buffers; the other one holds B, the tables go behind the carve
    const GatherCarve gc = gather_carve(a.len, a.level, a.src_kind, a.ntaps, true, a.pad != 0);
    cplx* B = (xs == (cplx*)smem) ? (cplx*)(smem + gc.off_region1) : (cplx*)smem;
    cplx* w37 = (cplx*)(smem + gc.total);
    cplx* wN2 = w37 + 40;
    double* P = (double*)(wN2 + N2);
    // the nfft-point twiddle table, staged in the idle buffer B until B is needed (the candidate DFTs and the rotation
    // below index it data-dependently: from global memory every step of those loops was a round trip to L2)
    cplx* twl = B;
    if (!fast) for (int n = tid; n < nfft; n += BT_THREADS) twl[n] = tw_g[n];
    fft37_tables(w37, wN2, N2, tid, tw_g);
    __syncthreads();
    // ---- spectrum argmax, first max in fftshift order (:149-150) ----
    // Fast exact route: the window is centred on the FCCH tone, so nearly all of its energy sits in a few bins
    // around a known prior (the fine search's winning bin; bin 37 = symbol_rate/4 once the carrier is
    // corrected).  Evaluate the 7 bins prior-3..prior+3 directly (one wave each) and use Parseval,
    //   sum_k |X_k|^2 = N * sum_n |x[n]|^2,
    // to bound every other bin by R = N*E - sum(candidates); if R < max(candidates) the global first-max is
    // among the candidates.  Otherwise (weak or absent tone) fall back to the full 37 x N2 spectrum.
    {
        double e = 0.0;
        for (int n = tid; n < nfft; n += BT_THREADS) { const cplx v = xs[n]; e = fma(v.x, v.x, fma(v.y, v.y, e)); }
        e = wave_sum(e);
        if ((tid & 63) == 0) red[tid >> 6] = e;
        const int wave = tid >> 6, lane = tid & 63;
        const int prior = prior_mode == 1 ? st_i32(&st->prior_bin[w]) : nfft / 32;       // 37 for every oversampling ratio
        if (wave < 7) {
            int k = prior - 3 + wave;
            k = ((k % nfft) + nfft) % nfft;
            double ar = 0.0, ai = 0.0;
            int idx = (int)(((unsigned)k * (unsigned)lane) % (unsigned)nfft);      // (k < nfft <= 2^24: no overflow)
            const int stp = (int)(((unsigned)k * 64u) % (unsigned)nfft);
#pragma unroll 4
            for (int n = lane; n < nfft; n += 64) {
                const cplx v = xs[n], t = twl[idx];
                ar = fma(v.x, t.x, fma(-v.y, t.y, ar));
                ai = fma(v.x, t.y, fma(v.y, t.x, ai));
                idx += stp;
                if (idx >= nfft) idx -= nfft;
            }
            ar = wave_sum(ar);
            ai = wave_sum(ai);
            if (lane == 0) { red_p[wave] = ar * ar + ai * ai; red_t[wave] = (k + nfft / 2) % nfft; }
        }
        __syncthreads();
        if (tid == 0) {
            double etot = 0.0;
            for (int i = 0; i < BT_THREADS / 64; ++i) etot += red[i];
            double best = -1.0, sum = 0.0;
            int key = 0x7fffffff;
            for (int i = 0; i < 7; ++i) {
                sum += red_p[i];
                if (red_p[i] > best || (red_p[i] == best && red_t[i] < key)) { best = red_p[i]; key = red_t[i]; }
            }
            const double R = (double)nfft * etot - sum;           // energy left for all the other bins together
            sh_key = (best > 0.0 && R * 1.000001 + 1e-300 < best) ? key : -1;
        }
        __syncthreads();
    }
    const bool full = sh_key < 0;
    if (full) {                                                    // block-uniform: proof failed, full spectrum (B: table gone)
        fft37_step1(xs, B, w37, tw_g, nfft, N2, ldb, tid, BT_THREADS);
        __syncthreads();
        double best = -1.0;
        int key = 0x7fffffff;
        for (int k = tid; k < nfft; k += BT_THREADS) {
            const cplx X = fft37_step2_bin(B, wN2, N2, ldb, k);
            const double p = X.x * X.x + X.y * X.y;
            const int sk = (k + nfft / 2) % nfft;
            if (p > best || (p == best && sk < key)) { best = p; key = sk; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double op = __shfl_down(best, off, 64);
            const int ok = __shfl_down(key, off, 64);
            if (op > best || (op == best && ok < key)) { best = op; key = ok; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { red_p[tid >> 6] = best; red_t[tid >> 6] = key; }
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < BT_THREADS / 64; ++i)
                if (red_p[i] > best || (red_p[i] == best && red_t[i] < key)) { best = red_p[i]; key = red_t[i]; }
            sh_key = key;
        }
        __syncthreads();
    }
    BT_STAMP(2);
    const int max_idx = sh_key + 1;                                    // 1-based index after fftshift
    const double sampling_rate = GSM_SYMBOL_RATE * (double)ov;
    // :151  int_phase_rotate = 2.*pi.*(max_idx - ((fft_len/2)+1))./fft_len
    const double ipr = (TWO_PI_D * (double)(max_idx - (nfft / 2 + 1))) / (double)nfft;
    // :152  fcch_mat .* exp(-1i.*((0:fft_len-1)')*int_phase_rotate).  int_phase_rotate is an
    // integer number of bins j, so exp(-1i*n*ipr) = exp(-2*pi*i*(n*j mod N)/N): taken from the exact table
    // (the reference's fl(n*ipr) differs from it by < 1e-12 rad, far below what the estimator resolves).
    // :153-154  mean( exp(1i*angle(x(2:end))) ./ exp(1i*angle(x(1:end-1))) ): unit phasors u[n] = x/|x|, then
    // u[n+1]*conj(u[n]) (the quotient of two unit phasors).  One pass: every lane derotates and normalises the two
    // samples of its product itself (each u[n] is formed twice, identically) -- a rotated copy of the burst and a buffer
    // of phasors cost two more passes over LDS and two barriers; xs stays as gathered (the gate below rotates on read).
    const int jb = max_idx - (nfft / 2 + 1);
    const unsigned jm = (unsigned)(jb < 0 ? jb + nfft : jb);
    double sr = 0.0, si = 0.0;
    {
        auto unit = [](const cplx& p0) {
            const double m2 = p0.x * p0.x + p0.y * p0.y;
            const double inv = rsqrt(m2);
            return m2 > 0.0 ? make_double2(p0.x * inv, p0.y * inv) : make_double2(1.0, 0.0);   // angle(0) = 0
        };
        auto pass = [&](const cplx* tw) {
            for (int n = tid; n < nfft - 1; n += BT_THREADS) {
                const unsigned i0 = ((unsigned)n * jm) % (unsigned)nfft;
                const unsigned i1 = i0 + jm >= (unsigned)nfft ? i0 + jm - (unsigned)nfft : i0 + jm;
                const cplx ub = unit(cmul(xs[n], tw[i0]));
                const cplx ua = unit(cmul(xs[n + 1], tw[i1]));
                sr += ua.x * ub.x + ua.y * ub.y;
                si += ua.y * ub.x - ua.x * ub.y;
            }
        };
        if (full) pass(tw_g); else pass(twl);
    }
    sr = wave_sum(sr);
    si = wave_sum(si);
    if ((tid & 63) == 0) { red[2 * (tid >> 6)] = sr; red[2 * (tid >> 6) + 1] = si; }
    __syncthreads();
    if (tid == 0) {
        double tr = 0.0, ti = 0.0;
        for (int i = 0; i < BT_THREADS / 64; ++i) { tr += red[2 * i]; ti += red[2 * i + 1]; }
        const double cnt = (double)(nfft - 1);
        const double phase = atan2(ti / cnt, tr / cnt);
        sh_phase = phase;
        const double fo = sampling_rate * (ipr + phase) / TWO_PI_D;                    // :155
        if (res) res[0] = fo; else coherent_store(&st->fo_burst[w], fo);
    }
    BT_STAMP(3);
    if (!GATE) return;
    if (state_in_lds) PCR_PRIO(0, 1, 2)                               // k_post_chain_r: from here through the SCH stage the last-dispatched workgroups issue first
    __syncthreads();
    // ---- SNR gate, FCCH_fine_correction.m:185-189: bins [0,hnl) and [nfft-hnl,nfft) only ----
    // exp(-1i*n*phase) = base[n/16] * pw[n%16]: accurate sincos only for the 16 powers and every 16th sample
    // (the gate is a 5 dB threshold on band powers; 1e-16-level phase differences cannot move it)
    const double phase = sh_phase;
    cplx* pw = (cplx*)P;                       // 16 entries; P proper is written after the barrier below
    cplx* base = pw + 16;                      // nfft/16 + 1 entries
    if (tid < 16) {
        double sn, cs;
        sincos((double)tid * phase, &sn, &cs);
        pw[tid] = make_double2(cs, -sn);
    } else if (tid >= 64 && tid < 64 + (nfft + 15) / 16) {
        double sn, cs;
        sincos((double)((tid - 64) * 16) * phase, &sn, &cs);
        base[tid - 64] = make_double2(cs, -sn);
    }
    __syncthreads();
    // both rotations (:152's bin shift and this one) are applied as the 37-point step first reads each sample
    if (full) {
        fft37_step1_sym<true>(xs, B, w37, tw_g, nfft, N2, ldb, tid, BT_THREADS, [&](int n, const cplx& v) {
            return cmul(cmul(v, tw_g[((unsigned)n * jm) % (unsigned)nfft]), cmul(base[n >> 4], pw[n & 15])); });
    } else {
        fft37_step1_sym<true>(xs, B, w37, tw_g, nfft, N2, ldb, tid, BT_THREADS, [&](int n, const cplx& v) {
            return cmul(cmul(v, twl[((unsigned)n * jm) % (unsigned)nfft]), cmul(base[n >> 4], pw[n & 15])); });
    }
    __syncthreads();
    const int hnl = (int)ceil(((double)nfft * 200e3 / sampling_rate) / 2.0);     // :22 half_noise_len
    const int nb = 2 * hnl;
    for (int b = tid; b < nb; b += BT_THREADS) {
        const int k = b < hnl ? b : nfft - nb + b;
        const cplx X = fft37_step2_bin(B, wN2, N2, ldb, k);
        const double m = hypot(X.x, X.y);
        P[b] = m * m;                               // P[b]: b < hnl -> bin b; else bin nfft-nb+b
    }
    __syncthreads();
    if (tid < 64) {
        // signal: fd([1:3, end-1:end]); noise: fd([4:hnl, end-hnl+1:end-2])   (1-based).  The ~100 noise bins are
        // added by one wave (a tree instead of a serial loop: the value moves by ~1e-16 relative, it feeds a 5 dB gate)
        double noi = 0.0;
        for (int k = 3 + tid; k < nb - 2; k += 64) noi += P[k];
        noi = wave_sum(noi);
        if (tid == 0) {
            double sig = 0.0;
            for (int k = 0; k < 3; ++k) sig += P[k];
            for (int k = nb - 2; k < nb; ++k) sig += P[k];
            const double sn = 10.0 * log10(sig / noi);
            if (res) res[1] = sn; else coherent_store(&st->snr_burst[w], sn);
        }
    }
    BT_STAMP(4);
}

// ------------------------------------------------------------------------------------------------
// k_window_sch: one workgroup per FCCH gathers its SCH search window (nshift-1+len_ts samples at
// level a.level) into LDS and correlates it with the training sequence at the nshift offsets
// (SCH_corr_rate_correction.m:45-55); 4 lanes per offset.  grid (H, S), block 512.
// Stores SCH_pos(i) = sp + max_idx - 1 into st->sch_first[w]; an edge peak sets st->sch_edge (:59).
// ------------------------------------------------------------------------------------------------
#define SCH_PARTS 4
// ------------------------------------------------------------------------------------------------
// The correlation for the drivers' geometry (NSH = 89 offsets, LT = 512 taps), organised around LDS traffic (round 4: at
// 1 024 streams the chain's time is the SUM of its VALU and LDS pipe times, and this loop -- two 16-byte LDS reads per
// complex MAC -- was a ninth of all LDS cycles).  A lane takes R = 4 consecutive offsets and one of Q = 16 stretches of M = 39
// window samples: per sample it reads the sample (the same address in every lane of its stretch: a broadcast) and ONE new tap
// (its four offsets see the taps sliding by one per sample), and issues four complex MACs -- 2 reads per 16 FMAs instead of
// 2 per 4.  conj(ts) is staged zero-padded on both sides, so taps outside the sequence multiply as zeros and no lane tests
// an index; the stretches of one offset group are the 16 lanes of a DPP row, which adds them up without LDS.  The 39-sample
// stride (624 B) puts the 16 lanes of a row on 16 different bank groups for both reads.  cv[o] = |sch_ts' * window_o|^2.
// tcp: (R*NG + Q*M) entries; xs_w: the window, writable up to Q*M entries (the gather buffers have len + 40).
// ------------------------------------------------------------------------------------------------
template <int NSH, int LT>
__device__ __forceinline__ void sch_corr_rows(cplx* __restrict__ xs_w, const cplx* __restrict__ ts, cplx* __restrict__ tcp,
                                              double* __restrict__ cv, int tid, int nthreads) {
    constexpr int R = 4, Q = 16, M = 39;
    constexpr int NG = (NSH + R - 1) / R, PADL = R * NG, NTC = PADL + Q * M, WL = NSH - 1 + LT;
    static_assert(Q * M >= WL && Q * M <= WL + 40, "the stretches cover the window and stay inside the gather buffer's slack");
    for (int i = tid; i < NTC; i += nthreads) {
        const int n = i - PADL;
        tcp[i] = (n >= 0 && n < LT) ? make_double2(ts[n].x, -ts[n].y) : make_double2(0.0, 0.0);
    }
    for (int i = WL + tid; i < Q * M; i += nthreads) xs_w[i] = make_double2(0.0, 0.0);   // finite values behind the window (their taps are zeros)
    __syncthreads();
    if (tid < NG * Q) {
        const int g = tid >> 4, q = tid & 15;
        const cplx* __restrict__ xp = xs_w + q * M;
        const cplx* __restrict__ tp = tcp + PADL + q * M - R * g;     // tp[s - j]: tap of offset 4g+j at sample q*M + s
        cplx T1 = tp[-1], T2 = tp[-2], T3 = tp[-3];
        double ar0 = 0.0, ai0 = 0.0, ar1 = 0.0, ai1 = 0.0, ar2 = 0.0, ai2 = 0.0, ar3 = 0.0, ai3 = 0.0;
#define SCH_MAC(AR, AI, C, V) AR = fma(C.x, V.x, fma(-C.y, V.y, AR)); AI = fma(C.x, V.y, fma(C.y, V.x, AI));
#define SCH_STEP(S, TA, TB, TC, TD) { const cplx xv = xp[S]; TA = tp[S]; SCH_MAC(ar0, ai0, TA, xv) SCH_MAC(ar1, ai1, TB, xv) SCH_MAC(ar2, ai2, TC, xv) SCH_MAC(ar3, ai3, TD, xv) }
        cplx T0;
        int s = 0;
#pragma unroll 1
        for (; s + 4 <= M; s += 4) {                                  // four samples per trip: the tap window rotates by renaming
            SCH_STEP(s, T0, T1, T2, T3)
            SCH_STEP(s + 1, T3, T0, T1, T2)
            SCH_STEP(s + 2, T2, T3, T0, T1)
            SCH_STEP(s + 3, T1, T2, T3, T0)
        }
        static_assert(M % 4 == 3, "three samples left");
        SCH_STEP(s, T0, T1, T2, T3)
        SCH_STEP(s + 1, T3, T0, T1, T2)
        SCH_STEP(s + 2, T2, T3, T0, T1)
#undef SCH_STEP
#undef SCH_MAC
        // the 16 stretches of an offset group are one DPP row: xor 1, xor 2, half-row mirror, row mirror leave the row's total in every lane
#define SCH_ROWSUM(V) V += dpp_move_f64<0xB1, 0xF>(V); V += dpp_move_f64<0x4E, 0xF>(V); V += dpp_move_f64<0x141, 0xF>(V); V += dpp_move_f64<0x140, 0xF>(V);
        SCH_ROWSUM(ar0) SCH_ROWSUM(ai0) SCH_ROWSUM(ar1) SCH_ROWSUM(ai1) SCH_ROWSUM(ar2) SCH_ROWSUM(ai2) SCH_ROWSUM(ar3) SCH_ROWSUM(ai3)
#undef SCH_ROWSUM
        const double sr = q == 0 ? ar0 : (q == 1 ? ar1 : (q == 2 ? ar2 : ar3)), si = q == 0 ? ai0 : (q == 1 ? ai1 : (q == 2 ? ai2 : ai3));
        const int o = R * g + q;
        if (q < R && o < NSH) {
            const double m = hypot(sr, si);
            cv[o] = m * m;                                            // :53 abs(...).^2
        }
    }
    __syncthreads();
}

// FAST: the drivers' geometry (11*8+1 = 89 offsets, 512-sample training sequence): sch_corr_rows
template <int FIR_UNR = 1, bool FAST = false>
__device__ __forceinline__ void window_sch_body(StreamState* __restrict__ sts, const GatherArgs& a,
                                                const cplx* __restrict__ ts, int len_ts, int nshift,
                                                unsigned char* smem, double* res = nullptr) {   // res: {SCH_pos, edge flag}
    const int s = blockIdx.y, w = blockIdx.x, tid = threadIdx.x;
    DEV_STAMP(KID_SCH, blockIdx.y * gridDim.x + blockIdx.x, 0);
    cplx* xs = gather_core<512, -1, FIR_UNR>(sts, a, smem, w, s, true);
    if (!xs) return;
    DEV_STAMP(KID_SCH, blockIdx.y * gridDim.x + blockIdx.x, 1);
    const GatherCarve gc = gather_carve(a.len, a.level, a.src_kind, a.ntaps, true, a.pad != 0);
    cplx* tc = (cplx*)(smem + gc.total);                                  // conj(ts), behind the gather carve
    cplx* part = tc + len_ts;                                             // nshift * SCH_PARTS partial sums
    double* cv = (double*)(part + nshift * SCH_PARTS);                    // nshift correlation powers
    __syncthreads();
    StreamState* st = sts + s;
    if (FAST) {
        // (its zero-padded tap copy, 92 + 624 entries, and the 89 powers take less room than tc | part | cv above)
        sch_corr_rows<89, 512>(xs, ts, tc, cv, tid, 512);
    } else {
    for (int i = tid; i < len_ts; i += 512) tc[i] = make_double2(ts[i].x, -ts[i].y);
    __syncthreads();
    const int seg = (len_ts + SCH_PARTS - 1) / SCH_PARTS;
    // (shift, part) tasks, one per thread and round.  When whole rounds leave one or two shifts over (8x oversampling:
    // 129 shifts x 4 parts = one round of 512 + 4 tasks), those are not given a nearly empty round of their own: the whole
    // block forms each of their sums together.
    const int full = (nshift * SCH_PARTS / 512) * (512 / SCH_PARTS);          // shifts covered by whole rounds
    const int n_task_sh = (full > 0 && nshift - full <= 2) ? full : nshift;
    __shared__ double rest[2][2 * 8];
    for (int t = tid; t < n_task_sh * SCH_PARTS; t += 512) {
        const int o = t % n_task_sh, q = t / n_task_sh;   // consecutive lanes: consecutive offsets (conflict-free reads)
        const int n0 = q * seg, n1 = n0 + seg < len_ts ? n0 + seg : len_ts;
        double ar = 0.0, ai = 0.0;
#pragma unroll 8
        for (int n = n0; n < n1; ++n) {
            const cplx c = tc[n], v = xs[o + n];
            ar = fma(c.x, v.x, fma(-c.y, v.y, ar));
            ai = fma(c.x, v.y, fma(c.y, v.x, ai));
        }
        part[q * nshift + o] = make_double2(ar, ai);
    }
    for (int o = n_task_sh; o < nshift; ++o) {            // (block-uniform) a left-over shift: 512 products at a time, per-wave sums
        double ar = 0.0, ai = 0.0;
        for (int n = tid; n < len_ts; n += 512) {
            const cplx c = tc[n], v = xs[o + n];
            ar = fma(c.x, v.x, fma(-c.y, v.y, ar));
            ai = fma(c.x, v.y, fma(c.y, v.x, ai));
        }
        ar = wave_sum(ar);
        ai = wave_sum(ai);
        if ((tid & 63) == 0) { rest[o - n_task_sh][2 * (tid >> 6)] = ar; rest[o - n_task_sh][2 * (tid >> 6) + 1] = ai; }
    }
    __syncthreads();
    for (int o = tid; o < nshift; o += 512) {
        double ar = 0.0, ai = 0.0;
        if (o < n_task_sh) {
            for (int q = 0; q < SCH_PARTS; ++q) { ar += part[q * nshift + o].x; ai += part[q * nshift + o].y; }
        } else {
            for (int q = 0; q < 8; ++q) { ar += rest[o - n_task_sh][2 * q]; ai += rest[o - n_task_sh][2 * q + 1]; }
        }
        const double m = hypot(ar, ai);
        cv[o] = m * m;                      // :53 abs(...).^2
    }
    __syncthreads();
    }
    DEV_STAMP(KID_SCH, blockIdx.y * gridDim.x + blockIdx.x, 2);
    if (tid < 64) {                                       // first maximum over the offsets: one wave, then a shuffle tree
        int mi = 0x7fffffff;
        double mx = -1.0;
        for (int o = tid; o < nshift; o += 64)
            if (cv[o] > mx) { mx = cv[o]; mi = o; }       // (ascending o per lane: strict > keeps the first)
        for (int off = 32; off > 0; off >>= 1) {
            const double om = __shfl_down(mx, off, 64);
            const int oi = __shfl_down(mi, off, 64);
            if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
        }
        if (tid == 0) {
            const double sp = (double)(st_i64(&st->win_start[w]) + 1 + mi);   // sp + max_idx - 1
            const bool edge = mi == 0 || mi == nshift - 1;                    // :59
            if (res) { res[0] = sp; res[1] = edge ? 1.0 : 0.0; }
            else {
                coherent_store(&st->sch_first[w], sp);
                if (edge) atomicOr(&st->sch_edge, 1);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// decision steps: the reference's per-stream control logic between the transforms.  Every function here is called by
// all 64 lanes of ONE wave with the stream's state in LDS (or, for d_fine_setup at the end of k_coarse_scan, in that
// kernel's LDS copy).  Control flow is wave-uniform (it depends on the shared state only); the loops over hits / rows
// of the reference run one element per lane, their `break` / early `return` exits resolved with ballots; scalars are
// written by lane 0.  A loop that sums floating-point values keeps the reference's order (every lane adds the same
// LDS values serially).  The integer-valued position grids (acc += 10 or 11 frames, :127-133) are exact in any order.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wsync() {      // lane-to-lane hand-over through LDS inside one wave
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
#define LANE0(...) do { if (lane == 0) { __VA_ARGS__; } } while (0)
__device__ __forceinline__ unsigned long long bits_below(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }

__device__ __forceinline__ void fine_sentinel(StreamState* st) {
    st->n_fcch = 0; st->fcch_is_sentinel = 1;
    st->sampling_ppm1 = INFINITY; st->carrier_ppm1 = INFINITY;
    st->r1_kind = 0; st->n_win = 0; st->n_fine = 0; st->n_fine_ws = 0;
}

// FCCH_fine_correction.m:8-46 -- window list for the fine search (level `lvl`)
__device__ void d_fine_setup(StreamState* st, int ov, int lvl, int min_hits, int lane) {
    LANE0(fine_sentinel(st));
    if (lane >= 1 && lane < NLEVELS && lane > lvl) { st->op[lane].type = OP_NONE; st->op[lane].n = 0; }
    if (st->status < 0) return;
    if (st->n_coarse < min_hits) { LANE0(set_status(st, 0, GSMCAL_S_FEW_HITS)); return; }   // :12
    const long len_s_ov = level_len(st, lvl);
    const long len_s = len_s_ov / ov;                                          // :28
    const int max_offset = 64, len_cw = 148;
    const int nc = st->n_coarse < MAXH ? st->n_coarse : MAXH;
    const bool in = lane < nc;
    const long position = in ? (long)st->coarse_pos[lane] : 0;
    const bool brk = in && position + max_offset > len_s - len_cw + 1;         // :35 (the loop stops at the first one)
    const long sp = (position - max_offset - 1) * ov + 1;                      // :40,43
    const unsigned long long m_brk = __ballot(brk), m_bad = __ballot(in && sp < 1);
    const int cnt = m_brk ? __ffsll((long long)m_brk) - 1 : nc;
    if (m_bad & bits_below(cnt)) { LANE0(set_status(st, 0, GSMCAL_E_INDEX)); return; }   // (n_win stays 0)
    if (lane < cnt) { st->win_start[lane] = sp - 1; st->fine_ws[lane] = sp - 1; }
    LANE0(st->n_win = cnt; st->n_fine_ws = lvl == 0 ? cnt : 0);
}

// the reference's spacing test of consecutive positions (FCCH_fine_correction.m:85-93, SCH_corr_rate_correction.m:96-104):
// bit i of a / b: pos[i+1]-pos[i] within the tolerance of 10 / 11 frames
__device__ __forceinline__ void spacing_masks(const double* pos, int n, double d_ov, double d1_ov, double max_ppm, int lane,
                                              unsigned* a_mask, unsigned* b_mask) {
    const double max_th = floor(d_ov * max_ppm * 1e-6), max_th1 = floor(d1_ov * max_ppm * 1e-6);
    const bool act = lane < n - 1;
    const double diff = act ? pos[lane + 1] - pos[lane] : 0.0;
    *a_mask = (unsigned)__ballot(act && fabs(diff - d_ov) < max_th);
    *b_mask = (unsigned)__ballot(act && fabs(diff - d1_ov) < max_th1);
}
// the regenerated grid (FCCH_fine_correction.m:127-133, SCH_corr_rate_correction.m:135): acc = 1, then + 10 or 11 frames
// per gap (11 wins where both tests passed); every term is an integer below 2^53, so the sum is exact in any order
__device__ __forceinline__ double grid_pos(unsigned a_mask, unsigned b_mask, double d_ov, double d1_ov, double first, int lane) {
    const unsigned below = lane >= 32 ? ~0u : ((1u << lane) - 1u);
    const double acc = 1.0 + ((double)__popc(a_mask & ~b_mask & below) * d_ov + (double)__popc(b_mask & below) * d1_ov);
    return acc + first - 1.0;
}

// FCCH_fine_correction.m:52-137 -- positions, sampling error, new grid, burst windows at level lvl+1
__device__ void d_fine_decide(StreamState* st, int ov, int lvl, const DevParams& P, int lane) {
    if (st->status < 0 || st->stage_status[0] != 0) { LANE0(st->n_win = 0); return; }
    const int last_idx = st->n_win;        // fine_first[0..last_idx) was filled by the lanes of step_body
    LANE0(st->n_fine = last_idx; st->n_win = 0);
    const int fft_len = 148 * ov;
    if (last_idx < P.min_hits) {                                               // :69 not taken
        if (lane < last_idx) st->fcch_pos[lane] = st->fine_first[lane];
        LANE0(st->fcch_is_sentinel = 0; st->n_fcch = last_idx; set_status(st, 0, GSMCAL_S_FINE_FEW));
        return;
    }
    const double d_ov = 10.0 * 1250.0 * (double)ov, d1_ov = 11.0 * 1250.0 * (double)ov;   // :80-81
    unsigned a_mask, b_mask;
    spacing_masks(st->fine_first, last_idx, d_ov, d1_ov, P.fine_max_ppm, lane, &a_mask, &b_mask);   // :83-93
    const int na = __popc(a_mask), nb = __popc(b_mask);
    if (na + nb != last_idx - 1) {                                             // :95-102
        LANE0(st->r1_kind = 1; set_status(st, 0, GSMCAL_S_FINE_SPACING));      // r = s was already assigned (:72)
        return;
    }
    const double expected = (double)na * d_ov + (double)nb * d1_ov;            // :111
    const double actual = st->fine_first[last_idx - 1] - st->fine_first[0];
    const double e = (actual - expected) / expected;                           // :113
    const long len_r = level_len(st, lvl);
    const long max_len = e >= 0.0 ? (long)floor((double)len_r / (1.0 + e)) : len_r;   // :118-122
    LANE0(st->sampling_ppm1 = e * 1e6;
          st->op[lvl + 1].type = OP_LERP; st->op[lvl + 1].param = 1.0 + e; st->op[lvl + 1].n = max_len;
          st->r1_kind = 2);
    // :127-133 regenerated grid
    const double first = round((st->fine_first[0] - 1.0) / (1.0 + e)) + 1.0;
    const double pos = grid_pos(a_mask, b_mask, d_ov, d1_ov, first, lane);
    if (lane < last_idx) st->fcch_pos[lane] = pos;
    int n = last_idx;
    if (__shfl(pos, n - 1, 64) + (double)fft_len - 1.0 > (double)max_len) --n;   // :135
    LANE0(st->n_fcch = n; st->fcch_is_sentinel = 0);
    if (n >= P.min_hits) {                                                     // :142
        const long sp = (long)pos;
        if (__ballot(lane < n && (sp < 1 || sp + fft_len - 1 > max_len))) { LANE0(set_status(st, 0, GSMCAL_E_INDEX)); return; }
        if (lane < n) st->win_start[lane] = sp - 1;
        LANE0(st->n_win = n);
    } else {
        LANE0(set_status(st, 0, GSMCAL_S_FINE_FEW_BURSTS));
    }
}

// mean(fo), carrier ppm and the derotation op; shared by fine (:158-165) and post-SCH (:75-83).  (The mean keeps the
// reference's order of additions: every lane adds the same values.)
__device__ __forceinline__ void carrier_from_bursts(StreamState* st, int nb, int ov, double carrier_freq,
                                                    int op_level, long n, double* ppm, int lane) {
    const double sampling_rate = GSM_SYMBOL_RATE * (double)ov;
    const double target_freq = GSM_SYMBOL_RATE / 4.0;
    double fo = 0.0;
    for (int i = 0; i < nb; ++i) fo += st->fo_burst[i];
    fo = fo / (double)nb;
    const double comp_freq = target_freq - fo;
    const double cpr = comp_freq * 2.0 * PI_D / sampling_rate;
    LANE0(*ppm = 1e6 * (fo - target_freq) / carrier_freq;
          st->op[op_level].type = OP_MIX; st->op[op_level].param = cpr; st->op[op_level].n = n);
}

// FCCH_fine_correction.m:158-165,192-196
__device__ void d_carrier_decide(StreamState* st, int s, int ov, const double* carrier_freq, int lvl, const DevParams& P, int lane) {
    const int nb = st->n_win;
    LANE0(st->n_win = 0);
    if (nb == 0) return;
    carrier_from_bursts(st, nb, ov, carrier_freq[s], lvl + 2, st->op[lvl + 1].n, &st->carrier_ppm1, lane);
    LANE0(st->r1_kind = 3);
    if (__ballot(lane < nb && st->snr_burst[lane] < P.fine_gate_snr))          // :192
        LANE0(st->n_fcch = 0; st->fcch_is_sentinel = 1; set_status(st, 0, GSMCAL_S_FINE_LOW_SNR));
}

// SCH_corr_rate_correction.m:8-48 -- correlation windows at level lvl
__device__ void d_sch_setup(StreamState* st, int ov, int len_ts, int lvl, const DevParams& P, int lane) {
    LANE0(st->sch_edge = 0;
          st->n_win = 0; st->n_sch_first = 0; st->n_sch = 0; st->n_rows = 0; st->n_sent_rows = 1;   // :9 pos_info = [-1, -1]
          st->sampling_ppm2 = INFINITY; st->r2_kind = 0);
    if (lane >= 1 && lane < NLEVELS && lane > lvl) { st->op[lane].type = OP_NONE; st->op[lane].n = 0; }
    if (st->status < 0) return;
    if (st->fcch_is_sentinel || st->n_fcch < P.min_hits) { LANE0(set_status(st, 1, GSMCAL_S_FEW_HITS)); return; }  // :11
    const long len_s_ov = level_len(st, lvl);
    const long fix_off = (long)((1250 + 42) * ov);       // :26-27
    const long max_offset = 8 * ov;                      // :36
    const int nf = st->n_fcch;
    const bool in = lane < nf;
    const long training_sp = in ? (long)st->fcch_pos[lane] + fix_off : 0;
    const bool brk = in && training_sp + max_offset > len_s_ov - len_ts + 1;   // :40
    const long sp = training_sp - max_offset;
    const unsigned long long m_brk = __ballot(brk), m_bad = __ballot(in && sp < 1);
    const int cnt = m_brk ? __ffsll((long long)m_brk) - 1 : nf;
    if (m_bad & bits_below(cnt)) { LANE0(set_status(st, 1, GSMCAL_E_INDEX)); return; }
    if (lane < cnt) st->win_start[lane] = sp - 1;
    LANE0(st->n_win = cnt);
}

// SCH_corr_rate_correction.m:59-181
__device__ void d_sch_decide(StreamState* st, int ov, int lvl, const DevParams& P, int lane) {
    const int num_sch = st->n_win;
    LANE0(st->n_win = 0);
    if (st->status < 0 || st->stage_status[1] != 0) return;
    LANE0(st->n_sch_first = num_sch);
    if (st->sch_edge) { LANE0(set_status(st, 1, GSMCAL_S_SCH_EDGE)); return; }   // :59-63 pos_info = [-1, -1]
    LANE0(st->n_sent_rows = 3 * st->n_fcch);                                   // :32 pos_info = -ones(3*num_fcch_hit, 2) from here on
    if (num_sch < P.min_hits) { LANE0(set_status(st, 1, GSMCAL_S_SCH_FEW)); return; }   // :84
    const double frame_ov = 1250.0 * (double)ov, slot_ov = 156.25 * (double)ov;
    const double d_ov = 10.0 * frame_ov, d1_ov = 11.0 * frame_ov;
    unsigned a_mask, b_mask;
    spacing_masks(st->sch_first, num_sch, d_ov, d1_ov, P.sch_max_ppm, lane, &a_mask, &b_mask);   // :94-104
    const int na = __popc(a_mask), nb = __popc(b_mask);
    LANE0(st->r2_kind = 1);                                                    // :87 r = s
    if (na + nb != num_sch - 1) { LANE0(set_status(st, 1, GSMCAL_S_SCH_SPACING)); return; }   // :106-112
    const double expected = (double)na * d_ov + (double)nb * d1_ov;
    const double actual = st->sch_first[num_sch - 1] - st->sch_first[0];
    const double e = (actual - expected) / expected;
    const long len_in = level_len(st, lvl);
    long len_r = len_in;
    if (e != 0.0) len_r = e > 0.0 ? (long)floor((double)len_in / (1.0 + e)) : len_in;   // :120
    LANE0(st->sampling_ppm2 = e * 1e6;
          st->op[lvl + 1].type = e != 0.0 ? OP_LERP : OP_COPY; st->op[lvl + 1].param = e != 0.0 ? 1.0 + e : 1.0;
          st->op[lvl + 1].n = len_r;
          st->r2_kind = 2);
    const double first = round((st->sch_first[0] - 1.0) / (1.0 + e)) + 1.0;   // :135
    const double pos = grid_pos(a_mask, b_mask, d_ov, d1_ov, first, lane);
    if (lane < num_sch) st->sch_pos[lane] = pos;
    LANE0(st->n_sch = num_sch);
    // :138-141 BCCH_flag (1-based indices 1..num_sch+1): an 11-frame gap b_idx = i+1 flags b_idx+1 and, from the fifth on, b_idx-4
    const unsigned bcch = (b_mask << 2) | ((b_mask & ~0xFu) >> 3);
    // :143-181 rows of SCH i: its FCCH (type 0), itself (1) if the slot fits, then 4 BCCH slots (2) when flagged, each only
    // if it fits; the first slot that does not fit ends the whole table
    const double fix_off = (double)((1250 + 42) * ov), pre_ts = (double)(42 * ov);
    const bool in = lane < num_sch;
    const double sch_sp = pos - pre_ts;
    int count = 0;
    bool complete = true;
    if (in) {
        count = 1;
        if (sch_sp + slot_ov - 1.0 <= (double)len_r) {
            count = 2;
            if (bcch & (1u << (lane + 1))) {
                for (int idx = 1; idx <= 4; ++idx) {
                    const double sp = sch_sp + (double)idx * frame_ov;
                    if (complete && sp + slot_ov - 1.0 <= (double)len_r) ++count; else complete = false;
                }
            }
        } else complete = false;
    }
    const unsigned long long m_inc = __ballot(in && !complete);
    const int last = m_inc ? __ffsll((long long)m_inc) - 1 : num_sch - 1;     // the last SCH that contributes rows
    if (lane > last) count = 0;
    int incl = count;
    for (int off = 1; off < 32; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    int rows = __shfl(incl, last, 64);
    if (rows > MAXROWS) {
        LANE0(set_status(st, 1, GSMCAL_E_CAPACITY));
        rows = 0;
    } else if (count > 0) {
        double* pi0 = st->pos_info + (incl - count);
        double* pi1 = pi0 + MAXROWS;
        pi0[0] = pos - fix_off; pi1[0] = 0.0;
        if (count >= 2) { pi0[1] = sch_sp; pi1[1] = 1.0; }
        for (int idx = 1; idx <= 4; ++idx)
            if (count >= 2 + idx) { pi0[1 + idx] = sch_sp + (double)idx * frame_ov; pi1[1 + idx] = 2.0; }
    }
    LANE0(st->n_rows = rows);
}

// carrier_correct_post_SCH.m:8-62 -- FCCH-row windows at level lvl
__device__ void d_post_setup(StreamState* st, int ov, int lvl, const DevParams& P, int lane) {
    LANE0(st->n_win = 0; st->carrier_ppm2 = INFINITY; st->r3_kind = 0);
    if (lane >= 1 && lane < NLEVELS && lane > lvl) { st->op[lane].type = OP_NONE; st->op[lane].n = 0; }
    if (st->status < 0) return;
    const int n_rows = st->n_rows;
    if (n_rows == 0) { LANE0(set_status(st, 2, GSMCAL_S_POST_NO_POS)); return; }  // :10
    const double* pi0 = st->pos_info;
    const double* pi1 = st->pos_info + MAXROWS;
    int nb = 0;
    for (int r0 = 0; r0 < n_rows; r0 += 64) nb += __popcll(__ballot(r0 + lane < n_rows && pi1[r0 + lane] == 2.0));
    if (nb < P.post_min_bcch) { LANE0(set_status(st, 2, GSMCAL_S_POST_FEW_BCCH)); return; }   // :15-19
    const int fft_len = 148 * ov;
    const long len = level_len(st, lvl);
    int cnt = 0;
    bool bad = false;
    for (int r0 = 0; r0 < n_rows; r0 += 64) {
        const int r = r0 + lane;
        const bool is0 = r < n_rows && pi1[r] == 0.0;
        const unsigned long long m = __ballot(is0);
        const int idx = cnt + __popcll(m & bits_below(lane));
        const long sp = is0 ? (long)pi0[r] : 1;
        if (is0) {
            if (sp < 1 || sp + fft_len - 1 > len || idx >= MAXH) bad = true;
            else st->win_start[idx] = sp - 1;
        }
        cnt += __popcll(m);
    }
    if (__ballot(bad)) { LANE0(set_status(st, 2, GSMCAL_E_INDEX)); return; }
    LANE0(st->n_win = cnt);
}

// carrier_correct_post_SCH.m:75-83
__device__ void d_post_decide(StreamState* st, int s, int ov, const double* carrier_freq, int lvl, int lane) {
    const int nb = st->n_win;
    LANE0(st->n_win = 0);
    if (nb == 0) return;
    carrier_from_bursts(st, nb, ov, carrier_freq[s], lvl + 1, level_len(st, lvl), &st->carrier_ppm2, lane);
    LANE0(st->r3_kind = 3);
}

// total_ppm_calculation.m:5-21
__host__ __device__ inline double total_ppm(double a, double b) {
    if (a == INFINITY && b == INFINITY) return INFINITY;
    return ((1.0 + a * 1e-6) * (1.0 + b * 1e-6) - 1.0) * 1e6;
}

// gsm_sync_demod.m:123-124 + table row
__device__ void d_totals(const StreamState* st, int s, double* table, double* pos_info_out, long* r_len_out, int lane) {
    if (lane == 0) {
        double* row = table + (size_t)s * GSMCAL_TABLE_COLS;
        row[GSMCAL_T_SAMPLING_PPM_FCCH] = st->sampling_ppm1;
        row[GSMCAL_T_SAMPLING_PPM_SCH] = st->sampling_ppm2;
        row[GSMCAL_T_CARRIER_PPM_FCCH] = st->carrier_ppm1;
        row[GSMCAL_T_CARRIER_PPM_POST] = st->carrier_ppm2;
        row[GSMCAL_T_TOTAL_SAMPLING_PPM] = total_ppm(st->sampling_ppm1, st->sampling_ppm2);
        row[GSMCAL_T_TOTAL_CARRIER_PPM] = total_ppm(st->carrier_ppm1, st->carrier_ppm2);
        row[GSMCAL_T_N_FCCH] = st->fcch_is_sentinel ? 1.0 : (double)st->n_fcch;
        row[GSMCAL_T_N_POS_ROWS] = st->n_rows == 0 ? (double)(st->n_sent_rows > 0 ? st->n_sent_rows : 1) : (double)st->n_rows;
        row[GSMCAL_T_FIRST_FCCH_POS] = st->n_rows == 0 ? -1.0 : st->pos_info[0];
        row[GSMCAL_T_STATUS] = (double)st->status;
        if (r_len_out) r_len_out[s] = st->r3_kind == 3 ? st->op[4].n : -1;
    }
    if (pos_info_out) {
        double* o = pos_info_out + (size_t)s * 2 * MAXROWS;
        const int n_rows = st->n_rows;
        for (int i = lane; i < 2 * MAXROWS; i += 64) o[i] = (i < MAXROWS ? i : i - MAXROWS) < n_rows ? st->pos_info[i] : -1.0;
    }
}

// multi_rtl_sdr_gsm_FCCH_scanner.m:168-185 acceptance -> (snr, num_hit) per capture
__device__ void d_scan_accept(const StreamState* st, int s, double* snr_numhit, double* positions,
                              double* pos_snr, int* counts, const DevParams& P, int lane) {
    const int n = st->n_coarse;
    double snr = 0.0, num_hit = 0.0;
    if (n >= P.scan_min_hits) {
        bool fail = false;
        if (lane < n - 1) {
            const double d = st->coarse_pos[lane + 1] - st->coarse_pos[lane];
            fail = fabs(d - P.scan_spacing) > P.scan_tol && fabs(d - P.scan_spacing_idle) > P.scan_tol;
        }
        if (!__ballot(fail)) {
            double sum = 0.0;
            for (int i = 0; i < n; ++i) sum += st->coarse_snr[i];
            snr = sum / (double)n;
            num_hit = (double)n;
        }
    }
    if (lane == 0) {
        snr_numhit[2 * s] = snr;
        snr_numhit[2 * s + 1] = num_hit;
        if (counts) counts[s] = n;
    }
    if (positions && lane < MAXH) {
        positions[(size_t)s * MAXH + lane] = lane < n ? st->coarse_pos[lane] : (n == 0 && lane == 0 ? -1.0 : 0.0);
        if (pos_snr) pos_snr[(size_t)s * MAXH + lane] = lane < n ? st->coarse_snr[lane] : (n == 0 && lane == 0 ? -1.0 : 0.0);
    }
}

// ------------------------------------------------------------------------------------------------
// Launchers for the decision steps: grid S, block 64.  The stream's state is copied into LDS with
// coalesced 16-byte accesses, lane 0 runs the (serial, branchy) reference logic on the LDS copy, and
// the state is copied back -- the logic is a chain of dependent loads/stores that would otherwise
// each pay a global-memory round trip.
// ------------------------------------------------------------------------------------------------
struct StateLds {
    __device__ static void load(StreamState* sh, const StreamState* g, int lane) {
        const uint4* src = (const uint4*)g;
        uint4* dst = (uint4*)sh;
        for (int i = lane; i < (int)(sizeof(StreamState) / 16); i += 64) dst[i] = src[i];
    }
    __device__ static void store(StreamState* g, const StreamState* sh, int lane) {
        const uint4* src = (const uint4*)sh;
        uint4* dst = (uint4*)g;
        for (int i = lane; i < (int)(sizeof(StreamState) / 16); i += 64) dst[i] = src[i];
    }
};

struct StepArgs {
    int ov, lvl, len_ts, H, NB;
    const PeakOut* peaks;
    const double* carrier_freq;
    double* table; double* pos_info_out; long* r_len_out;
    double* snr_numhit; double* positions; double* pos_snr; int* counts;
    DevParams P;
};

enum { STEP_FINE_SETUP = 1, STEP_FINE_DECIDE = 2, STEP_CARRIER_DECIDE = 4, STEP_SCH_SETUP = 8, STEP_SCH_DECIDE = 16,
       STEP_POST_SETUP = 32, STEP_POST_DECIDE = 64, STEP_TOTALS = 128, STEP_SCAN_ACCEPT = 256 };

// `steps` is a bit set executed in the order of the enum; lvl_a / lvl_b: input level of the first /
// second step of a merged pair (e.g. carrier_decide works on the fine stage's level, sch_setup on the
// SCH stage's input level).  Called by every thread of a workgroup (>= 64 threads); `sh` is an LDS copy of the state.
template <bool COHERENT, bool WT_STORE = false>
__device__ __forceinline__ void step_body(StreamState* __restrict__ sts, const StepArgs& a, int steps, int lvl_a,
                                          int lvl_b, int s, StreamState* sh, int kid = -1) {
    const int lane = threadIdx.x;
#define TAIL_STAMP(i) do { if (kid >= 0) DEV_STAMP(kid, blockIdx.y * gridDim.x + blockIdx.x, i); } while (0)
    TAIL_STAMP(8);
    if (COHERENT) {       // inside the kernel that produced part of the state: read it past the caches
        const unsigned long long* src = (const unsigned long long*)(sts + s);
        unsigned long long* dst = (unsigned long long*)sh;
        for (int i = lane; i < (int)(sizeof(StreamState) / 8); i += blockDim.x) dst[i] = coherent_load(src + i);
    } else if (lane < 64) {
        StateLds::load(sh, sts + s, lane);
    }
    __syncthreads();
    TAIL_STAMP(9);
    if ((steps & STEP_FINE_DECIDE) && lane < sh->n_win && lane < MAXH) {
        // merge the NB partial peaks of window `lane` (the loads of all windows are in flight together)
        const PeakOut* pw = a.peaks + ((size_t)s * a.H + lane) * a.NB;
        const PeakOut pk = COHERENT ? merge_peaks_coherent(pw, a.NB) : merge_peaks(pw, a.NB);
        sh->fine_first[lane] = (double)(sh->win_start[lane] + 1 + pk.tie);          // sp + max_idx - 1
        sh->prior_bin[lane] = pk.k;
    }
    if (steps & STEP_FINE_DECIDE) __syncthreads();
    TAIL_STAMP(10);
    if (lane < 64) {      // one wave; wsync() hands each step's LDS writes to the next step's lanes
        if (steps & STEP_FINE_SETUP) { d_fine_setup(sh, a.ov, lvl_a, a.P.min_hits, lane); wsync(); }
        if (steps & STEP_FINE_DECIDE) { d_fine_decide(sh, a.ov, lvl_a, a.P, lane); wsync(); }
        if (steps & STEP_CARRIER_DECIDE) { d_carrier_decide(sh, s, a.ov, a.carrier_freq, lvl_a, a.P, lane); wsync(); }
        if (steps & STEP_SCH_SETUP) { d_sch_setup(sh, a.ov, a.len_ts, lvl_b, a.P, lane); wsync(); }
        if (steps & STEP_SCH_DECIDE) { d_sch_decide(sh, a.ov, lvl_a, a.P, lane); wsync(); }
        if (steps & STEP_POST_SETUP) { d_post_setup(sh, a.ov, lvl_b, a.P, lane); wsync(); }
        if (steps & STEP_POST_DECIDE) { d_post_decide(sh, s, a.ov, a.carrier_freq, lvl_a, lane); wsync(); }
        if (steps & STEP_TOTALS) d_totals(sh, s, a.table, a.pos_info_out, a.r_len_out, lane);
        if (steps & STEP_SCAN_ACCEPT) d_scan_accept(sh, s, a.snr_numhit, a.positions, a.pos_snr, a.counts, a.P, lane);
    }
    __syncthreads();
    TAIL_STAMP(11);
    if (WT_STORE) {
        // other workgroups of THIS launch read the state next (k_post_chain): write-through stores by the whole workgroup
        // (one or two words per thread), every storing wave drains its own before the caller raises the stream's flag
        const unsigned long long* src = (const unsigned long long*)sh;
        unsigned long long* dst = (unsigned long long*)(sts + s);
        for (int i = lane; i < (int)(sizeof(StreamState) / 8); i += blockDim.x)
            __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (!(steps == STEP_TOTALS || steps == STEP_SCAN_ACCEPT) && lane < 64) StateLds::store(sts + s, sh, lane);
    TAIL_STAMP(12);
#undef TAIL_STAMP
}

template <int STEPS>
__global__ void __launch_bounds__(64) k_step(StreamState* __restrict__ sts, StepArgs a, int lvl_a, int lvl_b) {
    __shared__ StreamState sh;
    step_body<false>(sts, a, STEPS, lvl_a, lvl_b, blockIdx.x, &sh);
}

// ------------------------------------------------------------------------------------------------
// Decision steps riding on the per-window kernels.  The per-window kernels run H workgroups per stream; the LAST
// of them to finish (a per-stream counter, re-armed by that workgroup) runs the stream's decision steps right
// there, on the LDS its window no longer needs -- no separate one-block-per-stream launch between the stages.
// Every workgroup of the grid must call stream_tail (also those without a window), with all its threads.
// ------------------------------------------------------------------------------------------------
struct TailArgs {
    unsigned* ctr;            // one counter per stream, zero between kernels; nullptr: no tail (decisions launched apart)
    int steps, lvl_a, lvl_b;
    StepArgs sa;
};

__device__ __forceinline__ void stream_tail(StreamState* __restrict__ sts, const TailArgs& t, unsigned char* smem, int kid = -1) {
    if (!t.ctr) return;
    __shared__ int sh_last;
    // this workgroup's hand-over values were stored write-through (coherent_store); wait until they are acknowledged
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = atomicAdd(&t.ctr[blockIdx.y], 1u);   // relaxed, agent scope
        sh_last = old == gridDim.x - 1;
        if (sh_last) atomicExch(&t.ctr[blockIdx.y], 0u);          // re-arm for the next kernel
    }
    __syncthreads();
    if (!sh_last) return;                                   // block-uniform
    step_body<true>(sts, t.sa, t.steps, t.lvl_a, t.lvl_b, blockIdx.y, (StreamState*)smem, kid);
}

// OV, LT, NTAPS > 0 (here and in the kernels below): the reference geometry as compile-time constants, see k_post_chain_r
template <int GATE, int OV, int NTAPS>
__global__ void __launch_bounds__(BT_THREADS) __attribute__((amdgpu_waves_per_eu(6, 8)))
k_burst_tone(StreamState* __restrict__ sts, GatherArgs a_in, int nfft_rt, const cplx* __restrict__ tw_g, int ov_rt, int prior_mode,
             TailArgs tail) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GatherArgs a = a_in;
    const int nfft = OV > 0 ? 148 * OV : nfft_rt, ov = OV > 0 ? OV : ov_rt;
    if (OV > 0) { a.len = 148 * OV; a.ntaps = NTAPS; a.src_kind = SRC_RAW; a.tiles = 0; a.level = GATE ? 1 : 3; }   // (the chain on raw bytes from level 0: bursts at levels 1 and 3)
    burst_tone_body<GATE, (OV > 0 ? 2 : 1)>(sts, a, nfft, tw_g, ov, prior_mode, smem);
    DEV_STAMP(GATE ? KID_BT1 : KID_BT0, blockIdx.y * gridDim.x + blockIdx.x, 5);
    stream_tail(sts, tail, smem, GATE ? KID_BT1 : KID_BT0);
    DEV_STAMP(GATE ? KID_BT1 : KID_BT0, blockIdx.y * gridDim.x + blockIdx.x, 6);
}

template <int OV, int LT, int NTAPS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 8))) k_window_sch(StreamState* __restrict__ sts, GatherArgs a_in,
                                                    const cplx* __restrict__ ts, int len_ts_rt, int nshift_rt, TailArgs tail) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GatherArgs a = a_in;
    const int len_ts = OV > 0 ? LT : len_ts_rt, nshift = OV > 0 ? 11 * OV + 1 : nshift_rt;
    if (OV > 0) { a.len = 11 * OV + LT; a.ntaps = NTAPS; a.src_kind = SRC_RAW; a.tiles = 0; a.level = 2; }
    window_sch_body<(OV > 0 ? 2 : 1), (OV == 8 && LT == 512)>(sts, a, ts, len_ts, nshift, smem);
    DEV_STAMP(KID_SCH, blockIdx.y * gridDim.x + blockIdx.x, 3);
    stream_tail(sts, tail, smem, KID_SCH);
    DEV_STAMP(KID_SCH, blockIdx.y * gridDim.x + blockIdx.x, 4);
}

__global__ void __launch_bounds__(FV_THREADS) k_fine_verify(const StreamState* __restrict__ sts,
                                                     const cplx* __restrict__ win, long win_stream_stride,
                                                     long win_stride, int nshift, int nfft,
                                                     const cplx* __restrict__ tw_g, const ChunkRec* __restrict__ rec,
                                                     PeakOut* __restrict__ out, int H,
                                                     const FineCert* __restrict__ cert, int* __restrict__ n_open,
                                                     StreamState* __restrict__ sts_rw, TailArgs tail) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DEV_STAMP(KID_VERIFY, blockIdx.y * gridDim.x + blockIdx.x, 0);
    fine_verify_body<FV_THREADS>(sts, win, win_stream_stride, win_stride, nshift, nfft, tw_g, rec, out, H, cert, n_open, smem);
    DEV_STAMP(KID_VERIFY, blockIdx.y * gridDim.x + blockIdx.x, 1);
    stream_tail(sts_rw, tail, smem, KID_VERIFY);
    DEV_STAMP(KID_VERIFY, blockIdx.y * gridDim.x + blockIdx.x, 2);
}

#define PC_THREADS 512
struct PostChainArgs {
    GatherArgs ga1, ga_sch, ga0;       // burst windows at level lvl+1, SCH search windows, post-SCH burst windows
    StepArgs sa;                       // one set of step arguments serves every decision step (NB = 1)
    unsigned* done;                    // pinned host words of the context, one per stream: the launch count of the stream's last finished
                                       // fused tail (gsmcal_ctx::fused_done; nullptr: not reported)
    unsigned* timed_out;               // pinned host word: set when a workgroup gave up waiting for a peer (the host then re-runs the tail as four launches)
    unsigned long long poll_ticks;     // how long a workgroup waits for its stream's peers at an exchange, in 100 MHz wall-clock ticks
    int test_stall;                    // test hook: > 0: workgroup (1, 0) never publishes its stage-(test_stall - 1) granules
    int lvl_fine, lvl_sch, lvl_post;   // input levels of the three reference functions
    int nfft, ov, len_ts, sch_nshift, fine_nshift, H;
    const cplx* tw_g; const cplx* ts;
    const cplx* win; long win_stream_stride, win_stride;
    const ChunkRec* rec; const FineCert* cert; PeakOut* peaks; int* n_open;
    int with_totals, pad;
};

// ------------------------------------------------------------------------------------------------
// k_post_chain_r: everything behind the fine search's chunk sweep in ONE launch -- the exact last word of the fine search
// (fine_verify_body) and the three per-burst stages (burst_tone_body<1> -> window_sch_body -> burst_tone_body<0>) with the
// reference's decision steps between them, REPLICATED.  grid (H, S), block 512: workgroup (w, s) handles window w of stream
// s in every stage and keeps its own copy of the stream's state in LDS for the whole launch.  A stage ends with each
// workgroup publishing its two result words (tone frequency + gate SNR, SCH position + edge flag, peak power + shift/bin)
// as 8-byte write-through granules in the stream's exchange block; it then reads the granules of all H workgroups of its
// stream (one lane per granule, polling past the L1 until none is EMPTY) and runs the reference's decision step ITSELF on
// its own LDS copy -- the same instructions on the same inputs in every workgroup, so all copies stay identical.  No
// counter, no last arriver, no state written and read back through memory, no flag, no fence (every word that crosses
// workgroups travels by sc1 write-through stores and L1-bypassing loads: MI355X_MICROARCH.md, inter-workgroup visibility):
// after the slowest workgroup of the stream has published, its peers continue as soon as they see its granule (~1 us) and
// have run the step on LDS (~1.5 us).  Workgroup 0 of the stream writes the table row and the final state.  The exchange
// block is double-buffered by the parity of a per-stream launch counter (each workgroup clears its own granules of the
// other parity on entry), so the launch can be replayed from a hipGraph unchanged.  EMPTY = all ones (a NaN no computation
// produces: published NaNs are canonicalised).
//
// Forward progress.  A workgroup waits only for the H workgroups of its own stream, which are neighbours in dispatch order
// (blockIdx.y = stream): the oldest unfinished stream of the launch is first in line for every slot that frees, so the
// launch advances whenever the device gives it slots at all -- other tenants (another context's kernels, RCCL, a kernel
// hogging the CUs) only delay it, like any kernel.  What could stop it is a SECOND spinning launch holding the slots this
// one's next workgroups need while waiting for slots this one holds; the host therefore never has two fused tails of this
// process in flight on one device (host_plan.h: fused_gate -- the later caller takes the four-launch tail).  Another
// PROCESS's fused tail is outside that gate: there the poll limit of pcr_exchange (PostChainArgs::poll_ticks, 3 s of the
// 100 MHz counter) ends the wait, the stream's rows report GSMCAL_E_HIP, a pinned flag tells the host, and gsmcal_sync /
// the host-buffer entry points run the call again as four launches (round 6; abi_calls.h: fused_recover).
// (A take-over scheme -- a workgroup whose poll runs out computes the missing peer's window itself -- was built in round
// 5 and dropped: every form of it (a loop around the stage bodies, a restart loop around the kernel, an out-of-line cold
// helper) cost 0.9-1.5 KB of scratch per lane in a kernel that has none, and a scratch frame that size delays the launch
// of every wave: NOTES_r05.md.)
// ------------------------------------------------------------------------------------------------
#define PCR_EMPTY 0xFFFFFFFFFFFFFFFFull
#define PCR_STATE_BYTES ((sizeof(StreamState) + 15) & ~(size_t)15)

__device__ __forceinline__ unsigned long long pcr_word(double v) {
    return v != v ? 0x7FF8000000000000ull : (unsigned long long)__double_as_longlong(v);
}

// publish r0, r1 as this workgroup's granules of `stage`, then collect the stream's 2*H granules into all[] (LDS)
__device__ __forceinline__ void pcr_exchange(unsigned long long* __restrict__ slots, int H, int w, unsigned long long r0,
                                             unsigned long long r1, unsigned long long* all, StreamState* sh, bool collect,
                                             unsigned long long poll_ticks, unsigned* __restrict__ timed_out, bool publish = true) {
    const int tid = threadIdx.x;
    if (tid < 2 && publish) __hip_atomic_store(slots + 2 * w + tid, tid ? r1 : r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!collect) return;
    if (tid < 64) {
        const bool mine = tid < 2 * H;
        unsigned long long v = PCR_EMPTY;
        unsigned spins = 0;
        unsigned long long t0 = 0;
        bool bad = false;
        while (true) {
            if (mine && v == PCR_EMPTY) v = __hip_atomic_load(slots + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!__ballot(mine && v == PCR_EMPTY)) break;
            __builtin_amdgcn_s_sleep(4);
            // a peer that does not come within poll_ticks (seconds; the constant-rate 100 MHz counter, looked at every 256 polls):
            // give up -- the rows of this stream report GSMCAL_E_HIP and the host, told through its pinned word, re-runs the tail
            // of the call as four launches (host_plan.h: fused_recover).  Never a hang, and no longer a lost call.
            if ((++spins & 255u) == 0u) {
                const unsigned long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                else if (now - t0 > poll_ticks) { bad = true; break; }
            }
        }
        if (mine) all[tid] = v;
        if (bad && tid == 0) {
            if (sh->status >= 0) sh->status = GSMCAL_E_HIP;
            sh->n_win = 0;                                           // (nothing downstream computes on the words that never came)
            if (timed_out) __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
}

// OV > 0: the oversampling ratio, the training-sequence length LT and the tap count NTAPS as compile-time constants (the
// reference geometry: 8, 512, 47 -- burst length 148*OV, 128*OV+1 fine shifts, 11*OV+1 SCH shifts): every `% nfft`, `/ N2`
// and loop bound of the stage bodies folds, 3.4 us of the launch at 64 streams.  OV = 0: all taken from the arguments.
template <int OV, int LT, int NTAPS>
__global__ void __launch_bounds__(PC_THREADS) __attribute__((amdgpu_waves_per_eu(6, 8)))
k_post_chain_r(StreamState* __restrict__ sts, PostChainArgs a_in, unsigned long long* __restrict__ xch, unsigned* __restrict__ epoch) {
    PostChainArgs a = a_in;
    // what the host guarantees for this kernel (run_fine: raw source, chain starting at level 0), as constants for the gathers
    a.ga1.src_kind = SRC_RAW; a.ga_sch.src_kind = SRC_RAW; a.ga0.src_kind = SRC_RAW;
    a.ga1.tiles = 0; a.ga_sch.tiles = 0; a.ga0.tiles = 0;
    a.ga1.level = 1; a.ga_sch.level = 2; a.ga0.level = 3;
    a.ga1.pad = 1; a.ga_sch.pad = 0; a.ga0.pad = 1;
    a.lvl_fine = 0; a.lvl_sch = 2; a.lvl_post = 3;
    if (OV > 0) {
        a.ov = OV; a.sa.ov = OV;
        a.nfft = 148 * OV; a.ga1.len = 148 * OV; a.ga0.len = 148 * OV;
        a.fine_nshift = 128 * OV + 1;
        a.sch_nshift = 11 * OV + 1;
        a.len_ts = LT; a.sa.len_ts = LT;
        a.ga_sch.len = 11 * OV + LT;
        a.ga1.ntaps = NTAPS; a.ga_sch.ntaps = NTAPS; a.ga0.ntaps = NTAPS;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    __shared__ unsigned long long all[64];
    __shared__ double res[2];
    __shared__ PeakOut pk;
    const int s = blockIdx.y, w = blockIdx.x, tid = threadIdx.x, H = gridDim.x, lane = tid;
    StreamState* sh = (StreamState*)smem_all;                       // this workgroup's copy of the stream's state
    unsigned char* smem = smem_all + PCR_STATE_BYTES;               // the stages' work area
    StreamState* shv = sh - s;                                      // so that the stage bodies' `sts + blockIdx.y` is the LDS copy
    DEV_STAMP(KID_GATHER, blockIdx.y * gridDim.x + blockIdx.x, 0);
    // exchange block of stream s: [parity][stage][MAXH][2] -- a layout that does not depend on this launch's H or S, so launches
    // of any geometry (eager, or replayed from graphs captured at different times) can follow each other on one block
    const unsigned ep = epoch[s], par = ep & 1u;
    constexpr int XST = 2 * MAXH;                                                // granules per stage
    unsigned long long* mine_x = xch + ((size_t)s * 2 + par) * 4 * XST;          // [stage][w][2]
    unsigned long long* other_x = xch + ((size_t)s * 2 + (par ^ 1u)) * 4 * XST;
    // the parity this launch does not use is left EMPTY for ALL MAXH windows (workgroup w clears w, w+H, w+2H, ...): the next
    // launch may run more windows per stream than this one
    if (tid < 8)
        for (int wq = w; wq < MAXH; wq += H)
            __hip_atomic_store(other_x + (size_t)(tid >> 1) * XST + 2 * wq + (tid & 1), PCR_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
        const uint4* src = (const uint4*)(sts + s);
        uint4* dst = (uint4*)sh;
        for (int i = tid; i < (int)(sizeof(StreamState) / 16); i += PC_THREADS) dst[i] = src[i];
    }
    if (tid == 0) { pk.p = -1.0; pk.tie = 0; pk.k = 0; res[0] = 0.0; res[1] = 0.0; }
    __syncthreads();
    // A peer of this stream that never publishes (pcr_exchange gives up: status < 0 and n_win = 0 in every workgroup that waited for
    // it) loses the stream for this launch: with no windows left every later stage body returns at once and every decision step is
    // a no-op (they all start from n_win / a non-negative status), the remaining exchanges are passed without waiting (everybody
    // publishes at once), and workgroup 0 still writes the row (status GSMCAL_E_HIP), the launch count and the host's word.  The
    // host then runs the call again as four launches (abi_calls.h: fused_recover).
    // ---- stage 0: the fine search's exact last word per window -> FINE_DECIDE (FCCH_fine_correction.m:52-137) ----
    fine_verify_body<PC_THREADS, 6>(shv, a.win, a.win_stream_stride, a.win_stride, a.fine_nshift, a.nfft, a.tw_g, a.rec, a.peaks, a.H,
                                 a.cert, a.n_open, smem, &pk);
    __syncthreads();
    DEV_STAMP(KID_GATHER, blockIdx.y * gridDim.x + blockIdx.x, 1);
    pcr_exchange(mine_x, H, w, pcr_word(pk.p), (unsigned long long)(unsigned)pk.tie | ((unsigned long long)(unsigned)pk.k << 32), all,
                 sh, true, a.poll_ticks, a.timed_out, !(a.test_stall == 1 && w == 1 && s == 0));
    if (tid < 64) {
        __builtin_amdgcn_s_setprio(3);                              // the stream waits for this wave: it issues ahead of everything else on its SIMD (62.2 -> 61.2 us)
        if (lane < sh->n_win && lane < MAXH && lane < H) {
            sh->fine_first[lane] = (double)(sh->win_start[lane] + 1 + (int)(all[2 * lane + 1] & 0xffffffffu));   // sp + max_idx - 1
            sh->prior_bin[lane] = (int)(all[2 * lane + 1] >> 32);
        }
        wsync();
        d_fine_decide(sh, a.sa.ov, a.lvl_fine, a.sa.P, lane); wsync();
        if (lane == 0) { res[0] = 0.0; res[1] = 0.0; }
    }
    __syncthreads();
    DEV_STAMP(KID_GATHER, blockIdx.y * gridDim.x + blockIdx.x, 3);
    // ---- stage 1: bursts of the resampled stream (:141-165, :185-196) -> CARRIER_DECIDE + SCH window setup ----
    PCR_PRIO(2, 1, 0)                                               // (the age order, made explicit; the gate inside turns it round)
    burst_tone_body<1>(shv, a.ga1, a.nfft, a.tw_g, a.ov, 1, smem, res, true);
    __syncthreads();
    DEV_STAMP(KID_GATHER, blockIdx.y * gridDim.x + blockIdx.x, 4);
    pcr_exchange(mine_x + XST, H, w, pcr_word(res[0]), pcr_word(res[1]), all, sh, true, a.poll_ticks, a.timed_out, !(a.test_stall == 2 && w == 1 && s == 0));
    if (tid < 64) {
        __builtin_amdgcn_s_setprio(3);
        if (lane < H && lane < MAXH) {
            sh->fo_burst[lane] = __longlong_as_double((long long)all[2 * lane]);
            sh->snr_burst[lane] = __longlong_as_double((long long)all[2 * lane + 1]);
        }
        wsync();
        d_carrier_decide(sh, s, a.sa.ov, a.sa.carrier_freq, a.lvl_fine, a.sa.P, lane); wsync();
        d_sch_setup(sh, a.sa.ov, a.sa.len_ts, a.lvl_sch, a.sa.P, lane); wsync();
        if (lane == 0) { res[0] = 0.0; res[1] = 0.0; }
    }
    __syncthreads();
    DEV_STAMP(KID_GATHER, blockIdx.y * gridDim.x + blockIdx.x, 6);
    // ---- stage 2: SCH_corr_rate_correction.m:45-55 -> SCH_DECIDE (:59-181) + post-SCH window setup ----
    const int n_sch_win = sh->n_win;
    PCR_PRIO(0, 1, 2)                                               // (the deciding wave returns to its round's priority)
    window_sch_body<(OV > 0 ? 2 : 1), (OV == 8 && LT == 512)>(shv, a.ga_sch, a.ts, a.len_ts, a.sch_nshift, smem, res);
    __syncthreads();
    DEV_STAMP(KID_GATHER, blockIdx.y * gridDim.x + blockIdx.x, 7);
    pcr_exchange(mine_x + 2 * XST, H, w, pcr_word(res[0]), pcr_word(res[1]), all, sh, true, a.poll_ticks, a.timed_out, !(a.test_stall == 3 && w == 1 && s == 0));
    if (tid < 64) {
        __builtin_amdgcn_s_setprio(3);
        const bool act = lane < H && lane < MAXH && lane < n_sch_win;
        if (act) sh->sch_first[lane] = __longlong_as_double((long long)all[2 * lane]);
        const unsigned long long edge = __ballot(act && __longlong_as_double((long long)all[2 * lane + 1]) != 0.0);
        if (lane == 0 && edge) sh->sch_edge = 1;
        wsync();
        d_sch_decide(sh, a.sa.ov, a.lvl_sch, a.sa.P, lane); wsync();
        d_post_setup(sh, a.sa.ov, a.lvl_post, a.sa.P, lane); wsync();
        if (lane == 0) { res[0] = 0.0; res[1] = 0.0; }
    }
    __syncthreads();
    DEV_STAMP(KID_GATHER, blockIdx.y * gridDim.x + blockIdx.x, 9);
    // ---- stage 3: carrier_correct_post_SCH.m:51-79 -> POST_DECIDE (:75-83) + the table row (gsm_sync_demod.m:123-124) ----
    PCR_PRIO(2, 1, 0)
    burst_tone_body<0>(shv, a.ga0, a.nfft, a.tw_g, a.ov, 0, smem, res, true);
    __syncthreads();
    DEV_STAMP(KID_GATHER, blockIdx.y * gridDim.x + blockIdx.x, 10);
    pcr_exchange(mine_x + 3 * XST, H, w, pcr_word(res[0]), 0ull, all, sh, w == 0, a.poll_ticks, a.timed_out, !(a.test_stall == 4 && w == 1 && s == 0));
    if (w != 0) return;                                             // workgroup 0 finishes the stream
    if (tid < 64) {
        __builtin_amdgcn_s_setprio(3);
        if (lane < H && lane < MAXH) sh->fo_burst[lane] = __longlong_as_double((long long)all[2 * lane]);
        wsync();
        d_post_decide(sh, s, a.sa.ov, a.sa.carrier_freq, a.lvl_post, lane); wsync();
        if (a.with_totals) d_totals(sh, s, a.sa.table, a.sa.pos_info_out, a.sa.r_len_out, lane);
        StateLds::store(sts + s, sh, lane);
        if (lane == 0) {
            epoch[s] = ep + 1u;                                     // the stream's launch count (its parity selects the exchange block)
            // the host's gate (one fused tail of the process in flight per device) reads this pinned word: a plain posted store
            // like the table row's (one ATOMIC on a shared host word per stream serialises on PCIe: +56 us at 64 streams)
            if (a.done) __hip_atomic_store(a.done + s, ep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
