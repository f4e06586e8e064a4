// kernels_frontend.h -- raw bytes -> DC removal -> FIR -> (decimate | windows | whole stream).
//
//   k_dc_sum        raw2iq.m:8      per-stream integer sums of the I and Q bytes (exact mean)
//   k_finish_mean                    sums -> StreamState.mean_{re,im}
//   k_raw2iq        raw2iq.m:6-8    materialise c - mean as complex double
//   k_fir_decim_raw gsm_sync_demod.m:107,110,117 / ..FCCH_scanner.m:132-135 fused: only the kept
//                   rows r(1:decim:end) of filter(coef,1,raw2iq(s)) are computed
//   k_front_fused   batch paths: ONE pass over the raw bytes -> exact byte sums + FIR of the RAW samples at the
//   k_front_fast<NT,SYM>  kept rows (any geometry / 47- or 31-tap decimate-by-64 with rows held in registers)
//   k_fir_arr       filter(coef,1,s) (+ r(1:decim:end,:)) on a complex array (chn_filter_8x_4x.m:13-15)
//   k_gather        evaluates a window (or every tile) of a stream at any level of the lazy chain
//                   (state.h) through LDS: raw -> FIR -> lerp -> mix -> lerp -> mix
#pragma once
#include "state.h"

typedef double2 cplx;

// sin/cos of a LARGE fp64 argument t (|t| up to ~1e6 rad: t = k*comp_phase_rotate, k up to 1e6) with the
// accuracy of a correctly reduced argument, several times cheaper than the library's general path:
// n = round(t/2pi); y = t - n*2pi with 2pi = C1 + C2 split so that the first fma is EXACT (t and n*C1 are
// both multiples of 2^-50 and the difference is < 8, so it fits a double), the second adds n*C2 <= 4e-11
// with an error <= ulp(y); then the library routine on |y| <= pi (its cheap small-argument path).
__device__ __forceinline__ void sincos_large(double t, double* sn, double* cs) {
    const double n = rint(t * 0.15915494309189535);            // 1/(2*pi)
    double y = fma(-n, 6.28318530717958623e+00, t);            // C1 = fl(2*pi)
    y = fma(-n, 2.44929359829470641e-16, y);                   // C2 = 2*pi - C1
    sincos(y, sn, cs);
}

// Sum of v over the 64 lanes of the wave, returned in every lane (wave-uniform).  Data-parallel-primitive moves instead of
// LDS-crossbar shuffles: xor 1, xor 2, half-row mirror, row mirror give every lane its 16-lane row total; row_bcast15 /
// row_bcast31 (GFX9) chain the four rows into lane 63.  ~20 VALU instructions, no LDS traffic, no s_waitcnt.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move_f64<0xB1, 0xF>(v);      // quad_perm [1,0,3,2]
    v += dpp_move_f64<0x4E, 0xF>(v);      // quad_perm [2,3,0,1]
    v += dpp_move_f64<0x141, 0xF>(v);     // row_half_mirror
    v += dpp_move_f64<0x140, 0xF>(v);     // row_mirror
    v += dpp_move_f64<0x142, 0xA>(v);     // row_bcast15 into rows 1 and 3 (other rows add 0)
    v += dpp_move_f64<0x143, 0xC>(v);     // row_bcast31 into rows 2 and 3
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// the same for 32-bit unsigned integers (exact: no overflow is the caller's business); the total is returned in every lane
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_move_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
    v += dpp_move_u32<0xB1, 0xF>(v);
    v += dpp_move_u32<0x4E, 0xF>(v);
    v += dpp_move_u32<0x141, 0xF>(v);
    v += dpp_move_u32<0x140, 0xF>(v);
    v += dpp_move_u32<0x142, 0xA>(v);
    v += dpp_move_u32<0x143, 0xC>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Inclusive prefix sum over the lanes of the wave (lane l gets v_0 + ... + v_l): row_shr 1, 2, 4, 8 inside the 16-lane
// rows, then the two row broadcasts.
__device__ __forceinline__ double wave_scan_incl(double v) {
    v += dpp_move_f64<0x111, 0xF>(v);
    v += dpp_move_f64<0x112, 0xF>(v);
    v += dpp_move_f64<0x114, 0xF>(v);
    v += dpp_move_f64<0x118, 0xF>(v);
    v += dpp_move_f64<0x142, 0xA>(v);
    v += dpp_move_f64<0x143, 0xC>(v);
    return v;
}

__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// ------------------------------------------------------------------------------------------------
// Stage raw samples [first, first+span) of one stream (I | Q<<8 per ushort) into LDS with 16-byte
// global loads.  Returns `first_al` <= first: sample g lives at r_s[g - first_al].  Samples outside
// [0, n) read as 0 (filter()'s zero initial state is applied by the callers through g < 0 tests).
// r_s must hold span + 16 ushorts and be 16-byte aligned.
// ------------------------------------------------------------------------------------------------
// PAD: insert 8 ushorts (16 B) after every 64 samples so that lanes whose samples are 64 apart (the
// decimate-by-64 FIR) do not all hit the same LDS bank; use lds_pad() to index.
__device__ __forceinline__ int lds_pad(int rel) { return rel + ((rel >> 6) << 3); }

template <bool PAD = false>
__device__ __forceinline__ long stage_raw(unsigned short* r_s, const unsigned short* base, long n, long first,
                                          int span, int tid, int nthreads) {
    const long ao = (long)(((uintptr_t)base >> 1) & 7);          // samples past a 16-byte boundary at g = 0
    long m = (first + ao) % 8;
    if (m < 0) m += 8;
    const long first_al = first - m;                             // address of sample first_al is 16-byte aligned
    const int nchunk = (int)((first + span - first_al + 7) >> 3);
    for (int c = tid; c < nchunk; c += nthreads) {
        const long g0 = first_al + 8L * c;
        uint4 v;
        if (g0 >= 0 && g0 + 8 <= n) {
            v = *(const uint4*)(base + g0);
        } else {
            unsigned short t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = (g0 + i >= 0 && g0 + i < n) ? base[g0 + i] : (unsigned short)0;
            v.x = t[0] | ((unsigned)t[1] << 16); v.y = t[2] | ((unsigned)t[3] << 16);
            v.z = t[4] | ((unsigned)t[5] << 16); v.w = t[6] | ((unsigned)t[7] << 16);
        }
        *(uint4*)(r_s + (PAD ? lds_pad(8 * c) : 8 * c)) = v;
    }
    return first_al;
}

// ------------------------------------------------------------------------------------------------
// DC sums.  grid (B, S), block 256.  Bytes at even addresses are I, odd are Q (stream start even).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_dc_sum(const uint8_t* __restrict__ raw, long stream_bytes,
                                                StreamState* __restrict__ st) {
    const int s = blockIdx.y;
    const uint8_t* base = raw + (size_t)s * stream_bytes;
    const uintptr_t A = (uintptr_t)base, B = A + stream_bytes;
    const uintptr_t a0 = (A + 15) & ~(uintptr_t)15, b0 = B & ~(uintptr_t)15;
    unsigned int si = 0, sq = 0;  // per-thread sums: < 2^32 for any stream below 16M samples/thread
    unsigned long long ti = 0, tq = 0;
    if (a0 < b0) {
        const long nvec = (long)((b0 - a0) >> 4);
        const uint4* v = (const uint4*)a0;
        const long per_block = (nvec + gridDim.x - 1) / gridDim.x;
        const long v0 = (long)blockIdx.x * per_block;
        long v1 = v0 + per_block;
        if (v1 > nvec) v1 = nvec;
        for (long i = v0 + threadIdx.x; i < v1; i += 256) {
            const uint4 w = v[i];
            // v_sad_u8(x,0,acc) = acc + sum of the 4 bytes of x
            si = __builtin_amdgcn_sad_u8(w.x & 0x00FF00FFu, 0u, si);
            sq = __builtin_amdgcn_sad_u8(w.x & 0xFF00FF00u, 0u, sq);
            si = __builtin_amdgcn_sad_u8(w.y & 0x00FF00FFu, 0u, si);
            sq = __builtin_amdgcn_sad_u8(w.y & 0xFF00FF00u, 0u, sq);
            si = __builtin_amdgcn_sad_u8(w.z & 0x00FF00FFu, 0u, si);
            sq = __builtin_amdgcn_sad_u8(w.z & 0xFF00FF00u, 0u, sq);
            si = __builtin_amdgcn_sad_u8(w.w & 0x00FF00FFu, 0u, si);
            sq = __builtin_amdgcn_sad_u8(w.w & 0xFF00FF00u, 0u, sq);
        }
    }
    ti = si;
    tq = sq;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // unaligned head and tail bytes
        const uintptr_t h1 = a0 < b0 ? a0 : B;
        for (uintptr_t p = A; p < h1; ++p) {
            if (p & 1) tq += *(const uint8_t*)p; else ti += *(const uint8_t*)p;
        }
        if (a0 < b0)
            for (uintptr_t p = b0; p < B; ++p) {
                if (p & 1) tq += *(const uint8_t*)p; else ti += *(const uint8_t*)p;
            }
    }
    // wave reduction then one atomic per wave (integer: order independent, exact)
    for (int off = 32; off > 0; off >>= 1) {
        ti += __shfl_down(ti, off, 64);
        tq += __shfl_down(tq, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&st[s].sum_i, ti);
        atomicAdd(&st[s].sum_q, tq);
    }
}

// raw2iq.m:8  mean = sum(c,1)./size(c,1) : exact integer sums divided once in double
// (the state array was zeroed by hipMemsetAsync; this also writes the non-zero defaults = the
// sentinels the reference functions start from)
__global__ void k_finish_mean(StreamState* st, int S, long n0) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const double n = (double)n0;
    st[s].n0 = n0;
    st[s].mean_re = (double)st[s].sum_i / n;
    st[s].mean_im = (double)st[s].sum_q / n;
    st[s].hit_avg_snr = INFINITY;
    st[s].sampling_ppm1 = INFINITY; st[s].carrier_ppm1 = INFINITY;
    st[s].sampling_ppm2 = INFINITY; st[s].carrier_ppm2 = INFINITY;
    st[s].fcch_is_sentinel = 1;
}

// ------------------------------------------------------------------------------------------------
// raw2iq materialised: out[n] = (I - mean_re) + 1i (Q - mean_im).  grid (B, S).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_raw2iq(const uint8_t* __restrict__ raw, long stream_bytes,
                                                const StreamState* __restrict__ st,
                                                cplx* __restrict__ out, long out_stride) {
    const int s = blockIdx.y;
    const uint8_t* base = raw + (size_t)s * stream_bytes;
    const long n = stream_bytes >> 1;
    const double mr = st[s].mean_re, mi = st[s].mean_im;
    cplx* o = out + (size_t)s * out_stride;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const uchar2 b = *(const uchar2*)(base + 2 * i);
        o[i] = make_double2((double)b.x - mr, (double)b.y - mi);
    }
}

// ------------------------------------------------------------------------------------------------
// Fused front end with decimation: y[j] = f[j*decim], f = filter(coef,1,raw2iq(raw)).
// grid (ceil(Nd/256), S), block 256.  LDS: the block's raw span (decim*256 + ntaps samples).
// Sum order follows filter()'s transposed direct form: oldest tap first, newest last.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_fir_decim_raw(const uint8_t* __restrict__ raw, long stream_bytes,
                                                       const StreamState* __restrict__ st,
                                                       const double* __restrict__ coef, int ntaps,
                                                       int decim, long nd, cplx* __restrict__ out,
                                                       long out_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* c_s = (double*)smem;                                      // ntaps doubles
    unsigned short* r_s = (unsigned short*)(smem + ((ntaps * 8 + 15) & ~15));  // raw samples (I | Q<<8)
    const int s = blockIdx.y;
    const long n = stream_bytes >> 1;
    const unsigned short* base = (const unsigned short*)(raw + (size_t)s * stream_bytes);
    const long j0 = (long)blockIdx.x * 256;
    if (j0 >= nd) return;
    long jn = nd - j0;
    if (jn > 256) jn = 256;
    const long first = j0 * decim - (ntaps - 1);            // first sample index needed (may be < 0)
    const long last = (j0 + jn - 1) * decim;                // last sample index needed
    const int span = (int)(last - first + 1);
    for (int i = threadIdx.x; i < ntaps; i += 256) c_s[i] = coef[i];
    const long first_al = stage_raw<true>(r_s, base, n, first, span, threadIdx.x, 256);
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= jn) return;
    const double mr = st[s].mean_re, mi = st[s].mean_im;
    const long i_out = (j0 + t) * decim;                     // sample index of this output
    double ar = 0.0, ai = 0.0;
    for (int k = ntaps - 1; k >= 0; --k) {
        const long g = i_out - k;
        if (g < 0) continue;                                 // zero initial state
        const unsigned short v = r_s[lds_pad((int)(g - first_al))];
        const double c = c_s[k];
        ar = fma(c, (double)(v & 0xFF) - mr, ar);
        ai = fma(c, (double)(v >> 8) - mi, ai);
    }
    out[(size_t)s * out_stride + j0 + t] = make_double2(ar, ai);
}

// ------------------------------------------------------------------------------------------------
// k_front_fused: ONE pass over the raw bytes for the batch paths.  Per block of 256 kept rows it
//   (1) writes the exact integer I/Q byte sums of its own decim*256 samples (raw2iq.m:8) as a per-block partial,
//   (2) writes y'[j] = sum_k coef[k]*raw[j*decim-k] for its rows -- the FIR of the RAW samples (for
//       linear-phase taps the two samples sharing a tap are added as integers first).
// The detector subtracts mean*sum(coef) on load (DecView in kernels_detect.h): filter() is linear, so
// y = y' - mean*sum_{valid k} coef[k]; this differs from filtering (raw-mean) only in fp64 rounding
// (~1e-15 relative) and feeds nothing but the FCCH coarse detector's threshold decisions.
// grid (ceil(nd/256), S), block 256.  LDS as k_fir_decim_raw.
// ------------------------------------------------------------------------------------------------
#ifndef FF_INFLIGHT
#define FF_INFLIGHT 8
#endif
__global__ void __launch_bounds__(256) k_front_fused(const uint8_t* __restrict__ raw, long stream_bytes,
                                                     unsigned long long* __restrict__ partial,
                                                     const double* __restrict__ coef, int ntaps,
                                                     int decim, long nd, cplx* __restrict__ out,
                                                     long out_stride, int symmetric) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* c_s = (double*)smem;
    unsigned short* r_s = (unsigned short*)(smem + ((ntaps * 8 + 15) & ~15));
    const int s = blockIdx.y;
    const long n = stream_bytes >> 1;
    const unsigned short* base = (const unsigned short*)(raw + (size_t)s * stream_bytes);
    const long j0 = (long)blockIdx.x * 256;
    if (j0 >= nd) return;
    long jn = nd - j0;
    if (jn > 256) jn = 256;
    const long first = j0 * decim - (ntaps - 1);
    // this block owns samples [j0*decim, min((j0+256)*decim, n)) for the sums: stage up to the end of it
    long own_end = (j0 + 256) * decim;
    if (own_end > n) own_end = n;
    const long last = own_end - 1 > (j0 + jn - 1) * decim ? own_end - 1 : (j0 + jn - 1) * decim;
    const int span = (int)(last - first + 1);
    for (int i = threadIdx.x; i < ntaps; i += 256) c_s[i] = coef[i];
    const int t = threadIdx.x;
    // stage [first, first+span) into LDS with 16-byte loads, FF_INFLIGHT in flight per lane, and take the integer
    // I/Q sums (raw2iq.m:8) of the owned samples straight from the registers (v_sad_u8)
    const long ao = (long)(((uintptr_t)base >> 1) & 7);
    long mm = (first + ao) % 8;
    if (mm < 0) mm += 8;
    const long first_al = first - mm;                       // 16-byte aligned in memory
    const int nchunk = (int)((first + span - first_al + 7) >> 3);
    const long o0 = j0 * decim;
    unsigned int si = 0, sq = 0;
    for (int c0 = t; c0 < nchunk; c0 += 256 * FF_INFLIGHT) {
        uint4 v[FF_INFLIGHT];
#pragma unroll
        for (int u = 0; u < FF_INFLIGHT; ++u) {
            const int c = c0 + 256 * u;
            const long g0 = first_al + 8L * c;
            v[u] = make_uint4(0u, 0u, 0u, 0u);
            if (c < nchunk) {
                if (g0 >= 0 && g0 + 8 <= n) {
                    v[u] = *(const uint4*)(base + g0);
                } else {
                    unsigned short q[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) q[i] = (g0 + i >= 0 && g0 + i < n) ? base[g0 + i] : (unsigned short)0;
                    v[u].x = q[0] | ((unsigned)q[1] << 16); v[u].y = q[2] | ((unsigned)q[3] << 16);
                    v[u].z = q[4] | ((unsigned)q[5] << 16); v[u].w = q[6] | ((unsigned)q[7] << 16);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < FF_INFLIGHT; ++u) {
            const int c = c0 + 256 * u;
            if (c >= nchunk) continue;
            const long g0 = first_al + 8L * c;
            *(uint4*)(r_s + lds_pad(8 * c)) = v[u];
            if (g0 >= o0 && g0 + 8 <= own_end) {             // chunk fully owned (the common case)
                const unsigned w4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {                // sample = I | Q<<8: bytes 0,2 are I, bytes 1,3 are Q
                    si = __builtin_amdgcn_sad_u8(w4[i] & 0x00FF00FFu, 0u, si);
                    sq = __builtin_amdgcn_sad_u8(w4[i] & 0xFF00FF00u, 0u, sq);
                }
            } else if (g0 + 8 > o0 && g0 < own_end) {        // ragged edge: sample by sample
                const unsigned w4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const long g = g0 + i;
                    if (g >= o0 && g < own_end) {
                        const unsigned smp = (w4[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu;
                        si += smp & 0xFF;
                        sq += smp >> 8;
                    }
                }
            }
        }
    }
    {
        __shared__ unsigned long long sh_sum[8];
        unsigned long long ti = si, tq = sq;
        for (int off = 32; off > 0; off >>= 1) {
            ti += __shfl_down(ti, off, 64);
            tq += __shfl_down(tq, off, 64);
        }
        if ((t & 63) == 0) { sh_sum[2 * (t >> 6)] = ti; sh_sum[2 * (t >> 6) + 1] = tq; }
        __syncthreads();
        if (t == 0) {     // this block's exact byte sums; consumers add the gridDim.x partials (integers: any order)
            unsigned long long* p = partial + ((size_t)s * gridDim.x + blockIdx.x) * 2;
            p[0] = sh_sum[0] + sh_sum[2] + sh_sum[4] + sh_sum[6];
            p[1] = sh_sum[1] + sh_sum[3] + sh_sum[5] + sh_sum[7];
        }
    }
    // (2) FIR of the raw samples
    if (t >= jn) return;
    const long i_out = (j0 + t) * decim;
    double ar = 0.0, ai = 0.0;
    if (symmetric && i_out >= ntaps - 1) {
        // linear-phase taps (coef[k] == coef[ntaps-1-k], checked on the host): add the two samples that share a
        // tap as INTEGERS first (exact), then one conversion and one FMA per pair -- half the fp64 work
        const int base = (int)(i_out - (ntaps - 1) - first_al);          // oldest sample of this output
        const int half = ntaps >> 1;
#pragma unroll 8
        for (int k = 0; k < half; ++k) {                 // unrolled: 16 independent LDS reads in flight
            const unsigned a = r_s[lds_pad(base + k)], b = r_s[lds_pad(base + ntaps - 1 - k)];
            const double c = c_s[k];
            ar = fma(c, (double)((a & 0xFF) + (b & 0xFF)), ar);
            ai = fma(c, (double)((a >> 8) + (b >> 8)), ai);
        }
        if (ntaps & 1) {
            const unsigned a = r_s[lds_pad(base + half)];
            ar = fma(c_s[half], (double)(a & 0xFF), ar);
            ai = fma(c_s[half], (double)(a >> 8), ai);
        }
    } else {
        for (int k = ntaps - 1; k >= 0; --k) {                   // oldest tap first; zero initial state
            const long g = i_out - k;
            if (g < 0) continue;
            const unsigned short v = r_s[lds_pad((int)(g - first_al))];
            const double c = c_s[k];
            ar = fma(c, (double)(v & 0xFF), ar);
            ai = fma(c, (double)(v >> 8), ai);
        }
    }
    out[(size_t)s * out_stride + j0 + t] = make_double2(ar, ai);
}

// ------------------------------------------------------------------------------------------------
// k_front_fast<NT>: k_front_fused for the production geometry -- NT odd linear-phase taps (NT <= 49), decimation
// 64, every stream starting on a 16-byte boundary.  Same outputs (per-block byte sums + FIR of the raw samples),
// but the VALU work per sample is a third: the block's 256 rows of 64 samples are staged through LDS only to
// turn the coalesced global reads into one contiguous 128-byte row per lane; each lane then holds its row in
// 32 registers and
//   * takes the row's I/Q byte sums with v_dot4_u32_u8 (2 per dword),
//   * forms each tap pair's integer sum a+b with two v_dot4_u32_u8 (compile-time byte selectors), converts
//     once and issues one fp64 FMA per pair and component.
// Row t of block j0 = samples [64(j0+t)-48, 64(j0+t)+16): it contains the NT taps of output j0+t (samples
// 64j-NT+1 .. 64j) and the rows tile the stream, so the byte sums of the rows are the stream's sums (the last
// block adds the <= 48 samples past its last row).  Samples outside [0, n) are staged as zeros: zero initial
// state of filter(), and nothing for the sums.
// grid (ceil(nd/256), S), block 256.  LDS: 2048 chunks of 8 samples, swizzled (32 KB); partial sums: 4 per block.
// ------------------------------------------------------------------------------------------------
#define FFAST_CHUNKS (2048 + 6)
// LDS slot (in 16-byte chunks) of chunk i of row r: an XOR swizzle instead of padding.  Any 16 consecutive lanes --
// of the staging stores (two rows, eight chunks each) and of the row loads (16 rows, one chunk index) -- touch
// 16 different 16-byte bank groups, and the 2048 chunks take exactly 32 KB: five workgroups per CU instead of four.
__device__ __forceinline__ int ffast_slot(int c) { return (c & ~7) | ((c & 7) ^ ((c >> 4) & 7)); }
// 16 raw bytes that this launch reads exactly once: a non-temporal load (no claim on L2 / Infinity Cache lines that the
// decimated rows and the detector's tables can use)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream16(const uint4* p) {
    const u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// ... and 16 bytes of an output stream nobody in this launch reads back (r_correct, 1 GB per 64-stream step): stored non-temporal,
// so that the step's input survives in the Infinity Cache (stream mode 0.766 -> 0.738 ms: the next step's front kernel 44 -> 23.5 us)
__device__ __forceinline__ void st_stream16(cplx* p, const cplx v) {
    typedef double f64x2_t __attribute__((ext_vector_type(2)));
    f64x2_t o;
    o.x = v.x; o.y = v.y;
    __builtin_nontemporal_store(o, (f64x2_t*)p);
}
__device__ __forceinline__ uint4 ffast_chunk(const unsigned short* __restrict__ base, long g0, long n) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);                   // 8 samples from g0, zeros outside [0, n)
    if (g0 >= 0 && g0 + 8 <= n) {
        v = *(const uint4*)(base + g0);
    } else if (g0 + 8 > 0 && g0 < n) {                      // straddles an end of the stream
        unsigned short q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = (g0 + i >= 0 && g0 + i < n) ? base[g0 + i] : (unsigned short)0;
        v.x = q[0] | ((unsigned)q[1] << 16); v.y = q[2] | ((unsigned)q[3] << 16);
        v.z = q[4] | ((unsigned)q[5] << 16); v.w = q[6] | ((unsigned)q[7] << 16);
    }
    return v;
}

template <int NT, bool SYM>
__global__ void __launch_bounds__(256) k_front_fast(const uint8_t* __restrict__ raw, long stream_bytes,
                                                    unsigned long long* __restrict__ partial,
                                                    const double* __restrict__ coef, long nd,
                                                    cplx* __restrict__ out, long out_stride, int nt_loads) {
    static_assert((NT & 1) == 1 && NT <= 49, "odd tap count that fits the 56-sample register window");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short* r_s = (unsigned short*)smem;
    // the one bandwidth-bound kernel issues ahead of whatever compute-bound kernel shares its CUs (the scanner pipeline runs the
    // previous stage's detector beside it): its few instructions turn into memory requests sooner -- 12 800 captures 3.93 -> 3.87 ms
    __builtin_amdgcn_s_setprio(3);
    const int s = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long n = stream_bytes >> 1;
    const unsigned short* base = (const unsigned short*)(raw + (size_t)s * stream_bytes);
    const long j0 = (long)blockIdx.x * 256;
    long jn = nd - j0;
    if (jn > 256) jn = 256;
    const long first_al = 64 * j0 - 48;                     // sample at row 0, position 0 (16-byte aligned in memory)
    // Each WAVE stages its own 64 rows (512 chunks, one contiguous 8 KB piece of the stream) and reads back only
    // those: LDS operations of one wave execute in order, so no block barrier is needed and the four waves of a
    // block drift apart freely (loads of one overlap the arithmetic of another).
    {
        uint4 v[8];
        // every block but the first and the last of a capture lies wholly inside it: eight plain 16-byte loads off one lane
        // pointer (the per-chunk range tests of ffast_chunk were an eighth of this kernel's vector instructions)
        if (first_al >= 0 && first_al + 8L * 2048 <= n) {       // (block-uniform)
            const uint4* p = (const uint4*)(base + first_al) + (512 * wave + lane);
            if (nt_loads) {                                     // (launch-uniform) a batch that no cache can hold: see front_fused()
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = ld_stream16(p + 64 * u);
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = p[64 * u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ffast_chunk(base, first_al + 8L * (512 * wave + lane + 64 * u), n);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) *(uint4*)(r_s + 8 * ffast_slot(512 * wave + lane + 64 * u)) = v[u];
    }
    unsigned w[32];                                         // this lane's row
    {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint4 q = *(const uint4*)(r_s + 8 * ffast_slot(8 * t + u));
            w[4 * u] = q.x; w[4 * u + 1] = q.y; w[4 * u + 2] = q.z; w[4 * u + 3] = q.w;
        }
    }
    unsigned si = 0, sq = 0;                                // sample = I | Q<<8, two samples per dword
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        si = __builtin_amdgcn_udot4(w[i], 0x00010001u, si, false);
        sq = __builtin_amdgcn_udot4(w[i], 0x01000100u, sq, false);
    }
    if (blockIdx.x == gridDim.x - 1 && t < FFAST_CHUNKS - 2048) {   // the samples past the last row of the stream's last block
        const uint4 q = ffast_chunk(base, first_al + 8L * (2048 + t), n);
        const unsigned e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            si = __builtin_amdgcn_udot4(e[i], 0x00010001u, si, false);
            sq = __builtin_amdgcn_udot4(e[i], 0x01000100u, sq, false);
        }
    }
    {   // exact byte sums per wave; consumers add the 4*gridDim.x partials of the stream (integers: any order).  A lane's sum is
        // below 2^15 and a wave's below 2^21: 32-bit adds through DPP moves (seven instructions per sum, no LDS-crossbar shuffles)
        const unsigned ti = wave_sum_u32(si), tq = wave_sum_u32(sq);
        if (lane == 0) {
            unsigned long long* p = partial + (((size_t)s * gridDim.x + blockIdx.x) * 4 + wave) * 2;
            p[0] = ti;
            p[1] = tq;
        }
    }
    if (t >= jn) return;
    const double* __restrict__ c_s = coef;                  // uniform, compile-time indices: scalar loads
    // ---- FIR of the raw samples.  Output sample 64j sits at row position 48, tap k at position 48 - k.
    double ar = 0.0, ai = 0.0;
    if (SYM) {
        // exactly symmetric taps (coef[k] == coef[NT-1-k], checked on the host): oldest pair first, the two samples
        // of a pair added as integers (the order of k_front_fused's symmetric loop)
#pragma unroll
        for (int k = 0; k < NT / 2; ++k) {
            const int ia = 49 - NT + k, ib = 48 - k;
            const unsigned selIa = (ia & 1) ? 0x00010000u : 0x00000001u, selIb = (ib & 1) ? 0x00010000u : 0x00000001u;
            const unsigned pi = __builtin_amdgcn_udot4(w[ib >> 1], selIb, __builtin_amdgcn_udot4(w[ia >> 1], selIa, 0u, false), false);
            const unsigned pq = __builtin_amdgcn_udot4(w[ib >> 1], selIb << 8, __builtin_amdgcn_udot4(w[ia >> 1], selIa << 8, 0u, false), false);
            const double c = c_s[k];
            ar = fma(c, (double)pi, ar);
            ai = fma(c, (double)pq, ai);
        }
        const int im = 49 - NT + NT / 2;                    // the middle tap
        const unsigned smp = (w[im >> 1] >> ((im & 1) * 16)) & 0xFFFFu;
        ar = fma(c_s[NT / 2], (double)(smp & 0xFF), ar);
        ai = fma(c_s[NT / 2], (double)(smp >> 8), ai);
    } else {
        // any taps: oldest tap first, one conversion and one FMA per tap and component (k_front_fused's order)
#pragma unroll
        for (int k = NT - 1; k >= 0; --k) {
            const int ip = 48 - k;
            const unsigned smp = w[ip >> 1] >> ((ip & 1) * 16);
            const double c = c_s[k];
            ar = fma(c, (double)(smp & 0xFFu), ar);
            ai = fma(c, (double)((smp >> 8) & 0xFFu), ai);
        }
    }
    out[(size_t)s * out_stride + j0 + t] = make_double2(ar, ai);
}

// ------------------------------------------------------------------------------------------------
// Synthetic-input utility (benchmarks and tests; not part of the reference's path): expand K seeded base captures
// into D distinct captures on the device.  Capture u = base[u mod K] rotated by shift(u) samples, every I and Q
// byte dithered by -1/0/+1 drawn from a counter-based hash of (seed, u, sample index), clipped to 0..255.  The
// host twin (synth.expand_capture) produces the same bytes, so any capture of a 131 GB batch can be handed to
// the CPU oracle without ever existing on the host.  grid (blocks, D), block 256; 8 samples (16 bytes) per lane-step.
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline unsigned long long synth_mix64(unsigned long long x) {      // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline unsigned long long synth_shift(unsigned long long seed, unsigned long long unit, unsigned long long n) {
    return synth_mix64(seed ^ (unit * 0xD1B54A32D192ED03ull)) % n;
}
__global__ void __launch_bounds__(256) k_synth_expand(const uint8_t* __restrict__ base, int K, long n,
                                                       uint8_t* __restrict__ out, long first_unit,
                                                       unsigned long long seed) {
    const long u = first_unit + blockIdx.y;
    const unsigned short* b = (const unsigned short*)(base + (size_t)(u % K) * 2 * n);
    unsigned short* o = (unsigned short*)(out + (size_t)blockIdx.y * 2 * n);
    const unsigned long long sh = synth_shift(seed, (unsigned long long)u, (unsigned long long)n);
    const unsigned long long key = synth_mix64(seed + 0x632BE59BD9B4E019ull * (unsigned long long)u);
    for (long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i0 < n; i0 += (long)gridDim.x * 256 * 8) {
        unsigned short v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long i = i0 + j;
            if (i >= n) { v[j] = 0; continue; }
            long src = i + (long)sh;
            if (src >= n) src -= n;
            const unsigned smp = b[src];
            const unsigned long long z = synth_mix64(key ^ (unsigned long long)i);
            int I = (int)(smp & 0xFF) + (int)((z & 3) == 0) - (int)((z & 3) == 1);
            int Q = (int)(smp >> 8) + (int)(((z >> 2) & 3) == 0) - (int)(((z >> 2) & 3) == 1);
            I = I < 0 ? 0 : (I > 255 ? 255 : I);
            Q = Q < 0 ? 0 : (Q > 255 ? 255 : Q);
            v[j] = (unsigned short)(I | (Q << 8));
        }
        if (i0 + 8 <= n && (((uintptr_t)(o + i0)) & 15) == 0) {
            uint4 w;
            w.x = v[0] | ((unsigned)v[1] << 16); w.y = v[2] | ((unsigned)v[3] << 16);
            w.z = v[4] | ((unsigned)v[5] << 16); w.w = v[6] | ((unsigned)v[7] << 16);
            *(uint4*)(o + i0) = w;
        } else {
            for (int j = 0; j < 8 && i0 + j < n; ++j) o[i0 + j] = v[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// filter(coef,1,s) on a complex array, keeping rows 1:decim:end.  grid (ceil(nd/256), D).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_fir_arr(const cplx* __restrict__ in, long in_stride, long n,
                                                 const double* __restrict__ coef, int ntaps, int decim,
                                                 long nd, cplx* __restrict__ out, long out_stride) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= nd) return;
    const cplx* x = in + (size_t)blockIdx.y * in_stride;
    const long i_out = j * decim;
    double ar = 0.0, ai = 0.0;
    for (int k = ntaps - 1; k >= 0; --k) {
        const long g = i_out - k;
        if (g < 0) continue;
        const cplx v = x[g];
        const double c = coef[k];
        ar = fma(c, v.x, ar);
        ai = fma(c, v.y, ai);
    }
    out[(size_t)blockIdx.y * out_stride + j] = make_double2(ar, ai);
}

// ------------------------------------------------------------------------------------------------
// k_gather: evaluate [start, start+len) of level `level` of every stream's lazy chain.
//   list mode : window w of stream s starts at st[s].win_start[w]   (grid.x = max windows)
//   tile mode : window w starts at w*len                            (grid.x = ceil(max n / len))
// LDS: two ping-pong buffers of (len+8) complex doubles + the raw bytes for SRC_RAW.
// ------------------------------------------------------------------------------------------------
struct GatherArgs {
    int src_kind, level, len, tiles, ntaps, pad;   // pad != 0: compact_xs (see gather_carve)
    const uint8_t* raw; long raw_stride;     // bytes per stream
    const cplx* arr;    long arr_stride;     // elements per stream
    const double* coef;
    cplx* dst; long dst_stream_stride, dst_win_stride;
    // raw sources: level-0 samples the fine search already filtered (window h of stream s covers level-0 indices
    // [st.fine_ws[h], st.fine_ws[h] + l0_len) at l0 + s*l0_stream_stride + h*l0_win_stride); nullptr: none
    const cplx* l0; long l0_stream_stride, l0_win_stride; int l0_len, pad2;
};

// State fields read past the vector L1 (relaxed agent-scope loads = `sc1`): what a workgroup needs when ANOTHER workgroup
// of the same launch may have rewritten the field (k_post_chain's in-launch decision steps).
__device__ __forceinline__ int st_i32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long st_i64(const long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double st_f64(const double* p) {
    return __longlong_as_double(__hip_atomic_load((const long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ long level_len(const StreamState* st, int level) {
    return level == 0 ? st->n0 : st->op[level].n;
}

// LDS position of staged input sample p: one 16-byte pad after every 4 samples, so that lanes reading
// samples 4 apart (the 4-outputs-per-lane FIR below) hit distinct banks with ds_read_b128.
__device__ __forceinline__ int xs_pad(int p) { return p + (p >> 2); }

// ------------------------------------------------------------------------------------------------
// filter(coef,1,.) for FOUR consecutive outputs i0 .. i0+3 from the padded LDS copy xq (xs_pad indexing; i0 a multiple of
// 4): y[i] = sum_k coef[k] x[i-k], every accumulator taking its taps oldest first (transposed direct-form order) -- the
// one FIR loop of gather_core and of k_fine_cert's fused window build, so both give bit-identical level-0 samples.
// NTP > 0: the tap count is a compile-time constant (the production fir1(46) filter): two trips per iteration with known
// bounds.  (Unrolled completely the compiler hoists every coefficient and sample load: 256 registers and scratch.)
// Tried and dropped (round 3): the taps through scalar loads from constant-address-space memory instead of LDS -- a third
// of this loop's LDS bytes are coefficient broadcasts, and the 1024-stream step gained 2.5 % -- but SMEM shares the LDS
// counter (lgkmcnt), every wait for a tap became a wait for the sample reads in flight, and the 64-stream step lost 3 us;
// pipelining it by hand (next trip's samples and taps requested before this trip's FMAs) cost registers the 80-register
// burst kernels do not have (scratch in the loop).
// PAD = false: the input is staged without xs_pad's bank padding (gather_carve's compact_xs).
// ------------------------------------------------------------------------------------------------
// UNR: trips of the four-tap loop the compiler may overlap (2: the next trip's LDS reads under this trip's FMAs, ~20 more
// registers; 1 where the caller has none to spare).
// taps of fir4_lds in LDS: rev[j] = coef[ntaps-1-j] (oldest tap first), zero-padded to a multiple of four
__host__ __device__ inline int fir_taps_padded(int ntaps) { return (ntaps + 3) & ~3; }
__device__ __forceinline__ void fir_stage_taps(double* c_s, const double* __restrict__ coef, int ntaps, int first, int step) {
    for (int i = first; i < fir_taps_padded(ntaps); i += step) c_s[i] = i < ntaps ? coef[ntaps - 1 - i] : 0.0;
}
template <int NTP, bool PAD = true, int UNR = 2>
__device__ __forceinline__ void fir4_lds(const cplx* __restrict__ xq, const double* __restrict__ c_s, int i0, int ntp_rt,
                                         cplx* y0, cplx* y1, cplx* y2, cplx* y3) {
#define XSP(p_) (PAD ? xs_pad(p_) : (p_))
    const int ntp = NTP > 0 ? NTP : ntp_rt;
    const int ntp4 = (ntp + 3) & ~3;
    double ar0 = 0.0, ai0 = 0.0, ar1 = 0.0, ai1 = 0.0, ar2 = 0.0, ai2 = 0.0, ar3 = 0.0, ai3 = 0.0;
    cplx w0 = xq[XSP(i0)], w1 = xq[XSP(i0 + 1)], w2 = xq[XSP(i0 + 2)], w3 = xq[XSP(i0 + 3)];
#define GSMCAL_FIR_TAP(C, A, B, D, E)                                   \
    ar0 = fma(C, A.x, ar0); ai0 = fma(C, A.y, ai0);                     \
    ar1 = fma(C, B.x, ar1); ai1 = fma(C, B.y, ai1);                     \
    ar2 = fma(C, D.x, ar2); ai2 = fma(C, D.y, ai2);                     \
    ar3 = fma(C, E.x, ar3); ai3 = fma(C, E.y, ai3);
    // four taps per trip: the next four samples are one aligned, contiguous 64-byte group (i0 is a multiple of 4),
    // fetched together, and so are the four taps (c_s holds them REVERSED -- oldest tap first -- and zero-padded to a multiple
    // of four: fir_stage_taps; two 16-byte reads per trip instead of four 8-byte ones, and no remainder loop:
    // fma(0, x, acc) = acc for the finite samples the callers stage behind the span).  Every accumulator still takes its
    // taps in the same order (oldest first), so the sums are bit-identical to a one-tap loop.
#pragma unroll UNR
    for (int t = 0; t < ntp4; t += 4) {
        const cplx* nx = xq + XSP(i0 + t + 4);
        const double c0 = c_s[t], c1 = c_s[t + 1], c2 = c_s[t + 2], c3 = c_s[t + 3];   // (before the samples: LDS returns in order, and the first taps need only these)
        const cplx n0s = nx[0], n1s = nx[1], n2s = nx[2], n3s = nx[3];
        GSMCAL_FIR_TAP(c0, w0, w1, w2, w3)
        GSMCAL_FIR_TAP(c1, w1, w2, w3, n0s)
        GSMCAL_FIR_TAP(c2, w2, w3, n0s, n1s)
        GSMCAL_FIR_TAP(c3, w3, n0s, n1s, n2s)
        w0 = n0s; w1 = n1s; w2 = n2s; w3 = n3s;
    }
#undef GSMCAL_FIR_TAP
#undef XSP
    *y0 = make_double2(ar0, ai0); *y1 = make_double2(ar1, ai1); *y2 = make_double2(ar2, ai2); *y3 = make_double2(ar3, ai3);
}

// LDS layout of gather_core (byte offsets from the dynamic LDS base); also used by the host to size launches.
struct GatherCarve {
    size_t bufn;          // elements per window buffer (0: none)
    size_t off_region1;   // buf1 / xs
    size_t off_coef, off_raw, off_rot, total;
};
#define GC_ROT_A 72                     /* rotator table: S | A[0..GC_ROT_A) | B[0..32): windows of up to 32*GC_ROT_A samples */
// compact_xs: the complex input of the raw-source FIR is staged without bank-conflict padding (5 KB less for a burst window;
// k_post_chain's burst stages, where that path is the rare fallback for a burst outside every fine window)
__host__ __device__ inline GatherCarve gather_carve(int len, int level, int kind, int ntaps, bool to_lds, bool compact_xs = false) {
    GatherCarve g;
    g.bufn = (level >= 1 || to_lds) ? (size_t)len + 40 : 0;          // +40: room for B[37][N2+1] of the fused kernels
    const size_t span_max = (size_t)len + 8 + ntaps + 24;
    const size_t xs_n = kind == SRC_RAW ? (compact_xs ? span_max + 16 : span_max + span_max / 4 + 16) : 0;   // (padded) input, raw sources only
    const bool need_buf1 = level >= 2 || (to_lds && level >= 1) || to_lds;
    size_t r1 = need_buf1 ? g.bufn : 0;
    if (xs_n > r1) r1 = xs_n;
    g.off_region1 = g.bufn * 16;
    g.off_coef = g.off_region1 + r1 * 16;
    g.off_raw = g.off_coef + (size_t)fir_taps_padded(ntaps) * 8;
    g.total = (g.off_raw + ((span_max + 7) & ~(size_t)7) * 2 + 15) & ~(size_t)15;
    // the rotator table of a derotation level lives where the raw bytes were (dead once level 0 exists); array sources
    // have no raw region, so it is appended there
    g.off_rot = g.off_raw;
    const size_t rot_end = g.off_rot + (size_t)(1 + GC_ROT_A + 32) * 16;
    if (level >= 2 && rot_end > g.total) g.total = (rot_end + 15) & ~(size_t)15;
    return g;
}

// gather_core<NT>: the body shared by k_gather and the fused per-window kernels (kernels_estim.h).
// to_lds = false: the window is written to global memory (a.dst).  to_lds = true: the last level
// stays in LDS and its address is returned (nullptr if this block has no window).
// The block's window index is `widx`, its stream `s`.  smem: the dynamic LDS base, carved as
//   buf0 | buf1 (aliased with the padded complex input xs) | coef | raw ushorts     (GatherCarve).
// FIR_UNR: trips of the raw-source FIR loop the compiler may overlap (fir4_lds' UNR): 2 where the caller's tap count is a
// compile-time constant and its register budget allows (the reference-geometry instantiations), else 1.
template <int NT, int KID = -1, int FIR_UNR = 1>
__device__ __forceinline__ cplx* gather_core(const StreamState* __restrict__ sts, const GatherArgs& a,
                                             unsigned char* smem, int widx, int s, bool to_lds) {
#define GC_STAMP(i) do { if (KID >= 0) DEV_STAMP(KID, blockIdx.y * gridDim.x + blockIdx.x, i); } while (0)
    // LDS carve (GatherCarve, shared with the host and the fused kernels):
    //   buf0 | region1 = buf1 ALIASED WITH xs (padded complex input) | coef | raw ushorts
    // xs is dead once level 0 is in buf0, and buf1 is first written at level 1, so they share storage.
    const bool compact = a.pad != 0;
    const GatherCarve gc = gather_carve(a.len, a.level, a.src_kind, a.ntaps, to_lds, compact);
    cplx* buf0 = (cplx*)smem;
    cplx* buf1 = (cplx*)(smem + gc.off_region1);
    cplx* xs = buf1;
    double* c_s = (double*)(smem + gc.off_coef);
    unsigned short* r_s = (unsigned short*)(smem + gc.off_raw);
    const StreamState* st = sts + s;
    const int level = a.level;
    const int tid = threadIdx.x;
    // The plan of this window -- where it starts and which range of every lower level it needs -- is a chain of small
    // dependent decisions on a handful of state fields.  Only the first wave works it out (every field fetched with
    // independent loads up front; one round trip to L2 per decision was a quarter of this function's time) and hands it
    // to the others through LDS: NT/64 waves stepping through the same serial code only compete for the issue slots.
    __shared__ long pl_lo[NLEVELS], pl_hi[NLEVELS], pl_n0, pl_start;
    __shared__ double pl_param[NLEVELS], pl_mr, pl_mi;
    __shared__ int pl_type[NLEVELS], pl_L, pl_l0h;
    __shared__ long pl_l0off;
    // ... which meanwhile fetch the filter taps
    if (a.src_kind != SRC_ARR && NT > 64 && tid >= 64)
        fir_stage_taps(c_s, a.coef, a.ntaps, tid - 64, NT - 64);
    if (tid < 64) {
        // (loads that bypass the vector L1: inside a fused launch these fields were rewritten by another workgroup's
        // decision step -- write-through stores -- since this CU may last have read them; L2-served, same cost as plain)
        const int pre_nwin = st_i32(&st->n_win);
        const long pre_n0 = st_i64(&st->n0);
        const long pre_ws = (!a.tiles && widx >= 0 && widx < MAXH) ? st_i64(&st->win_start[widx]) : 0;
        const double pre_mr = st_f64(&st->mean_re), pre_mi = st_f64(&st->mean_im);
        int op_type[NLEVELS];
        double op_param[NLEVELS];
        long lvl_n[NLEVELS];
        lvl_n[0] = pre_n0; op_type[0] = OP_NONE; op_param[0] = 0.0;
#pragma unroll
        for (int j = 1; j < NLEVELS; ++j) { op_type[j] = st_i32(&st->op[j].type); op_param[j] = st_f64(&st->op[j].param); lvl_n[j] = st_i64(&st->op[j].n); }
#ifdef GSMCAL_DEVTIMING
        __builtin_amdgcn_s_waitcnt(0);
        GC_STAMP(14);
#endif
        long start = 0, L = 0;
        if (a.tiles) {
            const long nq = lvl_n[level];
            start = (long)widx * a.len;
            L = start >= nq ? 0 : (nq - start < a.len ? nq - start : a.len);
        } else if (widx < pre_nwin) {
            start = pre_ws;
            L = a.len;
        }
        // backward range propagation
        long clo = start, chi = start + L - 1;
#pragma unroll
        for (int j = NLEVELS - 1; j >= 1; --j) {
            if (j <= level) {
                if (tid == 0) { pl_lo[j] = clo; pl_hi[j] = chi; }
                if (op_type[j] == OP_LERP) {
                    const double f = op_param[j];
                    const long nprev = lvl_n[j - 1];
                    clo = (long)floor((double)clo * f);
                    const long h = (long)floor((double)chi * f) + 1;
                    chi = h > nprev - 1 ? nprev - 1 : h;
                }
            }
        }
        // a window the fine search has filtered already and that holds all of [clo, chi]
        int l0h = -1;
        long l0off = 0;
        if (a.l0 && a.src_kind == SRC_RAW && L > 0) {
            const int nf = st_i32(&st->n_fine_ws);
            const long fw = (tid < nf && tid < MAXH) ? st_i64(&st->fine_ws[tid]) : 0;
            const unsigned long long m = __ballot(tid < nf && tid < MAXH && fw <= clo && chi < fw + a.l0_len);
            if (m) {
                l0h = __ffsll((long long)m) - 1;
                l0off = clo - __shfl(fw, l0h, 64);
            }
        }
        if (tid == 0) {
            pl_l0h = l0h; pl_l0off = l0off;
            pl_lo[0] = clo; pl_hi[0] = chi;
            pl_n0 = pre_n0; pl_start = start; pl_L = (int)L; pl_mr = pre_mr; pl_mi = pre_mi;
#pragma unroll
            for (int j = 1; j < NLEVELS; ++j) { pl_type[j] = op_type[j]; pl_param[j] = op_param[j]; }
        }
    }
    if (a.src_kind != SRC_ARR && NT <= 64)
        fir_stage_taps(c_s, a.coef, a.ntaps, tid, NT);
    GC_STAMP(15);
    __syncthreads();
    const int L = pl_L;
    if (L <= 0) return nullptr;                          // block-uniform: no window for this block
    const long start = pl_start, pre_n0 = pl_n0;
    const double pre_mr = pl_mr, pre_mi = pl_mi;
    cplx* dst = a.tiles ? a.dst + (size_t)s * a.dst_stream_stride + start
                        : a.dst + (size_t)s * a.dst_stream_stride + (size_t)widx * a.dst_win_stride;
    // ---- level 0 ----
    const long lo0 = pl_lo[0], hi0 = pl_hi[0];
    const int cnt0 = (int)(hi0 - lo0 + 1);
    cplx* out0 = (level == 0 && !to_lds) ? dst : buf0;
    if (a.src_kind == SRC_ARR) {
        const cplx* x = a.arr + (size_t)s * a.arr_stride;
        const long n0 = pre_n0;
        for (int i = tid; i < cnt0; i += NT) {
            const long g = lo0 + i;
            out0[i] = (g >= 0 && g < n0) ? x[g] : make_double2(0.0, 0.0);
        }
    } else if (pl_l0h >= 0) {
        // the same filter outputs (same taps, same order of additions) are in the fine search's window buffer
        const cplx* x = a.l0 + (size_t)s * a.l0_stream_stride + (size_t)pl_l0h * a.l0_win_stride + pl_l0off;
        for (int i = tid; i < cnt0; i += NT) out0[i] = x[i];
    } else {
        const unsigned short* base = (const unsigned short*)(a.raw + (size_t)s * a.raw_stride);
        const long n0 = pre_n0;
        const int ntp = a.ntaps;
        const long first = lo0 - (ntp - 1);
        const int span = cnt0 + ntp - 1;
        GC_STAMP(9);
        const long first_al = stage_raw(r_s, base, n0, first, span, tid, NT);
        __syncthreads();
        GC_STAMP(10);
        // raw2iq.m:6-8 on the staged span: (I - mean) + 1i (Q - mean); zero before the stream starts
        // (filter()'s zero initial state) and past its end
        const double mr = pre_mr, mi = pre_mi;
        const int off = (int)(first - first_al);
        for (int i = tid; i < span + 8; i += NT) {
            const long g = first + i;
            cplx v = make_double2(0.0, 0.0);
            if (i < span && g >= 0 && g < n0) {
                const unsigned short q = r_s[off + i];
                v = make_double2((double)(q & 0xFF) - mr, (double)(q >> 8) - mi);
            }
            xs[compact ? i : xs_pad(i)] = v;
        }
        __syncthreads();
        GC_STAMP(11);
        // filter(coef,1,.) : y[i] = sum_k coef[k] x[i-k], accumulated oldest tap first (transposed
        // direct form order).  Each lane produces 4 consecutive outputs from a sliding register window.
        for (int i0 = 4 * tid; i0 < cnt0; i0 += 4 * NT) {
            cplx y0, y1, y2, y3;
            // (the run-time-count form, one trip at a time: the two-trip forms need ~20 more registers than the 80 the fused
            // per-burst kernels have -- scratch in the loop, and a kernel with ANY scratch starts its workgroups later;
            // k_fine_cert, at 108 registers, uses fir4_lds<47>)
            if (compact) fir4_lds<0, false, FIR_UNR>(xs, c_s, i0, ntp, &y0, &y1, &y2, &y3);
            else fir4_lds<0, true, FIR_UNR>(xs, c_s, i0, ntp, &y0, &y1, &y2, &y3);
            out0[i0] = y0;
            if (i0 + 1 < cnt0) out0[i0 + 1] = y1;
            if (i0 + 2 < cnt0) out0[i0 + 2] = y2;
            if (i0 + 3 < cnt0) out0[i0 + 3] = y3;
        }
    }
    GC_STAMP(12);
    // ---- levels 1..level ----
    cplx* src = buf0;
    cplx* other = buf1;
    for (int j = 1; j <= level; ++j) {
        __syncthreads();
        cplx* o = (j == level && !to_lds) ? dst : other;
        const long lo_j = pl_lo[j];
        const int cnt = (int)(pl_hi[j] - lo_j + 1);
        const int type = pl_type[j];
        const double p = pl_param[j];
        const long plo = pl_lo[j - 1], phi_ = pl_hi[j - 1];
        if (type == OP_LERP) {
            for (int i = threadIdx.x; i < cnt; i += NT) {
                const long k = lo_j + i;
                const double xq = (double)k * p;            // interp_seq = (0:max_len-1)'.*(1+e)
                const long i0 = (long)floor(xq);
                const long i1 = i0 + 1 > phi_ ? phi_ : i0 + 1;  // beyond the last sample the weight is 0
                const double t = xq - (double)i0;
                const cplx v0 = src[i0 - plo], v1 = src[i1 - plo];
                o[i] = make_double2(v0.x + t * (v1.x - v0.x), v0.y + t * (v1.y - v0.y));
            }
        } else if (type == OP_MIX && cnt <= 32 * GC_ROT_A) {
            // exp(1i*(0:len-1)'*comp_phase_rotate) from a two-level rotator table of the window:
            // exp(1i*k*c) = S * A[(k-k0)>>5] * B[(k-k0)&31], S = exp(1i*fl(k0*c)), A[q] = exp(1i*fl(32q*c)), B[m] = exp(1i*fl(m*c))
            // -- 1 + ceil(cnt/32) + 32 accurate sincos per window instead of one per sample.  The reference rounds k*c once;
            // the three rounded pieces differ from that by < 2 ulp(k*c) ~ 3e-11 rad at k = 1e6 (see k_stream_tile).
            cplx* T = (cplx*)(smem + gc.off_rot);
            const int na = (cnt + 31) >> 5;
            for (int i = threadIdx.x; i < 1 + na + 32; i += NT) {
                const double arg = i == 0 ? (double)lo_j * p : (i <= na ? (double)(32 * (i - 1)) * p : (double)(i - 1 - na) * p);
                double sn, cs;
                sincos_large(arg, &sn, &cs);
                T[i == 0 ? 0 : (i <= na ? i : 1 + GC_ROT_A + (i - 1 - na))] = make_double2(cs, sn);
            }
            __syncthreads();
            for (int i = threadIdx.x; i < cnt; i += NT)
                o[i] = cmul(src[i], cmul(cmul(T[0], T[1 + (i >> 5)]), T[1 + GC_ROT_A + (i & 31)]));
        } else if (type == OP_MIX) {
            for (int i = threadIdx.x; i < cnt; i += NT) {
                const long k = lo_j + i;
                double sn, cs;
                sincos_large((double)k * p, &sn, &cs);      // exp(1i*(0:len-1)'*comp_phase_rotate): k*p rounded once
                o[i] = cmul(src[i], make_double2(cs, sn));
            }
        } else {
            for (int i = threadIdx.x; i < cnt; i += NT) o[i] = src[i];
        }
        cplx* tmp = src; src = other; other = tmp;
        (void)tmp;
    }
    GC_STAMP(13);
#undef GC_STAMP
    return to_lds ? src : nullptr;   // after the last swap `src` is the buffer written last
}

// ------------------------------------------------------------------------------------------------
// k_stream_tile: the corrected stream r_correct (gsm_sync_demod.m:118-120 hands it on to SCH_demod), one tile of
// ST_TILE output samples per workgroup -- the 18 B/sample mode of the batch path.  Same arithmetic per sample as
// gather_core at level 4 (raw -> raw2iq -> filter -> interp1 -> exp derotation -> interp1 -> exp derotation), organised
// for throughput instead of generality:
//   * three passes through LDS instead of five: FIR | lerp + derotation | lerp + derotation -> global;
//   * the derotation exp(1i*k*c) (FCCH_fine_correction.m:165, carrier_correct_post_SCH.m:83) comes from a per-tile
//     two-level rotator table, exp(1i*k*c) = S * A[(k-k0)>>5] * B[(k-k0)&31] with S = exp(1i*fl(k0*c)),
//     A[j] = exp(1i*fl(32j*c)), B[m] = exp(1i*fl(m*c)) -- 66 accurate sincos per tile instead of two per sample.  The
//     reference rounds k*c once; the three rounded pieces differ from that by < 2 ulp(k*c) ~ 3e-11 rad at k = 1e6,
//     far inside the 2e-8 stream tolerance (the argument of the reference's own exp carries the same uncertainty
//     from the estimated c).
// Streams whose chain is not (lerp, mix, lerp|copy, mix) have no r_correct (r = -1) and are skipped.
// grid (ceil(N/ST_TILE), S), block 256.
// ------------------------------------------------------------------------------------------------
#define ST_TILE 1016            /* level-4 samples per tile: with the <= 8 extra level-0 samples of the two lerps one FIR round of 256 x 4 */
#define ST_THREADS 256
#define ST_TPB 8                /* consecutive tiles per workgroup (state, taps and the A/B rotator tables are set up once) */
struct StreamTileArgs {
    const uint8_t* raw; long raw_stride;
    const double* coef; int ntaps;
    cplx* dst; long dst_stream_stride;
};
__host__ __device__ inline size_t stream_tile_lds(int ntaps) {
    const size_t span = 1024 + 8 + ntaps + 24;                    // level-0 samples a tile can need + taps + alignment slack
    const size_t xs_n = span + span / 4 + 16;                     // padded complex input
    const size_t buf = 1024 + 16;
    // buf0 | region1 = max(xs, buf1) | coef | raw ushorts | rotator tables 2 x (2 + 34 + 32)
    return (buf + (xs_n > buf ? xs_n : buf)) * sizeof(cplx) + (size_t)((ntaps + 7) & ~3) * 8 + ((span + 7) & ~(size_t)7) * 2 +
           2 * (68 + 34) * sizeof(cplx);     // (taps padded to a multiple of four; per derotation S|A|B and the tile's S*A[q])
}

// T[0], T[1] = S of the current / next tile, T[2..35] = A[0..33], T[36..67] = B[0..31]
__device__ __forceinline__ cplx st_rot(const cplx* T, int sidx, int m) {     // exp(1i*(k0+m)*c), 0 <= m < 34*32
    const cplx sa = cmul(T[sidx], T[2 + (m >> 5)]);
    return cmul(sa, T[36 + (m & 31)]);
}

struct StRange { long lo0, hi0, lo2, hi2, lo3; int L4, cnt0, cnt2; };
template <int TILE = ST_TILE>
__device__ __forceinline__ StRange st_range(long tile, long nq, long n0, long n2, double f1, double f3) {
    StRange r;
    r.lo3 = tile * TILE;
    r.L4 = (int)(nq - r.lo3 < TILE ? nq - r.lo3 : TILE);
    const long hi3 = r.lo3 + r.L4 - 1;
    r.lo2 = (long)floor((double)r.lo3 * f3);
    r.hi2 = (long)floor((double)hi3 * f3) + 1;
    if (r.hi2 > n2 - 1) r.hi2 = n2 - 1;
    r.lo0 = (long)floor((double)r.lo2 * f1);
    r.hi0 = (long)floor((double)r.hi2 * f1) + 1;
    if (r.hi0 > n0 - 1) r.hi0 = n0 - 1;
    r.cnt0 = (int)(r.hi0 - r.lo0 + 1);
    r.cnt2 = (int)(r.hi2 - r.lo2 + 1);
    return r;
}

// NTAPS > 0: the tap count as a compile-time constant (47, the drivers' fir1(46)); 0: from the arguments
template <int NTAPS>
__global__ void __launch_bounds__(ST_THREADS) k_stream_tile(const StreamState* __restrict__ sts, StreamTileArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = blockIdx.y, tid = threadIdx.x;
    const StreamState* st = sts + s;
    // ---- chain (all fields fetched up front) ----
    const long n0 = st->n0;
    const double mr = st->mean_re, mi = st->mean_im;
    int ty[NLEVELS]; double pa[NLEVELS]; long ln[NLEVELS];
    ty[0] = OP_NONE; pa[0] = 0.0; ln[0] = n0;
#pragma unroll
    for (int j = 1; j < NLEVELS; ++j) { ty[j] = st->op[j].type; pa[j] = st->op[j].param; ln[j] = st->op[j].n; }
    if (ty[1] != OP_LERP || ty[2] != OP_MIX || (ty[3] != OP_LERP && ty[3] != OP_COPY) || ty[4] != OP_MIX) return;
    const double f1 = pa[1], c2 = pa[2], f3 = ty[3] == OP_COPY ? 1.0 : pa[3], c4 = pa[4];
    const long nq = ln[4];
    const long ntile = (nq + ST_TILE - 1) / ST_TILE;
    long tile = (long)blockIdx.x * ST_TPB;
    if (tile >= ntile) return;
    const long tile_end = tile + ST_TPB < ntile ? tile + ST_TPB : ntile;
    // ---- LDS carve ----
    const int ntp = NTAPS > 0 ? NTAPS : a.ntaps;
    const size_t span_max = 1024 + 8 + ntp + 24;
    const size_t xs_n = span_max + span_max / 4 + 16, bufn = 1024 + 16;
    cplx* buf0 = (cplx*)smem;                                   // level 0 (FIR output)
    cplx* buf1 = buf0 + bufn;                                   // level 2 (after lerp + derotation); aliases xs
    cplx* xs = buf1;
    double* c_s = (double*)(buf1 + (xs_n > bufn ? xs_n : bufn));   // rev[j] = coef[ntp-1-j] (oldest tap first), zero-padded to ntp4
    const int ntp4 = (ntp + 3) & ~3;
    unsigned short* r_s = (unsigned short*)(c_s + ((ntp + 7) & ~3));
    cplx* T2 = (cplx*)(r_s + ((span_max + 7) & ~(size_t)7));
    cplx* T4 = T2 + 68;
    cplx* SA2 = T4 + 68;                                        // S*A[q] of the current tile, q < 34, per derotation
    cplx* SA4 = SA2 + 34;
    const unsigned short* base = (const unsigned short*)(a.raw + (size_t)s * a.raw_stride);
    const long ao = (long)(((uintptr_t)base >> 1) & 7);         // samples past a 16-byte boundary at g = 0
    // ---- once per workgroup: taps, the stream's A/B rotator tables, the first tile's S and raw bytes ----
    for (int i = tid; i < ntp4; i += ST_THREADS) c_s[i] = i < ntp ? a.coef[ntp - 1 - i] : 0.0;
    StRange rg = st_range(tile, nq, n0, ln[2], f1, f3);
    if (tid >= 64 && tid < 64 + 2 * 66) {                       // A[j] = exp(1i*fl(32j*c)), B[m] = exp(1i*fl(m*c)) for both derotations
        const int i = (tid - 64) % 66, which = (tid - 64) / 66;
        const double c = which ? c4 : c2;
        double sn, cs;
        sincos_large(i < 34 ? (double)(32 * i) * c : (double)(i - 34) * c, &sn, &cs);
        (which ? T4 : T2)[2 + i] = make_double2(cs, sn);
    } else if (tid < 2) {                                       // S = exp(1i*fl(k0*c)) of the first tile
        double sn, cs;
        sincos_large(tid ? (double)rg.lo3 * c4 : (double)rg.lo2 * c2, &sn, &cs);
        (tid ? T4 : T2)[0] = make_double2(cs, sn);
    }
    long first = rg.lo0 - (ntp - 1);
    long first_al = stage_raw(r_s, base, n0, first, rg.cnt0 + ntp - 1, tid, ST_THREADS);
    __syncthreads();
    int sidx = 0;
    for (; tile < tile_end; ++tile) {
        const int cnt0 = rg.cnt0, cnt2 = rg.cnt2, L4 = rg.L4;
        const long lo0 = rg.lo0, lo2 = rg.lo2, lo3 = rg.lo3;
        {   // raw2iq.m:6-8 on the staged span
            const int span = cnt0 + ntp - 1, off = (int)(first - first_al);
            for (int i = tid; i < span + 12; i += ST_THREADS) {       // (+12: finite zeros behind the span for the padding taps)
                const long g = first + i;
                cplx v = make_double2(0.0, 0.0);
                if (i < span && g >= 0 && g < n0) {
                    const unsigned short q = r_s[off + i];
                    v = make_double2((double)(q & 0xFF) - mr, (double)(q >> 8) - mi);
                }
                xs[xs_pad(i)] = v;
            }
        }
        __syncthreads();                                        // xs complete; r_s is free again
        // ---- the next tile's raw bytes (one 16-byte chunk per lane) and S: issued now, they land under the FIR ----
        StRange nx = rg;
        uint4 pre = make_uint4(0u, 0u, 0u, 0u);
        long nfirst = 0, nfirst_al = 0;
        int nchunk = 0;
        const bool more = tile + 1 < tile_end;
        if (more) {
            nx = st_range(tile + 1, nq, n0, ln[2], f1, f3);
            nfirst = nx.lo0 - (ntp - 1);
            long m = (nfirst + ao) % 8;
            if (m < 0) m += 8;
            nfirst_al = nfirst - m;
            nchunk = (int)((nfirst + nx.cnt0 + ntp - 1 - nfirst_al + 7) >> 3);
            if (tid < nchunk) {
                const long g0 = nfirst_al + 8L * tid;
                if (g0 >= 0 && g0 + 8 <= n0) pre = *(const uint4*)(base + g0);
                else {
                    unsigned short q[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) q[i] = (g0 + i >= 0 && g0 + i < n0) ? base[g0 + i] : (unsigned short)0;
                    pre.x = q[0] | ((unsigned)q[1] << 16); pre.y = q[2] | ((unsigned)q[3] << 16);
                    pre.z = q[4] | ((unsigned)q[5] << 16); pre.w = q[6] | ((unsigned)q[7] << 16);
                }
            }
        }
        // ---- pass 1: filter(coef,1,.) -> buf0, four consecutive outputs per lane, oldest tap first (gather_core's loop).
        // (Two outputs per lane with twice the threads ran 40 % slower: the LDS reads per FMA double.) ----
        // The taps sit in LDS reversed and zero-padded to a multiple of four: no remainder loop (fma(0, x, acc) = acc for the
        // finite samples staged behind the span, so the sums are those of the ntp-term loop bit for bit), and two trips per
        // iteration let the sliding window rotate by renaming instead of eight register moves per trip.
        for (int i0 = 4 * tid; i0 < cnt0; i0 += 4 * ST_THREADS) {
            double ar0 = 0.0, ai0 = 0.0, ar1 = 0.0, ai1 = 0.0, ar2 = 0.0, ai2 = 0.0, ar3 = 0.0, ai3 = 0.0;
            cplx w0 = xs[xs_pad(i0)], w1 = xs[xs_pad(i0 + 1)], w2 = xs[xs_pad(i0 + 2)], w3 = xs[xs_pad(i0 + 3)];
#define ST_FIR_TAP(C, A, B, D, E)                                   \
            ar0 = fma(C, A.x, ar0); ai0 = fma(C, A.y, ai0);         \
            ar1 = fma(C, B.x, ar1); ai1 = fma(C, B.y, ai1);         \
            ar2 = fma(C, D.x, ar2); ai2 = fma(C, D.y, ai2);         \
            ar3 = fma(C, E.x, ar3); ai3 = fma(C, E.y, ai3);
#pragma unroll 2
            for (int t = 0; t < ntp4; t += 4) {
                const cplx* nxs = xs + xs_pad(i0 + t + 4);
                const cplx n0s = nxs[0], n1s = nxs[1], n2s = nxs[2], n3s = nxs[3];
                const double c0 = c_s[t], c1 = c_s[t + 1], c2_ = c_s[t + 2], c3 = c_s[t + 3];
                ST_FIR_TAP(c0, w0, w1, w2, w3)
                ST_FIR_TAP(c1, w1, w2, w3, n0s)
                ST_FIR_TAP(c2_, w2, w3, n0s, n1s)
                ST_FIR_TAP(c3, w3, n0s, n1s, n2s)
                w0 = n0s; w1 = n1s; w2 = n2s; w3 = n3s;
            }
#undef ST_FIR_TAP
            buf0[i0] = make_double2(ar0, ai0);
            if (i0 + 1 < cnt0) buf0[i0 + 1] = make_double2(ar1, ai1);
            if (i0 + 2 < cnt0) buf0[i0 + 2] = make_double2(ar2, ai2);
            if (i0 + 3 < cnt0) buf0[i0 + 3] = make_double2(ar3, ai3);
        }
        if (more) {                                             // the prefetched bytes go to LDS; the next tile's S to the other slot
            if (tid < nchunk) *(uint4*)(r_s + 8 * tid) = pre;
            for (int cch = tid + ST_THREADS; cch < nchunk; cch += ST_THREADS) {   // (spans beyond 2048 samples: long filters)
                const long g0 = nfirst_al + 8L * cch;
                unsigned short q[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) q[i] = (g0 + i >= 0 && g0 + i < n0) ? base[g0 + i] : (unsigned short)0;
                uint4 v;
                v.x = q[0] | ((unsigned)q[1] << 16); v.y = q[2] | ((unsigned)q[3] << 16);
                v.z = q[4] | ((unsigned)q[5] << 16); v.w = q[6] | ((unsigned)q[7] << 16);
                *(uint4*)(r_s + 8 * cch) = v;
            }
            if (tid >= ST_THREADS - 2) {
                double sn, cs;
                sincos_large(tid == ST_THREADS - 1 ? (double)nx.lo3 * c4 : (double)nx.lo2 * c2, &sn, &cs);
                (tid == ST_THREADS - 1 ? T4 : T2)[sidx ^ 1] = make_double2(cs, sn);
            }
        }
        // S * A[q] of this tile for both derotations (34 + 34 products), so that a sample's rotator is one product, SA[m>>5]*B[m&31]:
        // the same three factors as st_rot(), associated the same way ((S*A)*B), bit-identical
        if (tid >= 128 && tid < 128 + 68) {
            const int q = (tid - 128) % 34, which = (tid - 128) / 34;
            const cplx* T = which ? T4 : T2;
            (which ? SA4 : SA2)[q] = cmul(T[sidx], T[2 + q]);
        }
        __syncthreads();                                        // (xs is dead: buf1 may be written)
        // ---- pass 2: level 1 = interp1 (FCCH_fine_correction.m:123-125), level 2 = .* exp(1i*k*c2) (:165) -> buf1 ----
        {
            const double dlo2 = (double)lo2, dlo0 = (double)lo0;    // (indices < 2^53: the double sums and differences below are exact)
            const int last0 = cnt0 - 1;
            for (int i = tid; i < cnt2; i += ST_THREADS) {
                const double xq = (dlo2 + (double)i) * f1;      // interp_seq = (0:max_len-1)'.*(1+e)
                const double j0f = floor(xq);
                const int j0 = (int)(j0f - dlo0);
                const int j1 = j0 + 1 > last0 ? last0 : j0 + 1; // beyond the last sample the weight is 0
                const double t = xq - j0f;
                const cplx v0 = buf0[j0], v1 = buf0[j1];
                const cplx v = make_double2(v0.x + t * (v1.x - v0.x), v0.y + t * (v1.y - v0.y));
                buf1[i] = cmul(v, cmul(SA2[i >> 5], T2[36 + (i & 31)]));
            }
        }
        __syncthreads();
        // ---- pass 3: level 3 = interp1 (SCH_corr_rate_correction.m:126-127), level 4 = .* exp(1i*k*c4)
        // (carrier_correct_post_SCH.m:83) -> global, one coalesced 16-byte store per lane ----
        cplx* dst = a.dst + (size_t)s * a.dst_stream_stride + lo3;
        {
            const double dlo3 = (double)lo3, dlo2 = (double)lo2;
            const int last2 = cnt2 - 1;
            for (int i = tid; i < L4; i += ST_THREADS) {
                const double xq = (dlo3 + (double)i) * f3;
                const double j0f = floor(xq);
                const int j0 = (int)(j0f - dlo2);
                const int j1 = j0 + 1 > last2 ? last2 : j0 + 1;
                const double t = xq - j0f;
                const cplx v0 = buf1[j0], v1 = buf1[j1];
                const cplx v = make_double2(v0.x + t * (v1.x - v0.x), v0.y + t * (v1.y - v0.y));
                st_stream16(dst + i, cmul(v, cmul(SA4[i >> 5], T4[36 + (i & 31)])));
            }
        }
        __syncthreads();                                        // buf1 (= xs) and the tables' S slot may be rewritten
        rg = nx; first = nfirst; first_al = nfirst_al; sidx ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------
// k_stream_tile_s47: k_stream_tile for the drivers' filter -- 47 exactly symmetric taps (fir1(46), gsm_sync_demod.m:34) -- with
// the two things the general kernel spends most of its LDS bandwidth on removed (round 4; the general kernel's FIR loop issued 24
// LDS reads per output sample, a third of them tap broadcasts, and the counters put the LDS pipe ahead of the VALU there):
//   * the 24 distinct taps live in registers (uniform loads, once per workgroup), and the FIR of a lane's four outputs is one
//     unrolled pass over its 50 input samples, each read ONCE (12.5 LDS reads per output sample) and used at once by the up to
//     four outputs it belongs to -- every output still takes its taps oldest first, so the sums are those of k_stream_tile
//     (and of gather_core) bit for bit;
//   * raw bytes go from the prefetched 16-byte chunk in registers straight to the padded complex copy (no raw-byte staging
//     area, no ds_read_u16 per sample, no per-sample 64-bit index tests inside the stream).
// Passes 2 and 3 (lerp x rotator) are k_stream_tile's.  grid (ceil(N / (ST_TILE * ST_TPB)), S), block 256.
// ------------------------------------------------------------------------------------------------
// TILE: level-4 samples per tile; TILE + 8 level-0 samples are filtered per tile (the two lerps stretch a tile by < 8).  1016
// (one FIR round of 256 x 4: three workgroups per CU, 0.592 against 0.566 ms) or the largest that still leaves four per CU: 952
// (240 x 4) while the per-tile S*A tables took another kilobyte, 1000 (252 x 4, 40 512 B with the static part) since they are
// gone -- stream mode 0.593 -> 0.589 ms (1008: the same; 744 with five workgroups per CU: slower)
#ifndef ST47_TILE
#define ST47_TILE 1000
#endif
#ifndef ST47_MAXW
#define ST47_MAXW 4           /* waves per SIMD the register allocation leaves room for */
#endif
__host__ __device__ constexpr size_t st47_xs_n(int tile) { return (size_t)(tile + 8 + 47 + 16) + (size_t)(tile + 8 + 47 + 16) / 4 + 2; }
__host__ __device__ constexpr size_t st47_buf_n(int tile) { return (size_t)tile + 8 + 4; }
__host__ __device__ inline size_t stream_tile_s47_lds(int tile = ST47_TILE) {
    const size_t xs_n = st47_xs_n(tile), buf = st47_buf_n(tile);
    return (buf + (xs_n > buf ? xs_n : buf)) * sizeof(cplx) + 2 * 68 * sizeof(cplx);
}
// four consecutive outputs i0 .. i0+3 (i0 a multiple of 4) of the 47-tap symmetric filter from the padded copy xq; c[t] = coef[t], t < 24
template <int S0, int G>
__device__ __forceinline__ void fir4_sym47_load(const cplx* __restrict__ x0, cplx (&v)[G]) {
#pragma unroll
    for (int u = 0; u < G; ++u) { constexpr int dummy = 0; (void)dummy; const int sn = S0 + u; v[u] = x0[sn + (sn >> 2)]; }
}
template <int S0, int G>
__device__ __forceinline__ void fir4_sym47_mac(const cplx (&v)[G], const double (&c)[24], double (&ar)[4], double (&ai)[4]) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
        const int s = S0 + u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = s - j;                          // output i0+j takes sample i0+s with tap rev[t] = coef[46-t] = coef[t]
            if (t >= 0 && t < 47) {
                const double cc = c[t < 24 ? t : 46 - t];
                ar[j] = fma(cc, v[u].x, ar[j]);
                ai[j] = fma(cc, v[u].y, ai[j]);
            }
        }
    }
}
__device__ __forceinline__ void fir4_sym47(const cplx* __restrict__ xq, const double (&c)[24], int i0, cplx* y) {
    double ar[4] = {0.0, 0.0, 0.0, 0.0}, ai[4] = {0.0, 0.0, 0.0, 0.0};
    const cplx* x0 = xq + xs_pad(i0);                    // i0 % 4 == 0: xs_pad(i0 + s) = xs_pad(i0) + s + (s >> 2)
    // ten groups of five samples in two register sets: the next group's LDS reads are issued ahead of this group's FMAs, and a
    // scheduling barrier per group keeps the compiler from hoisting all fifty reads to the top (255 registers, one wave per SIMD)
    constexpr int G = 5;
    cplx A[G], B[G];
    fir4_sym47_load<0, G>(x0, A);
#define FIR47_PAIR(g)                                                   \
    fir4_sym47_load<G * ((g) + 1), G>(x0, B);                           \
    fir4_sym47_mac<G * (g), G>(A, c, ar, ai);                           \
    __builtin_amdgcn_sched_barrier(0);                                  \
    if ((g) + 2 < 50 / G) fir4_sym47_load<G * ((g) + 2), G>(x0, A);     \
    fir4_sym47_mac<G * ((g) + 1), G>(B, c, ar, ai);                     \
    __builtin_amdgcn_sched_barrier(0);
    FIR47_PAIR(0) FIR47_PAIR(2) FIR47_PAIR(4) FIR47_PAIR(6) FIR47_PAIR(8)
#undef FIR47_PAIR
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = make_double2(ar[j], ai[j]);
}

// what a tile needs to know about itself: worked out ONCE per workgroup for its ST_TPB tiles, one tile per lane, and read back
// from LDS -- the ranges are floors of double products and 64-bit index arithmetic, which every lane of every wave would
// otherwise repeat for every tile (no scalar floating-point unit: ~230 vector instructions per wave and tile, a quarter of
// this kernel's instructions), and the two accurate sincos per tile cost the wave that holds their lanes another ~200
struct StTile { long lo0, lo2, lo3, first, first_al; int cnt0, cnt2, L4, nchunk; };
template <int TILE>
__global__ void __launch_bounds__(ST_THREADS) __attribute__((amdgpu_waves_per_eu(4, ST47_MAXW))) k_stream_tile_s47(const StreamState* __restrict__ sts, StreamTileArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ StTile tl[ST_TPB];
    __shared__ cplx S2[ST_TPB], S4[ST_TPB];                     // exp(1i*fl(k0*c)) per tile and derotation
    constexpr int ntp = 47;
    const int s = blockIdx.y, tid = threadIdx.x;
    const StreamState* st = sts + s;
    const long n0 = st->n0;
    const double mr = st->mean_re, mi = st->mean_im;
    int ty[NLEVELS]; double pa[NLEVELS]; long ln[NLEVELS];
    ty[0] = OP_NONE; pa[0] = 0.0; ln[0] = n0;
#pragma unroll
    for (int j = 1; j < NLEVELS; ++j) { ty[j] = st->op[j].type; pa[j] = st->op[j].param; ln[j] = st->op[j].n; }
    if (ty[1] != OP_LERP || ty[2] != OP_MIX || (ty[3] != OP_LERP && ty[3] != OP_COPY) || ty[4] != OP_MIX) return;
    const double f1 = pa[1], c2 = pa[2], f3 = ty[3] == OP_COPY ? 1.0 : pa[3], c4 = pa[4];
    const long nq = ln[4];
    const long ntile = (nq + TILE - 1) / TILE;
    const long tile0 = (long)blockIdx.x * ST_TPB;
    if (tile0 >= ntile) return;
    const int ntl = (int)(tile0 + ST_TPB < ntile ? ST_TPB : ntile - tile0);
    // ---- LDS carve: buf0 | region1 = xs / buf1 | rotator tables ----
    constexpr size_t xs_n = st47_xs_n(TILE), bufn = st47_buf_n(TILE);
    cplx* buf0 = (cplx*)smem;
    cplx* buf1 = buf0 + bufn;
    cplx* xs = buf1;
    cplx* T2 = buf1 + (xs_n > bufn ? xs_n : bufn);              // T[2..35] = A[0..33], T[36..67] = B[0..31] (k_stream_tile's layout)
    cplx* T4 = T2 + 68;
    const unsigned short* base = (const unsigned short*)(a.raw + (size_t)s * a.raw_stride);
    double cf[24];                                              // the taps (uniform addresses: scalar loads)
#pragma unroll
    for (int k = 0; k < 24; ++k) cf[k] = a.coef[k];
    // ---- once per workgroup: the tiles' ranges (lanes 0..), their S rotators (lanes 32..), the stream's A/B tables (lanes 64..) ----
    if (tid < ST_TPB) {
        if (tid < ntl) {
            const StRange r = st_range<TILE>(tile0 + tid, nq, n0, ln[2], f1, f3);
            const long ao = (long)(((uintptr_t)base >> 1) & 7);  // samples past a 16-byte boundary at g = 0
            StTile t;
            t.lo0 = r.lo0; t.lo2 = r.lo2; t.lo3 = r.lo3; t.cnt0 = r.cnt0; t.cnt2 = r.cnt2; t.L4 = r.L4;
            t.first = r.lo0 - (ntp - 1);
            long m = (t.first + ao) % 8;
            if (m < 0) m += 8;
            t.first_al = t.first - m;
            t.nchunk = (int)((t.first + r.cnt0 + ntp - 1 - t.first_al + 7) >> 3);
            tl[tid] = t;
        }
    } else if (tid >= 32 && tid < 32 + 2 * ST_TPB) {
        const int t = (tid - 32) % ST_TPB, which = (tid - 32) / ST_TPB;
        if (t < ntl) {
            const StRange r = st_range<TILE>(tile0 + t, nq, n0, ln[2], f1, f3);
            double sn, cs;
            sincos_large(which ? (double)r.lo3 * c4 : (double)r.lo2 * c2, &sn, &cs);
            (which ? S4 : S2)[t] = make_double2(cs, sn);
        }
    } else if (tid >= 64 && tid < 64 + 2 * 66) {
        const int i = (tid - 64) % 66, which = (tid - 64) / 66;
        const double c = which ? c4 : c2;
        double sn, cs;
        sincos_large(i < 34 ? (double)(32 * i) * c : (double)(i - 34) * c, &sn, &cs);
        (which ? T4 : T2)[2 + i] = make_double2(cs, sn);
    }
    __syncthreads();
    // B[m] of a lane's samples: i = tid + 256 k, so i & 31 = tid & 31 for every k -- one read per workgroup instead of one per sample
    const cplx b2 = T2[36 + (tid & 31)], b4 = T4[36 + (tid & 31)];
    // ... and so is A[i >> 5] for the lane's k-th sample of a pass, (tid >> 5) + 8 k: the products A*B of the lane's (up to) four
    // samples per pass live in registers for the whole workgroup, and a sample's rotator is S(tile) * (A*B) -- one product per
    // sample, no per-tile S*A table in LDS and no read of it per sample (stream mode 0.608 -> 0.594 ms).  (S*(A*B) instead of
    // (S*A)*B: the same three factors, associated the other way -- a relative difference of 1e-16 against the general kernel's
    // rotator.)  128 registers: still four waves per SIMD.
    static_assert(TILE + 8 <= 4 * ST_THREADS, "a pass is at most four samples per lane");
    cplx ab2[4], ab4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ab2[k] = cmul(T2[2 + (tid >> 5) + 8 * k], b2);
        ab4[k] = cmul(T4[2 + (tid >> 5) + 8 * k], b4);
    }
    // the first tile's raw bytes: chunk `tid` (8 samples, 16-byte aligned) of [first_al, first + span)
    uint4 pre = make_uint4(0u, 0u, 0u, 0u);
    if (tid < tl[0].nchunk) pre = ffast_chunk(base, tl[0].first_al + 8L * tid, n0);
    for (int t = 0; t < ntl; ++t) {
        const StTile c = tl[t];
        const int cnt0 = c.cnt0, cnt2 = c.cnt2, L4 = c.L4;
        const long lo0 = c.lo0, lo2 = c.lo2, lo3 = c.lo3, first = c.first;
        {   // raw2iq.m:6-8 from the registers: staged sample i = 8*tid + u - (first - first_al); zeros outside the span / the stream
            const int span = cnt0 + ntp - 1, off = (int)(first - c.first_al);
            const int nconv = (off + span + 8 + 7) >> 3;        // (finite zeros behind the span: the last lanes' discarded outputs read them)
            if (tid < nconv) {
                const int i_base = 8 * tid - off;
                const long g_base = first + i_base;
                const unsigned wv[4] = {pre.x, pre.y, pre.z, pre.w};
                if (i_base >= 0 && i_base + 8 <= span && g_base >= 0 && g_base + 8 <= n0) {
                    // xs_pad(i_base + u) = xs_pad(i_base) + u + ((r + u) >> 2) with r = i_base & 3 = (-off) & 3, the same in every
                    // lane: the eight slots are one lane address plus scalar offsets (three vector instructions per slot before)
                    const int r = __builtin_amdgcn_readfirstlane((-off) & 3);
                    cplx* xp = xs + xs_pad(i_base);
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
                        if (i >= 0) xs[xs_pad(i)] = v;
                    }
                }
            }
        }
        const cplx s2t = S2[t], s4t = S4[t];                    // the tile's S rotators (uniform reads)
        __syncthreads();                                        // xs complete
        // ---- the next tile's raw chunk: requested now, it lands under the FIR ----
        pre = make_uint4(0u, 0u, 0u, 0u);
        if (t + 1 < ntl && tid < tl[t + 1].nchunk) pre = ffast_chunk(base, tl[t + 1].first_al + 8L * tid, n0);
        // ---- pass 1: filter(coef,1,.) -> buf0 ----
        // (issue priority: the filter pass is a long run of independent FMAs, the other phases are short dependent chains behind LDS
        // round trips -- waves in those phases issue first, the filter passes of the CU's other workgroups fill the rest:
        // stream mode 0.585 -> 0.578 ms; the other way round 0.594)
        __builtin_amdgcn_s_setprio(0);
#pragma unroll 1
        for (int i0 = 4 * tid; i0 < cnt0; i0 += 4 * ST_THREADS) {
            cplx y[4];
            fir4_sym47(xs, cf, i0, y);
            // (all four stored: buf0 has room up to the next multiple of four, and the samples behind the span are staged as zeros.
            // With the last three stores conditional the compiler sinks their outputs' FMAs under the conditions and keeps all
            // fifty samples alive across them: 200 registers and a scratch frame)
            buf0[i0] = y[0]; buf0[i0 + 1] = y[1]; buf0[i0 + 2] = y[2]; buf0[i0 + 3] = y[3];
        }
        __builtin_amdgcn_s_setprio(2);
        __syncthreads();                                        // (xs is dead: buf1 may be written)
        // ---- pass 2: level 1 = interp1 (FCCH_fine_correction.m:123-125), level 2 = .* exp(1i*k*c2) (:165) -> buf1 ----
        {
            const double dlo0 = (double)lo0, p2 = (double)lo2 + (double)tid;   // (integers below 2^53: every sum here is exact)
            const int last0 = cnt0 - 1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {                       // (cnt2 <= 4 * ST_THREADS: static register indices for ab2)
                const int i = tid + k * ST_THREADS;
                if (i < cnt2) {
                    const double xq = (p2 + (double)(k * ST_THREADS)) * f1;
                    const double j0f = floor(xq);
                    const int j0 = (int)(j0f - dlo0);
                    const int j1 = j0 + 1 > last0 ? last0 : j0 + 1;
                    const double tt = xq - j0f;
                    const cplx v0 = buf0[j0], v1 = buf0[j1];
                    const cplx v = make_double2(v0.x + tt * (v1.x - v0.x), v0.y + tt * (v1.y - v0.y));
                    buf1[i] = cmul(v, cmul(s2t, ab2[k]));
                }
            }
        }
        __syncthreads();
        // ---- pass 3: level 3 = interp1 (SCH_corr_rate_correction.m:126-127), level 4 = .* exp(1i*k*c4) (carrier_correct_post_SCH.m:83) ----
        cplx* dst = a.dst + (size_t)s * a.dst_stream_stride + lo3;
        {
            const double dlo2 = (double)lo2, p3 = (double)lo3 + (double)tid;
            const int last2 = cnt2 - 1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = tid + k * ST_THREADS;
                if (i < L4) {
                    const double xq = (p3 + (double)(k * ST_THREADS)) * f3;
                    const double j0f = floor(xq);
                    const int j0 = (int)(j0f - dlo2);
                    const int j1 = j0 + 1 > last2 ? last2 : j0 + 1;
                    const double tt = xq - j0f;
                    const cplx v0 = buf1[j0], v1 = buf1[j1];
                    const cplx v = make_double2(v0.x + tt * (v1.x - v0.x), v0.y + tt * (v1.y - v0.y));
                    st_stream16(dst + i, cmul(v, cmul(s4t, ab4[k])));
                }
            }
        }
        __syncthreads();                                        // buf1 (= xs) may be rewritten
    }
}

__global__ void __launch_bounds__(256) k_gather(const StreamState* __restrict__ sts, GatherArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    DEV_STAMP(KID_GATHER, blockIdx.y * gridDim.x + blockIdx.x, 0);
    (void)gather_core<256>(sts, a, smem, blockIdx.x, blockIdx.y, false);
    DEV_STAMP(KID_GATHER, blockIdx.y * gridDim.x + blockIdx.x, 1);
}
