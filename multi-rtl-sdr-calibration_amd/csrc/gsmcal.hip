// gsmcal.hip -- host side of libgsmcal.so: context, workspace, launch sequences, C ABI (include/gsmcal.h).
//
// The calibration chain is enqueued as a fixed sequence of kernels on one HIP stream; all
// data-dependent control lives in StreamState on the device (see state.h).  The same building
// blocks serve the per-function MATLAB-signature entry points (level 0 = a complex array handed in)
// and the batched hot path (level 0 = FIR of the raw bytes, evaluated lazily window by window).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <sys/stat.h>
#include <time.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include "../../include/gsmcal.h"
#include "builtin_taps.h"
#include "kernels_detect.h"
#include "kernels_estim.h"
#include "kernels_frontend.h"
#include "kernels_demod.h"
#include "state.h"

#define GSMCAL_VERSION "gsmcal-mi355x 0.1 (gfx950)"
#define TILE 1024

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct ProfRec {
    int name_id;
    hipEvent_t e0, e1;
};

}  // namespace

// A lane = one HIP stream + the per-stream-group scratch of the chain.  A batch is split over several
// lanes so that one group's latency-bound stages (coarse scan, decisions, small FFTs) run underneath
// another group's compute-bound fine search.  Lane 0 runs on the context's own stream.
#define MAX_LANES 32
struct Lane {
    hipStream_t stream = nullptr;
    DevBuf state, dec, win, peaks, snrbuf, x0, chunkrec, openlist, cert, partial, tailctr, postctr, xch, xepoch;
    int npartial = 0;           // front-kernel blocks per stream of the last front_fused()
    hipEvent_t done = nullptr;
    hipEvent_t front_done = nullptr;   // scanner pipeline: this stage's front kernel has finished
    int lo = 0, n = 0;          // streams [lo, lo+n) of the last batch
    long snr_stride = 0, snr_nmove = 0; // SNR table of the last coarse(): entries per stream, and how many of them are the moving search's
    int win_l0_len = 0, win_l0_H = 0;  // > 0: `win` holds the fine search's level-0 windows (this length each, H per stream) of the call in progress
};

struct gsmcal_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    Lane lanes[MAX_LANES];
    Lane* cur = nullptr;        // lane the helpers below enqueue on
    int n_lanes_cfg = 4;        // GSMCAL_LANES: upper bound; a lane gets at least 64 streams (measured: 128 streams 325 / 343 Gsample/s with
                                // 1 / 2 lanes; 256: 377 / 397 / 421 with 1 / 2 / 4; 512: 421 / 424 / 449-457; 8 or 16 lanes no better)
    int n_lanes_used = 1;
    const double* cf_lane = nullptr;   // carrier_freq of the current lane's first stream (batch path)
    hipEvent_t fork = nullptr;
    // hipGraph replay of a repeated batch call (same pointers, sizes and parameters as the previous call)
    struct GraphSlot {
        std::vector<uintptr_t> key;
        unsigned long epoch = 0;
        int seen = 0;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        unsigned long used = 0;     // stamp of the last call that took this slot (least recently used one is recycled)
    };
    // callers that alternate between buffers (two output tables; the ingest ring's device slots) keep one graph per
    // combination: GRAPH_SLOTS keys per entry point, least recently used one recycled
    static constexpr int GRAPH_SLOTS = 4;
    GraphSlot g_calib[GRAPH_SLOTS], g_scan[GRAPH_SLOTS];
    unsigned long g_stamp = 0;
    unsigned long ws_epoch = 0;     // bumped whenever a workspace buffer is (re)allocated or a parameter upload happens
    bool use_graph = true;          // GSMCAL_GRAPH=0 disables
    bool graph_always = false;      // GSMCAL_GRAPH=2: also single-stream plans (default: only plans that fork onto internal streams)
    bool prescreen = true;          // GSMCAL_PRESCREEN=0: run the fp64 fine search on every bin
    int n_cu = 256;                 // compute units of the device (persistent-grid sizing)
    int snr_inline_min = 1;         // GSMCAL_SNR_INLINE_MIN: streams per lane from which k_coarse_scan computes the window SNRs itself when the full table is not built (0: never)
    int snr_inline_keep = 0;        // GSMCAL_SNR_INLINE_KEEP=1: ... and still writes the table out (gsmcal_last_batch_snr)
    int front_nt = -1;              // GSMCAL_FRONT_NT: non-temporal raw loads in k_front_fast (-1: by the size of the call, see front_fused())
    size_t call_raw_bytes = 0;      // raw bytes of the batch call in progress
    int snr_inline_pipe = 1;        // GSMCAL_SNR_INLINE_PIPE=0: the two-kernel detector in the scanner's pipeline stages (the inline form is used there
                                    // only while a stage's workgroups are all resident at once: 3 per CU)
    int scan_split = 88;            // GSMCAL_SCAN_SPLIT: percent of a pipeline stage's captures in the first of its two front-kernel launches (0: one launch;
                                    // 12 800 captures: 0 / 70 / 80 / 88 / 94 -> 3.64 / 3.58 / 3.525 / 3.515 / 3.57 ms)
    int scan_stages = 0;            // GSMCAL_SCAN_STAGES: pipeline stages of a big scanner batch (0: by batch size)
    int lane_min = 64;              // GSMCAL_LANE_MIN: fewest streams a lane is worth forking for
    bool certify = true;            // GSMCAL_CERT=0: no Parseval certificate, every chunk of every window is swept
    bool reuse_l0 = true;           // GSMCAL_REUSE_L0=0: every per-burst gather filters its raw bytes again
    bool snr_full = true;           // GSMCAL_SNR_FULL=0: the hop walk of FCCH_coarse_position computes its own 16-point spectra
    double snr_screen_db = 5.0;     // GSMCAL_SNR_SCREEN_DB: level below which k_coarse_snr proves windows instead of computing them
                                    // (typical thresholds hit_avg_snr + th sit at 6.5 .. 7.5 dB; ~95 % of the windows are below 5)
    bool fuse_fine_gather = true;   // GSMCAL_FUSE_GATHER=0: a k_gather launch writes the fine windows, k_fine_cert reads them back
    bool post_repl = true;          // GSMCAL_POST_REPL=0: k_post_chain (last arriver decides, state through memory) instead of k_post_chain_r
    int lane_stagger = -1;          // GSMCAL_LANE_STAGGER=0/1: calibration lanes start together / one front kernel apart; -1 (default): apart from 256 streams per lane on
    bool fuse_post = true;          // GSMCAL_FUSE_POST=0: k_fine_verify, k_burst_tone<1>, k_window_sch, k_burst_tone<0> as four launches
    bool fcert_s47 = true;          // GSMCAL_FCERT_S47=0: k_fine_cert builds its windows with the LDS-tap FIR loop also for the 47-tap symmetric filter
    bool stream_s47 = true;         // GSMCAL_STREAM_S47=0: the general k_stream_tile also for the 47-tap symmetric filter
    bool front_generic = false;     // GSMCAL_FRONT_GENERIC=1: the any-geometry front kernel also for the 47/31-tap production geometry
    bool capturing = false;
    struct OccEntry { int variant; size_t lds; int blocks; };
    std::vector<OccEntry> occ_cache;  // post_chain_blocks_per_cu()
    int post_slots_cap = 0;           // GSMCAL_POST_SLOTS: upper bound on the fused tail's workgroups per CU (0: the occupancy calculator's figure)
    gsmcal_params params;           // thresholds (defaults = the reference's literals)
    unsigned long params_epoch = 0; // bumped by gsmcal_set_params: captured graphs carry the old values
    // shared workspace
    DevBuf coef, ts, cf, table, snrhit, arr_in, arr_out, posinfo, rlen, misc, tw, csum_head, tw_sch;
    int tw_sch_n = 0;                        // length the SCH-demodulator twiddle table was built for
    std::vector<double> h_head;              // partial tap sums uploaded to csum_head (see coarse())
    unsigned long coef_epoch = 0, head_epoch = ~0ul;   // coef_epoch: bumped whenever h_coef changes
    int tw_n = 0;                            // length the twiddle table was built for
    std::vector<double> h_coef, h_ts, h_cf;   // host copies: upload only when changed
    int last_S = 0;
    // gsmcal_allgather_table_async: the collective on a side stream, behind / ahead of events on the context's stream
    static constexpr int AG_SLOTS = 4;
    hipStream_t ag_stream = nullptr;
    hipEvent_t ag_ready[AG_SLOTS] = {nullptr, nullptr, nullptr, nullptr}, ag_done[AG_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    bool ag_posted[AG_SLOTS] = {false, false, false, false};
    // profiling
    bool prof = false;
    std::string prof_filter;
    std::vector<std::string> prof_names;
    std::vector<double> prof_ms;
    std::vector<long> prof_n;
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> ev_pool;
};

namespace {

#define HIPCHK(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) {                                                            \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                \
            return GSMCAL_E_HIP;                                                            \
        }                                                                                   \
    } while (0)

#define RET_IF(x)             \
    do {                      \
        int r__ = (x);        \
        if (r__ < 0) return r__; \
    } while (0)

int ensure(gsmcal_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) {
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    HIPCHK(c, hipMalloc(&b.p, want));
    b.cap = want;
    ++c->ws_epoch;
    return 0;
}

int prof_id(gsmcal_ctx* c, const char* name) {
    for (size_t i = 0; i < c->prof_names.size(); ++i)
        if (c->prof_names[i] == name) return (int)i;
    c->prof_names.push_back(name);
    c->prof_ms.push_back(0.0);
    c->prof_n.push_back(0);
    return (int)c->prof_names.size() - 1;
}

hipEvent_t get_event(gsmcal_ctx* c) {
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    // device-scope release: a default event makes the queue flush to system scope at every record, which
    // stretches a 0.34 ms step by ~45 us with just four records in it
    if (hipEventCreateWithFlags(&e, hipEventReleaseToDevice) != hipSuccess) (void)hipEventCreate(&e);
    return e;
}

int prof_flush(gsmcal_ctx* c) {
    if (c->prof_pending.empty()) return 0;
    for (int i = 0; i < MAX_LANES; ++i)
        if (c->lanes[i].stream || i == 0) HIPCHK(c, hipStreamSynchronize(c->lanes[i].stream));
    for (auto& r : c->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            c->prof_ms[r.name_id] += ms;
            c->prof_n[r.name_id] += 1;
        }
        c->ev_pool.push_back(r.e0);
        c->ev_pool.push_back(r.e1);
    }
    c->prof_pending.clear();
    return 0;
}

// Profiled launches attach the start/stop events to the kernel's own dispatch packet (hipExtLaunchKernelGGL): the
// elapsed time is the kernel's execution time and no extra barrier packets enter the queue.  (Bracketing a launch
// with two hipEventRecord calls costs ~10 us of drained pipeline per record on this runtime.)
struct ProfScope {
    gsmcal_ctx* c;
    ProfRec r;
    bool on;
    ProfScope(gsmcal_ctx* ctx, const char* name) : c(ctx), on(ctx->prof) {
        if (on && !c->prof_filter.empty() && !strstr(name, c->prof_filter.c_str())) on = false;
        if (on) {
            r.name_id = prof_id(c, name);
            r.e0 = get_event(c);
            r.e1 = get_event(c);
        }
    }
    ~ProfScope() {
        if (on) {
            c->prof_pending.push_back(r);
            if (c->prof_pending.size() > 60000) (void)prof_flush(c);
        }
    }
};

// LAUNCH_GEOM: `kern_ref` when the call has the reference geometry (8x oversampling, 47 taps, ...: instantiations with
// compile-time loop bounds and divisors), `kern_any` otherwise
#define LAUNCH_GEOM(is_ref, c, kern_ref, kern_any, grid, block, shmem, ...)       \
    do {                                                                          \
        if (is_ref) LAUNCH(c, kern_ref, grid, block, shmem, __VA_ARGS__);         \
        else LAUNCH(c, kern_any, grid, block, shmem, __VA_ARGS__);                \
    } while (0)
#define LAUNCH(c, kern, grid, block, shmem, ...)                                  \
    do {                                                                          \
        ProfScope ps__(c, #kern);                                                 \
        if (ps__.on)                                                              \
            hipExtLaunchKernelGGL(kern, grid, block, shmem, (c)->cur->stream, ps__.r.e0, ps__.r.e1, 0, __VA_ARGS__); \
        else                                                                      \
            hipLaunchKernelGGL(kern, grid, block, shmem, (c)->cur->stream, __VA_ARGS__);   \
    } while (0)

#define CHECK_LAUNCH(c) HIPCHK(c, hipGetLastError())

int upload_cached(gsmcal_ctx* c, DevBuf& b, std::vector<double>& host, const double* src, size_t n) {
    if (host.size() == n && b.p && memcmp(host.data(), src, n * sizeof(double)) == 0) return 0;
    RET_IF(ensure(c, b, n * sizeof(double)));
    host.assign(src, src + n);
    ++c->ws_epoch;
    if (&host == &c->h_coef) ++c->coef_epoch;
    HIPCHK(c, hipMemcpyAsync(b.p, host.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return 0;
}

struct Geom {  // burst geometry for an oversampling ratio
    int ov, nfft, fine_wlen, fine_nshift, NB, sch_nshift;
    explicit Geom(int ov_) : ov(ov_) {
        nfft = 148 * ov;
        fine_nshift = 128 * ov + 1;          // FCCH_fine_correction.m:40-46: 2*max_offset*ov + 1
        fine_wlen = fine_nshift - 1 + nfft;
        NB = (nfft + 255) / 256;
        sch_nshift = 16 * ov - 5 * ov + 1;   // SCH_corr_rate_correction.m:45-48
    }
};

struct Source {
    int kind;
    const uint8_t* raw; long raw_stride;
    const cplx* arr; long arr_stride;
    const double* coef; int ntaps;
};

size_t gather_lds(int len, int level, int kind, int ntaps, bool to_lds = false) {
    return gather_carve(len, level, kind, ntaps, to_lds).total;
}

GatherArgs gather_args(const Source& src, int level, int len) {
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    a.src_kind = src.kind; a.level = level; a.len = len; a.tiles = 0; a.ntaps = src.ntaps;
    a.raw = src.raw; a.raw_stride = src.raw_stride; a.arr = src.arr; a.arr_stride = src.arr_stride;
    a.coef = src.coef;
    return a;
}

// LDS of a fused gather + estimator kernel: the gather carve followed by `scratch` bytes
size_t fused_lds(const Source& src, int level, int len, size_t scratch, bool compact_xs = false) {
    return (gather_carve(len, level, src.kind, src.ntaps, true, compact_xs).total + scratch + 15) & ~(size_t)15;
}

int launch_gather(gsmcal_ctx* c, int S, const Source& src, int level, int len, bool tiles, int nwin_grid,
                  cplx* dst, long dst_stream_stride, long dst_win_stride) {
    GatherArgs a;
    memset(&a, 0, sizeof(a));          // (l0 = nullptr: a stand-alone gather never reads the fine search's window buffer)
    a.src_kind = src.kind; a.level = level; a.len = len; a.tiles = tiles ? 1 : 0; a.ntaps = src.ntaps; a.pad = 0;
    a.raw = src.raw; a.raw_stride = src.raw_stride; a.arr = src.arr; a.arr_stride = src.arr_stride;
    a.coef = src.coef; a.dst = dst; a.dst_stream_stride = dst_stream_stride; a.dst_win_stride = dst_win_stride;
    const size_t lds = gather_lds(len, level, src.kind, src.ntaps);
    LAUNCH(c, k_gather, dim3(nwin_grid, S), dim3(256), lds, (const StreamState*)c->cur->state.p, a);
    CHECK_LAUNCH(c);
    return 0;
}

int ensure_twiddles(gsmcal_ctx* c, int nfft) {
    if (c->tw_n == nfft) return 0;
    RET_IF(ensure(c, c->tw, (size_t)nfft * sizeof(cplx)));
    LAUNCH(c, k_make_twiddles, dim3((nfft + 255) / 256), dim3(256), 0, (cplx*)c->tw.p, nfft);
    CHECK_LAUNCH(c);
    c->tw_n = nfft;
    ++c->ws_epoch;
    return 0;
}

size_t burst_scratch(const Geom& g) {   // w37 (40) | wN2 | P region: the SNR gate's rotator tables pw[16] | base[nfft/16 + 1], later P[2*hnl] in their place
    const size_t rot = ((size_t)16 + g.nfft / 16 + 2) * sizeof(cplx);
    const size_t pw = (size_t)2 * 56 * sizeof(double);      // hnl = ceil(148 * 200e3 / symbol_rate / 2) = 55 for every oversampling ratio
    return ((size_t)40 + g.nfft / 37) * sizeof(cplx) + (rot > pw ? rot : pw);
}

size_t fft_lds(const Geom& g) {   // xs | B[37][N2+1] | w37 (40) | wN2
    return ((size_t)g.nfft + (size_t)37 * (g.nfft / 37 + 1) + 40 + g.nfft / 37) * sizeof(cplx);
}

// decision steps that ride on a per-window kernel (stream_tail): one self-re-arming counter per stream
int make_tail(gsmcal_ctx* c, int S, const StepArgs& sa, int steps, int lvl_a, int lvl_b, TailArgs& t) {
    const size_t need = (size_t)S * sizeof(unsigned);
    if (c->cur->tailctr.cap < need) {
        RET_IF(ensure(c, c->cur->tailctr, need));
        HIPCHK(c, hipMemsetAsync(c->cur->tailctr.p, 0, c->cur->tailctr.cap, c->cur->stream));
    }
    t.ctr = (unsigned*)c->cur->tailctr.p;
    t.steps = steps; t.lvl_a = lvl_a; t.lvl_b = lvl_b; t.sa = sa;
    return 0;
}

DevParams dev_params(const gsmcal_ctx* c) {
    DevParams P;
    memset(&P, 0, sizeof(P));
    P.coarse_th = c->params.coarse_th_db; P.fine_max_ppm = c->params.fine_max_ppm; P.fine_gate_snr = c->params.fine_gate_snr_db;
    P.sch_max_ppm = c->params.sch_max_ppm; P.scan_spacing = c->params.scan_spacing; P.scan_spacing_idle = c->params.scan_spacing_idle;
    P.scan_tol = c->params.scan_tol; P.min_hits = c->params.min_hits; P.post_min_bcch = c->params.post_min_bcch;
    P.scan_min_hits = c->params.scan_min_hits;
    return P;
}

StepArgs step_args(gsmcal_ctx* c, const Geom& g, int H, int len_ts) {
    StepArgs a;
    memset(&a, 0, sizeof(a));
    a.P = dev_params(c);
    a.ov = g.ov; a.H = H; a.NB = g.NB; a.len_ts = len_ts;
    a.peaks = (const PeakOut*)c->cur->peaks.p;
    a.carrier_freq = c->cf_lane ? c->cf_lane : (const double*)c->cf.p;
    return a;
}

// Workgroups of the fused tail (k_post_chain_r<8,512,47> | k_post_chain_r<0,0,0> | k_post_chain) that fit one CU at this
// dynamic LDS size, from the occupancy calculator of the runtime (registers, LDS granules, wave slots of the compiled kernel);
// cached per (variant, LDS size).  0: the query failed -- the four-launch tail is used.
int post_chain_blocks_per_cu(gsmcal_ctx* c, int variant, size_t lds) {
    for (const auto& e : c->occ_cache)
        if (e.variant == variant && e.lds == lds) return e.blocks;
    int nb = 0;
    hipError_t r;
    if (variant == 0) r = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_post_chain_r<8, 512, 47>, PC_THREADS, lds);
    else if (variant == 1) r = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_post_chain_r<0, 0, 0>, PC_THREADS, lds);
    else r = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_post_chain, PC_THREADS, lds);
    if (r != hipSuccess) { (void)hipGetLastError(); nb = 0; }
    if (c->post_slots_cap > 0 && nb > c->post_slots_cap) nb = c->post_slots_cap;
    c->occ_cache.push_back({variant, lds, nb});
    return nb;
}

// ---- FCCH_fine_correction body (input at level lvl; creates levels lvl+1 (lerp), lvl+2 (mix)) ----
// setup_done: the window setup already ran at the end of k_coarse_scan (batch path).
// next_sch_lvl >= 0: also run SCH_corr_rate_correction's window setup in the last decision launch.
// chain != nullptr (batch path): everything behind the chunk sweep -- k_fine_verify and the three per-burst stages of
// FCCH_fine_correction / SCH_corr_rate_correction / carrier_correct_post_SCH -- goes out as ONE k_post_chain launch; the
// caller then skips run_sch / run_post.  *chain is set to false where the geometry does not allow it.
struct ChainOut { double* table; double* pos_info_out; long* r_len_out; bool fused; };
int run_fine(gsmcal_ctx* c, int S, const Source& src, int lvl, const Geom& g, int H, bool setup_done,
             int next_sch_lvl, int len_ts, ChainOut* chain = nullptr) {
    StreamState* st = (StreamState*)c->cur->state.p;
    const long wstride = g.fine_wlen, sstride = (long)H * g.fine_wlen;
    RET_IF(ensure(c, c->cur->win, (size_t)S * sstride * sizeof(cplx)));
    RET_IF(ensure(c, c->cur->peaks, (size_t)S * H * g.NB * sizeof(PeakOut)));
    RET_IF(ensure_twiddles(c, g.nfft));
    cplx* win = (cplx*)c->cur->win.p;
    PeakOut* peaks = (PeakOut*)c->cur->peaks.p;
    const StepArgs sa = step_args(c, g, H, len_ts);
    if (!setup_done) LAUNCH(c, k_step<STEP_FINE_SETUP>, dim3(S), dim3(64), 0, st, sa, lvl, 0);
    // raw sources at level 0: the certificate kernel builds the windows itself when a staging pass fits its free LDS
    const int fc_thr = fc_threads(g.fine_nshift);
    const size_t clds = (fc_lds_bytes(g.fine_nshift, g.nfft) + 15) & ~(size_t)15;
    const bool cert_ok = c->prescreen && c->certify && fc_thr <= 512 && clds <= 159 * 1024 &&
                         (g.fine_nshift - 1) % FS_CHUNK == 0 && g.nfft % 148 == 0 && g.nfft >= 2 * FC_NB;
    FusedGather fg;
    memset(&fg, 0, sizeof(fg));
    if (cert_ok && src.kind == SRC_RAW && lvl == 0 && c->fuse_fine_gather) {
        const size_t avail = clds - (size_t)FC_XP(g.fine_wlen) * sizeof(cplx) - 16;
        for (int np = 2; np <= 8; ++np) {
            const int per = ((g.fine_wlen + np - 1) / np + 3) & ~3;
            if ((per + src.ntaps + 14) / 8 + 1 <= fc_thr && fc_stage_bytes(per, src.ntaps) <= avail) {
                fg.raw = src.raw; fg.raw_stride = src.raw_stride; fg.coef = src.coef; fg.win_out = win;
                fg.ntaps = src.ntaps; fg.per = per;
                bool sym = src.ntaps == 47 && (int)c->h_coef.size() == 47 && c->fcert_s47;
                for (int k = 0; sym && k < 23; ++k) sym = c->h_coef[k] == c->h_coef[46 - k];
                fg.sym47 = sym ? 1 : 0;
                break;
            }
        }
    }
    if (!fg.raw) RET_IF(launch_gather(c, S, src, lvl, g.fine_wlen, false, H, win, sstride, wstride));
    // from here on the lane's window buffer holds level 0 of every fine window (nothing later in a batch call writes it)
    c->cur->win_l0_len = (src.kind == SRC_RAW && lvl == 0 && c->reuse_l0) ? g.fine_wlen : 0;
    c->cur->win_l0_H = H;
    StepArgs sa_fine = sa;
    if (c->prescreen) {
        // certificate (exact, tone bins) -> packed-fp32 sweep of the chunks it left open -> exact fp64 on what survives
        const int nchunk = (g.fine_nshift - 1 + FS_CHUNK - 1) / FS_CHUNK;
        if (nchunk > 255) return GSMCAL_E_UNSUPPORTED;     // (items carry the chunk in 8 bits; ov <= 127)
        if ((long)S * H >= (1L << 23)) return GSMCAL_E_UNSUPPORTED;
        RET_IF(ensure(c, c->cur->chunkrec, (size_t)S * H * nchunk * sizeof(ChunkRec)));
        // open-chunk work list: [0] = count (cleared by k_fine_verify after use), [4..] = items
        const size_t need_list = ((size_t)S * H * nchunk + 4) * sizeof(int);
        if (c->cur->openlist.cap < need_list) {
            RET_IF(ensure(c, c->cur->openlist, need_list));
            HIPCHK(c, hipMemsetAsync(c->cur->openlist.p, 0, 16, c->cur->stream));
        }
        int* n_open = (int*)c->cur->openlist.p;
        int* open_items = n_open + 4;
        const FineCert* certp = nullptr;
        if (cert_ok) {
            RET_IF(ensure(c, c->cur->cert, (size_t)S * H * sizeof(FineCert)));
            if (g.ov == 8 && (!fg.raw || fg.ntaps == 47))
                LAUNCH(c, (k_fine_cert<8, 47>), dim3(H, S), dim3(fc_thr), clds, (const StreamState*)st, (const cplx*)win, sstride, wstride,
                       g.fine_nshift, g.nfft, (const cplx*)c->tw.p, (FineCert*)c->cur->cert.p, H, open_items, n_open, fg);
            else
                LAUNCH(c, (k_fine_cert<0, 0>), dim3(H, S), dim3(fc_thr), clds, (const StreamState*)st, (const cplx*)win, sstride, wstride,
                       g.fine_nshift, g.nfft, (const cplx*)c->tw.p, (FineCert*)c->cur->cert.p, H, open_items, n_open, fg);
            certp = (const FineCert*)c->cur->cert.p;
        } else {
            LAUNCH(c, k_fine_openall, dim3(H, S), dim3(64), 0, (const StreamState*)st, nchunk, H, open_items, n_open);
        }
        long nblk = (long)S * H * nchunk;                  // persistent blocks: two of these 10-wave blocks are resident per CU
        if (nblk > 2 * c->n_cu) nblk = 2 * c->n_cu;
        LAUNCH(c, k_fine_chunk, dim3((unsigned)nblk), dim3(FK_THREADS), fk_lds_bytes(g.nfft),
               (const cplx*)win, sstride, wstride, g.fine_nshift, g.nfft, (const cplx*)c->tw.p, certp,
               (ChunkRec*)c->cur->chunkrec.p, H, (const int*)open_items, (const int*)n_open);
        const size_t vlds = ((size_t)g.fine_wlen * sizeof(cplx) + FV_MAX_ITEMS * (sizeof(int) + sizeof(cplx)) + 15) & ~(size_t)15;
        sa_fine.NB = 1;
        if (chain) {
            // ---- the fused tail of the chain: verify -> bursts -> SCH windows -> post-SCH bursts in one launch ----
            const int wl_sch = g.sch_nshift - 1 + len_ts;
            const size_t sch_scratch = (size_t)(len_ts + g.sch_nshift * SCH_PARTS) * sizeof(cplx) + (size_t)g.sch_nshift * sizeof(double);
            // replicated decisions (k_post_chain_r): the state copy stays in LDS in front of the stages' work area, and
            // the burst stages stage their (rare) raw-byte fallback without bank padding so that three workgroups still fit a CU
            const bool repl = c->post_repl && H <= MAXH;
            size_t lds = vlds;
            lds = std::max(lds, fused_lds(src, lvl + 1, g.nfft, burst_scratch(g), repl));
            lds = std::max(lds, fused_lds(src, lvl + 2, wl_sch, sch_scratch));
            lds = std::max(lds, fused_lds(src, lvl + 3, g.nfft, burst_scratch(g), repl));
            lds = std::max(lds, (sizeof(StreamState) + 15) & ~(size_t)15);
            if (repl) lds += PCR_STATE_BYTES;
            const bool ref_geom = g.ov == 8 && g.nfft == 148 * 8 && g.fine_nshift == 128 * 8 + 1 && g.sch_nshift == 11 * 8 + 1 &&
                                  len_ts == 512 && wl_sch == 11 * 8 + 512 && src.ntaps == 47;
            // only while every workgroup of the launch is resident at once: a workgroup waiting at a stream barrier holds its
            // slot, which costs nothing in the latency regime (64 streams: 0.234 vs 0.237 ms per step) and a fifth of the
            // throughput beyond it (256 streams: 0.73 vs 0.63 ms; 1024: 2.53 vs 2.13)
            // (and only for a call that runs on ONE lane: four lanes of 64 streams each would put 3 072 waiting workgroups on
            // 768 slots -- measured 0.70 against 0.63 ms at 256 streams).  The slots per CU come from the occupancy of the
            // very instantiation and LDS size that would be launched, not from a literal.
            const int variant = !repl ? 2 : (ref_geom ? 0 : 1);
            const int per_cu = lds <= 159 * 1024 ? post_chain_blocks_per_cu(c, variant, lds) : 0;
            chain->fused = c->fuse_post && cert_ok && src.kind == SRC_RAW && lvl == 0 && next_sch_lvl == 2 && per_cu > 0 &&
                           (long)H * S <= (long)per_cu * c->n_cu && c->n_lanes_used == 1;
            if (chain->fused) {
                const size_t need = (size_t)2 * S * sizeof(unsigned);
                if (c->cur->postctr.cap < need) {
                    RET_IF(ensure(c, c->cur->postctr, need));
                    HIPCHK(c, hipMemsetAsync(c->cur->postctr.p, 0, c->cur->postctr.cap, c->cur->stream));
                }
                // k_post_chain_r's exchange block [S][2 parities][4 stages][MAXH][2] and launch counters [S]: the layout does
                // not depend on the batch geometry (fixed MAXH stride per stage, one counter per stream in a buffer of its
                // own), and every launch leaves the parity it did not use EMPTY for all MAXH windows -- so launches of any
                // (S, H), eager or replayed from a graph captured here or by the caller, may follow each other.
                // Only growth re-creates the pair (all granules EMPTY, all counters zero); ensure() bumps ws_epoch then.
                const size_t need_x = (size_t)S * 2 * 4 * 2 * MAXH * sizeof(unsigned long long), need_e = (size_t)S * sizeof(unsigned);
                if (repl && (c->cur->xch.cap < need_x || c->cur->xepoch.cap < need_e)) {
                    if (c->capturing) return GSMCAL_E_HIP;      // (cannot happen: the eager call before a capture sized both)
                    RET_IF(ensure(c, c->cur->xch, need_x));
                    RET_IF(ensure(c, c->cur->xepoch, need_e));
                    HIPCHK(c, hipMemsetAsync(c->cur->xch.p, 0xFF, c->cur->xch.cap, c->cur->stream));
                    HIPCHK(c, hipMemsetAsync(c->cur->xepoch.p, 0, c->cur->xepoch.cap, c->cur->stream));
                }
                PostChainArgs pa;
                memset(&pa, 0, sizeof(pa));
                pa.ga1 = gather_args(src, lvl + 1, g.nfft);
                pa.ga_sch = gather_args(src, lvl + 2, wl_sch);
                pa.ga0 = gather_args(src, lvl + 3, g.nfft);
                if (c->cur->win_l0_len > 0) {
                    for (GatherArgs* ga : {&pa.ga1, &pa.ga0}) { ga->l0 = win; ga->l0_stream_stride = sstride; ga->l0_win_stride = wstride; ga->l0_len = c->cur->win_l0_len; }
                }
                if (repl) { pa.ga1.pad = 1; pa.ga0.pad = 1; }
                pa.sa = sa_fine;
                pa.sa.table = chain->table; pa.sa.pos_info_out = chain->pos_info_out; pa.sa.r_len_out = chain->r_len_out;
                pa.ctr = (unsigned*)c->cur->postctr.p; pa.gen = pa.ctr + S;
                pa.lvl_fine = lvl; pa.lvl_sch = lvl + 2; pa.lvl_post = lvl + 3;
                pa.nfft = g.nfft; pa.ov = g.ov; pa.len_ts = len_ts; pa.sch_nshift = g.sch_nshift; pa.fine_nshift = g.fine_nshift; pa.H = H;
                pa.tw_g = (const cplx*)c->tw.p; pa.ts = (const cplx*)c->ts.p;
                pa.win = win; pa.win_stream_stride = sstride; pa.win_stride = wstride;
                pa.rec = (const ChunkRec*)c->cur->chunkrec.p; pa.cert = certp; pa.peaks = peaks; pa.n_open = n_open;
                pa.with_totals = chain->table ? 1 : 0;
                if (repl && ref_geom) LAUNCH(c, (k_post_chain_r<8, 512, 47>), dim3(H, S), dim3(PC_THREADS), lds, st, pa, (unsigned long long*)c->cur->xch.p, (unsigned*)c->cur->xepoch.p);
                else if (repl) LAUNCH(c, (k_post_chain_r<0, 0, 0>), dim3(H, S), dim3(PC_THREADS), lds, st, pa, (unsigned long long*)c->cur->xch.p, (unsigned*)c->cur->xepoch.p);
                else LAUNCH(c, k_post_chain, dim3(H, S), dim3(PC_THREADS), lds, st, pa);
                CHECK_LAUNCH(c);
                return 0;
            }
        }
        TailArgs tl;
        RET_IF(make_tail(c, S, sa_fine, STEP_FINE_DECIDE, lvl, 0, tl));
        LAUNCH(c, k_fine_verify, dim3(H, S), dim3(FV_THREADS), vlds, (const StreamState*)st, (const cplx*)win, sstride, wstride,
               g.fine_nshift, g.nfft, (const cplx*)c->tw.p, (const ChunkRec*)c->cur->chunkrec.p, peaks, H, certp, n_open,
               st, tl);
    } else {
        RET_IF(ensure(c, c->cur->x0, (size_t)S * H * g.nfft * sizeof(cplx)));
        LAUNCH(c, k_fft_burst<1>, dim3(H, S), dim3(FFT_THREADS), fft_lds(g), (const StreamState*)st, (const cplx*)win, sstride,
               wstride, g.nfft, (const cplx*)c->tw.p, (PeakOut*)nullptr, (cplx*)c->cur->x0.p, H);
        LAUNCH(c, k_fine_search, dim3(g.NB, H, S), dim3(256), (size_t)(g.fine_nshift - 1 + FS_CHUNK) * sizeof(cplx),
               (const StreamState*)st, (const cplx*)win, sstride, wstride, g.fine_nshift, g.nfft, (const cplx*)c->cur->x0.p,
               peaks, H, g.NB);
        LAUNCH(c, k_step<STEP_FINE_DECIDE>, dim3(S), dim3(64), 0, st, sa_fine, lvl, 0);
    }
    // bursts of the resampled (not yet derotated) stream: level lvl+1 -- gather, spectrum argmax, tone estimate
    // and SNR gate fused per burst
    {
        GatherArgs ga = gather_args(src, lvl + 1, g.nfft);
        if (c->cur->win_l0_len > 0) { ga.l0 = win; ga.l0_stream_stride = sstride; ga.l0_win_stride = wstride; ga.l0_len = c->cur->win_l0_len; }
        TailArgs tl;   // FCCH_fine_correction's carrier decision (+ the SCH stage's window setup) rides on the last burst
        RET_IF(make_tail(c, S, sa, next_sch_lvl >= 0 ? (STEP_CARRIER_DECIDE | STEP_SCH_SETUP) : STEP_CARRIER_DECIDE, lvl,
                         next_sch_lvl >= 0 ? next_sch_lvl : 0, tl));
        LAUNCH_GEOM(g.ov == 8 && src.kind == SRC_RAW && src.ntaps == 47 && lvl == 0, c, (k_burst_tone<1, 8, 47>), (k_burst_tone<1, 0, 0>), dim3(H, S), dim3(BT_THREADS), fused_lds(src, lvl + 1, g.nfft, burst_scratch(g)), st, ga,
               g.nfft, (const cplx*)c->tw.p, g.ov, 1, tl);
    }
    CHECK_LAUNCH(c);
    return 0;
}

// ---- SCH_corr_rate_correction body (input at level lvl; creates level lvl+1) ----
int run_sch(gsmcal_ctx* c, int S, const Source& src, int lvl, const Geom& g, int H, int len_ts, bool setup_done,
            int next_post_lvl) {
    StreamState* st = (StreamState*)c->cur->state.p;
    const int wl = g.sch_nshift - 1 + len_ts;
    const long wstride = g.fine_wlen > wl ? g.fine_wlen : wl, sstride = (long)H * wstride;
    RET_IF(ensure(c, c->cur->win, (size_t)S * sstride * sizeof(cplx)));
    const StepArgs sa = step_args(c, g, H, len_ts);
    if (!setup_done) LAUNCH(c, k_step<STEP_SCH_SETUP>, dim3(S), dim3(64), 0, st, sa, 0, lvl);
    {
        const GatherArgs ga = gather_args(src, lvl, wl);
        const size_t scratch = (size_t)(len_ts + g.sch_nshift * SCH_PARTS) * sizeof(cplx) + (size_t)g.sch_nshift * sizeof(double);
        TailArgs tl;   // SCH_corr_rate_correction's decisions (+ the post stage's window setup) ride on the last window
        RET_IF(make_tail(c, S, sa, next_post_lvl >= 0 ? (STEP_SCH_DECIDE | STEP_POST_SETUP) : STEP_SCH_DECIDE, lvl,
                         next_post_lvl >= 0 ? next_post_lvl : 0, tl));
        LAUNCH_GEOM(g.ov == 8 && len_ts == 512 && src.kind == SRC_RAW && src.ntaps == 47 && lvl == 2, c, (k_window_sch<8, 512, 47>), (k_window_sch<0, 0, 0>), dim3(H, S), dim3(512), fused_lds(src, lvl, wl, scratch), st, ga, (const cplx*)c->ts.p,
               len_ts, g.sch_nshift, tl);
    }
    CHECK_LAUNCH(c);
    return 0;
}

// ---- carrier_correct_post_SCH body (input at level lvl; creates level lvl+1 (mix)) ----
// table != nullptr: also write the calibration table row (gsm_sync_demod.m:123-124) in the last launch.
int run_post(gsmcal_ctx* c, int S, const Source& src, int lvl, const Geom& g, int H, bool setup_done, double* table,
             double* pos_info_out, long* r_len_out) {
    StreamState* st = (StreamState*)c->cur->state.p;
    const long wstride = g.fine_wlen, sstride = (long)H * g.fine_wlen;
    RET_IF(ensure(c, c->cur->win, (size_t)S * sstride * sizeof(cplx)));
    RET_IF(ensure(c, c->cur->peaks, (size_t)S * H * g.NB * sizeof(PeakOut)));
    RET_IF(ensure_twiddles(c, g.nfft));
    cplx* win = (cplx*)c->cur->win.p;
    StepArgs sa = step_args(c, g, H, 0);
    sa.table = table; sa.pos_info_out = pos_info_out; sa.r_len_out = r_len_out;
    if (!setup_done) LAUNCH(c, k_step<STEP_POST_SETUP>, dim3(S), dim3(64), 0, st, sa, 0, lvl);
    {
        GatherArgs ga = gather_args(src, lvl, g.nfft);
        if (c->cur->win_l0_len == g.fine_wlen && c->cur->win_l0_H == H && src.kind == SRC_RAW) {
            ga.l0 = win; ga.l0_stream_stride = sstride; ga.l0_win_stride = wstride; ga.l0_len = c->cur->win_l0_len;
        }
        TailArgs tl;   // carrier_correct_post_SCH's decision (+ the calibration table row) rides on the last burst
        RET_IF(make_tail(c, S, sa, table ? (STEP_POST_DECIDE | STEP_TOTALS) : STEP_POST_DECIDE, lvl, 0, tl));
        LAUNCH_GEOM(g.ov == 8 && src.kind == SRC_RAW && src.ntaps == 47 && lvl == 3, c, (k_burst_tone<0, 8, 47>), (k_burst_tone<0, 0, 0>), dim3(H, S), dim3(BT_THREADS), fused_lds(src, lvl, g.nfft, burst_scratch(g)), st, ga,
               g.nfft, (const cplx*)c->tw.p, g.ov, 0, tl);
    }
    CHECK_LAUNCH(c);
    return 0;
}

int init_states(gsmcal_ctx* c, int S, long n0) {
    (void)n0;   // written with the other defaults by k_finish_mean
    RET_IF(ensure(c, c->cur->state, (size_t)S * sizeof(StreamState)));
    HIPCHK(c, hipMemsetAsync(c->cur->state.p, 0, (size_t)S * sizeof(StreamState), c->cur->stream));
    c->last_S = S;
    return 0;
}

int dc_means(gsmcal_ctx* c, const uint8_t* d_raw, int S, long n) {
    int blocks = (int)((2 * n / 16 + 256 * 8 - 1) / (256 * 8));
    if (blocks < 1) blocks = 1;
    int cap = 4096 / (S > 0 ? S : 1);
    if (cap < 1) cap = 1;
    if (blocks > cap) blocks = cap;
    LAUNCH(c, k_dc_sum, dim3(blocks, S), dim3(256), 0, d_raw, 2 * n, (StreamState*)c->cur->state.p);
    LAUNCH(c, k_finish_mean, dim3((S + 63) / 64), dim3(64), 0, (StreamState*)c->cur->state.p, S, n);
    CHECK_LAUNCH(c);
    return 0;
}

int fir_decim_raw(gsmcal_ctx* c, const uint8_t* d_raw, int S, long n, const double* d_coef, int ntaps, int decim,
                  cplx* d_out, long out_stride) {
    const long nd = (n + decim - 1) / decim;
    const size_t span = (size_t)256 * decim + ntaps + 24;
    const size_t lds = (size_t)((ntaps * 8 + 15) & ~15) + (span + span / 8 + 16) * 2;
    if (lds > 159 * 1024) return GSMCAL_E_UNSUPPORTED;
    LAUNCH(c, k_fir_decim_raw, dim3((unsigned)((nd + 255) / 256), S), dim3(256), lds, d_raw, 2 * n,
           (const StreamState*)c->cur->state.p, d_coef, ntaps, decim, nd, d_out, out_stride);
    CHECK_LAUNCH(c);
    return 0;
}

// batch front end: one pass over the raw bytes (per-block byte sums + FIR of the raw samples).  The means are
// formed from the partial sums by the coarse kernels, and k_coarse_scan builds each stream's state from scratch,
// so the batch path needs neither a memset of the state array nor a separate mean kernel.
// the two instances of the register-row front kernel (named so that profiles show them apart)
static const auto k_front_fast47_sym = &k_front_fast<47, true>;
static const auto k_front_fast47 = &k_front_fast<47, false>;
static const auto k_front_fast31_sym = &k_front_fast<31, true>;
static const auto k_front_fast31 = &k_front_fast<31, false>;

// instances of the coarse scan: 16-point windows with the latency / throughput register budgets, and any window length
static const auto k_coarse_scan_lat = &k_coarse_scan<3, true>;   // (one register budget serves both: no spills at 165 registers)
static const auto k_coarse_scan_thr = &k_coarse_scan<3, true>;
static const auto k_coarse_scan_gen = &k_coarse_scan<2, false>;
static const auto k_coarse_scan_ref = &k_coarse_scan<3, true, true>;   // the drivers' window geometry as constants
static const auto k_coarse_scan_inl = &k_coarse_scan<3, true, true, true>;   // ... with the window SNRs computed in place (throughput batches)

// (s_off, S_all: streams [s_off, s_off + S) of a lane that holds S_all -- the scanner pipeline launches a stage's front kernel in two parts)
int front_fused(gsmcal_ctx* c, const uint8_t* d_raw, int S, long n, const double* d_coef, int ntaps, int decim,
                cplx* d_out, long out_stride, int s_off = 0, int S_all = 0) {
    if (S_all < S + s_off) S_all = S + s_off;
    const long nd = (n + decim - 1) / decim;
    const size_t span = (size_t)256 * decim + ntaps + 24;
    const size_t lds = (size_t)((ntaps * 8 + 15) & ~15) + (span + span / 8 + 16) * 2;
    if (lds > 159 * 1024) return GSMCAL_E_UNSUPPORTED;
    const unsigned nblk = (unsigned)((nd + 255) / 256);
    RET_IF(ensure(c, c->cur->state, (size_t)S_all * sizeof(StreamState)));
    RET_IF(ensure(c, c->cur->partial, (size_t)S_all * nblk * 4 * 2 * sizeof(unsigned long long)));
    c->cur->npartial = (int)nblk;
    c->last_S = S_all;
    d_raw += (size_t)s_off * 2 * n;
    d_out += (size_t)s_off * out_stride;
    bool sym = (int)c->h_coef.size() == ntaps;                 // linear-phase taps? (fir1 and the .fda designs are)
    for (int k = 0; sym && k < ntaps / 2; ++k) sym = c->h_coef[k] == c->h_coef[ntaps - 1 - k];
    if ((ntaps == 47 || ntaps == 31) && decim == 64 && ((uintptr_t)d_raw & 15) == 0 && ((2 * n) & 15) == 0 &&
        !c->front_generic) {
        // the production geometries (fir1(46) / fir1(30), 8x oversampling, aligned captures): rows in registers
        const size_t flds = (size_t)2048 * 16;              // swizzled, unpadded: five workgroups per CU
        c->cur->npartial = (int)nblk * 4;                  // this kernel writes one partial per wave
        // Raw bytes of a call that the Infinity Cache (256 MiB) cannot hold are read with non-temporal loads: 800 captures (1 GB)
        // 199-216 -> 172-174 us = 6.7 TB/s for the kernel, the call 0.30-0.31 -> 0.276 ms; 12 800 captures 3.86 -> 3.75 ms.
        // A smaller batch that the caller processes again (bench.py's 64 streams, 130 MB; 200 captures, 244 MiB) is served from the
        // Infinity Cache from the second step on and keeps plain loads: there nt costs 1 us of 22.6 / 4 us of 88.
        // GSMCAL_FRONT_NT=0/1 overrides.
        const int nt = c->front_nt >= 0 ? c->front_nt : (c->call_raw_bytes > ((size_t)256 << 20) ? 1 : 0);
#define FRONT_FAST(K) LAUNCH(c, K, dim3(nblk, S), dim3(256), flds, d_raw, 2 * n, (unsigned long long*)c->cur->partial.p + (size_t)s_off * nblk * 4 * 2, d_coef, nd, d_out, out_stride, nt)
        if (ntaps == 47) { if (sym) FRONT_FAST(k_front_fast47_sym); else FRONT_FAST(k_front_fast47); }
        else { if (sym) FRONT_FAST(k_front_fast31_sym); else FRONT_FAST(k_front_fast31); }
#undef FRONT_FAST
    } else {
        LAUNCH(c, k_front_fused, dim3(nblk, S), dim3(256), lds, d_raw, 2 * n, (unsigned long long*)c->cur->partial.p + (size_t)s_off * nblk * 2, d_coef,
               ntaps, decim, nd, d_out, out_stride, sym ? 1 : 0);
    }
    CHECK_LAUNCH(c);
    return 0;
}

int hits_capacity(long len_dec, int dec_ratio) {
    // FCCH_coarse_position.m:38 max_num_fcch = ceil(len/(10*num_sym_per_frame/decimation_ratio))
    int h = (int)ceil((double)len_dec / (12500.0 / (double)dec_ratio));
    if (h < 1) h = 1;
    return h;
}

size_t coarse_scan_lds(long nwin, int mv_len) {
    return coarse_scan_lds_fixed() + (size_t)(nwin + mv_len + 128) * sizeof(double);
}

// Partial tap sums of the head rows (see coarse()): uploaded on the context's stream BEFORE fork_lanes(), so the fork
// event orders the copy ahead of every lane's k_coarse_snr (inside coarse() only lane 0 was ordered behind it).
int ensure_head(gsmcal_ctx* c, int front_decim) {
    const int ntaps = (int)c->h_coef.size();
    const int n_head = ntaps > 1 ? (ntaps - 1 + front_decim - 1) / front_decim : 1;
    if ((int)c->h_head.size() == n_head && c->head_epoch == c->coef_epoch && c->csum_head.p) return 0;
    c->h_head.assign(n_head, 0.0);
    for (int j = 0; j < n_head; ++j) {
        double h = 0.0;
        for (int k = 0; k < ntaps && k <= (long)front_decim * j; ++k) h += c->h_coef[k];
        c->h_head[j] = h;
    }
    RET_IF(ensure(c, c->csum_head, (size_t)n_head * sizeof(double)));
    HIPCHK(c, hipMemcpyAsync(c->csum_head.p, c->h_head.data(), (size_t)n_head * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->head_epoch = c->coef_epoch;
    ++c->ws_epoch;
    return 0;
}

int coarse(gsmcal_ctx* c, int S, const cplx* d_dec, long stride, long len, int dec_ratio, int fine_setup_ov,
           bool mean_corr = false, long n0 = 0, int front_decim = 64, const ScanAccept* accept = nullptr, bool allow_inline = true) {
    CoarseArgs a;
    memset(&a, 0, sizeof(a));
    if (accept) { a.accept = *accept; a.P = dev_params(c); }
    if (mean_corr) {   // input = FIR of the raw bytes (front_fused): DC removed on load
        a.mean_corr = 1;
        a.partial = (const unsigned long long*)c->cur->partial.p;
        a.npartial = c->cur->npartial;
        a.n0 = n0;
        double cs = 0.0;
        for (double v : c->h_coef) cs += v;
        a.csum_all = cs;
        // decimated rows j with front_decim*j < ntaps-1 see only taps 0..front_decim*j (zero initial state of filter()):
        // their partial tap sums were uploaded by ensure_head() before the lanes forked
        const int ntaps = (int)c->h_coef.size();
        const int n_head = ntaps > 1 ? (ntaps - 1 + front_decim - 1) / front_decim : 1;
        if ((int)c->h_head.size() != n_head || c->head_epoch != c->coef_epoch) { c->err = "coarse(): csum_head not prepared"; return GSMCAL_E_ARG; }
        a.csum_head = (const double*)c->csum_head.p;
        a.n_head = n_head;
    }
    a.s = d_dec; a.s_stride = stride; a.len = len; a.decimation_ratio = dec_ratio; a.mode = 0;
    a.th0 = c->params.coarse_th_db; a.min_hits = c->params.min_hits;
    a.fine_setup_ov = fine_setup_ov;
    const int fft_len = 1 << (int)floor(log2(148.0 / (double)dec_ratio));
    const long n_first = (long)ceil(23.0 * 1250.0 / (double)dec_ratio);
    const long nwin = n_first - (fft_len - 1);
    a.g_fft_len = fft_len; a.g_n_first = n_first;
    const size_t lds = coarse_scan_lds(n_first, 10 * fft_len);
    if (lds > 159 * 1024 || nwin < 1) return GSMCAL_E_UNSUPPORTED;
    // latency path (few streams: one wave of k_coarse_snr workgroups still fits the chip): k_coarse_snr fills in every window of
    // the stream and the hop walk of k_coarse_scan becomes table look-ups; bigger batches keep the table to the moving
    // search's windows (at 200 captures the longer table kernel already costs what the shorter walk saves)
    long ntab = nwin;
    unsigned sblocks = (unsigned)((nwin + 255) / 256);
    if (fft_len == 16 && 2 * S <= c->n_cu && c->snr_full && len - (fft_len - 1) > nwin) {
        ntab = len - (fft_len - 1);
        a.snr_nwin = ntab;
        a.snr_screen_db = c->snr_screen_db;
        {
            const double rho = pow(10.0, a.snr_screen_db / 10.0);
            const double gx = (0.9238795325112867 * rho - 1.0) / (rho + 1.0);
            a.snr_gx2 = gx > 0.0 ? gx * gx * (1.0 - 1e-9) : 0.0;   // (margin over the ~1e-14 rounding of the sums)
        }
        const long rest = ntab - nwin;
        if ((rest + sblocks - 1) / sblocks > CS_TILE - 3) sblocks = (unsigned)((rest + CS_TILE - 4) / (CS_TILE - 3));
        a.snr_tile = (int)((((rest + sblocks - 1) / sblocks) + 3) & ~3L);
    }
    const bool refg = dec_ratio == 8 && fft_len == 16 && n_first == 3594;
    // throughput batches (every batch too big for the full table above): the scan kernel computes the moving search's SNRs itself
    // -- no table in HBM, one launch less: 200 / 800 captures 0.091 / 0.317 -> 0.088 / 0.303 ms, 1 024 streams 1.813 -> 1.782 ms.
    // The table is written out only on request (GSMCAL_SNR_INLINE_KEEP=1); gsmcal_last_batch_snr has nothing to return otherwise.
    // In the scanner's pipeline stages only while the stage's workgroups are all resident at once (<= 3 per CU): with 800-capture
    // stages the second, partial round of this long kernel beside the next stage's front kernel cost more than the table's
    // traffic saved (12 800 captures 3.85 -> 3.98 ms); with 534-capture stages it wins (3.71 -> 3.69).
    if (allow_inline && refg && a.snr_nwin == 0 && c->snr_inline_min > 0 && S >= c->snr_inline_min) {
        a.snr_g = nullptr; a.snr_stride = ntab;
        c->cur->snr_stride = 0; c->cur->snr_nmove = nwin;
        if (c->snr_inline_keep) {
            RET_IF(ensure(c, c->cur->snrbuf, (size_t)S * ntab * sizeof(double)));
            a.snr_g = (double*)c->cur->snrbuf.p;
            c->cur->snr_stride = ntab;
        }
        LAUNCH(c, k_coarse_scan_inl, dim3(S), dim3(256), lds, (StreamState*)c->cur->state.p, a);
        CHECK_LAUNCH(c);
        return 0;
    }
    RET_IF(ensure(c, c->cur->snrbuf, (size_t)S * ntab * sizeof(double)));
    a.snr_g = (double*)c->cur->snrbuf.p; a.snr_stride = ntab;
    c->cur->snr_stride = ntab; c->cur->snr_nmove = nwin;
    const dim3 sgrid(sblocks, S);
    if (fft_len == 16 && a.snr_nwin > 0) LAUNCH_GEOM(refg, c, (k_coarse_snr<true, true, true>), (k_coarse_snr<true, true>), sgrid, dim3(CS_SNR_THREADS), 0, a);
    else if (fft_len == 16) LAUNCH_GEOM(refg, c, (k_coarse_snr<true, false, true>), (k_coarse_snr<true>), sgrid, dim3(256), 0, a);
    else LAUNCH(c, k_coarse_snr<false>, sgrid, dim3(256), 0, a);
    // register budgets of the same kernel: small batches run one workgroup per CU anyway, big ones want four
    if (fft_len != 16) LAUNCH(c, k_coarse_scan_gen, dim3(S), dim3(256), lds, (StreamState*)c->cur->state.p, a);
    else if (refg) LAUNCH(c, k_coarse_scan_ref, dim3(S), dim3(256), lds, (StreamState*)c->cur->state.p, a);
    else if (S <= 512) LAUNCH(c, k_coarse_scan_lat, dim3(S), dim3(256), lds, (StreamState*)c->cur->state.p, a);
    else LAUNCH(c, k_coarse_scan_thr, dim3(S), dim3(256), lds, (StreamState*)c->cur->state.p, a);
    CHECK_LAUNCH(c);
    return 0;
}

int fetch_states(gsmcal_ctx* c, int S, std::vector<StreamState>& out) {
    out.resize(S);
    HIPCHK(c, hipMemcpyAsync(out.data(), c->cur->state.p, (size_t)S * sizeof(StreamState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int push_states(gsmcal_ctx* c, const std::vector<StreamState>& in) {
    RET_IF(ensure(c, c->cur->state, in.size() * sizeof(StreamState)));
    HIPCHK(c, hipMemcpyAsync(c->cur->state.p, in.data(), in.size() * sizeof(StreamState), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

void host_init_state(StreamState& st, long n0) {
    memset(&st, 0, sizeof(st));
    st.n0 = n0;
    st.hit_avg_snr = INFINITY;
    st.sampling_ppm1 = st.carrier_ppm1 = st.sampling_ppm2 = st.carrier_ppm2 = INFINITY;
    st.fcch_is_sentinel = 1;
}

// materialise level `level` of stream 0 (API mode, array source) into host buffer r
int materialise_to_host(gsmcal_ctx* c, const Source& src, int level, long n_out, double* r) {
    RET_IF(ensure(c, c->arr_out, (size_t)n_out * sizeof(cplx)));
    const int tiles = (int)((n_out + TILE - 1) / TILE);
    RET_IF(launch_gather(c, 1, src, level, TILE, true, tiles, (cplx*)c->arr_out.p, n_out, 0));
    HIPCHK(c, hipMemcpyAsync(r, c->arr_out.p, (size_t)n_out * sizeof(cplx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int upload_array(gsmcal_ctx* c, const double* s, size_t n_cplx) {
    RET_IF(ensure(c, c->arr_in, n_cplx * sizeof(cplx)));
    HIPCHK(c, hipMemcpyAsync(c->arr_in.p, s, n_cplx * sizeof(cplx), hipMemcpyHostToDevice, c->stream));
    return 0;
}

// Split d units over the lanes: returns the number of lanes used and fills lo/n per lane.
int plan_lanes(gsmcal_ctx* c, int d, bool latency_bound = true) {
    // Calibration chain (a string of short latency-bound kernels): lanes run side by side.  Scanner path: one
    // bandwidth-bound front kernel followed by the compute-bound detector -- side-by-side lanes only put two front
    // kernels in each other's way (measured: 1 lane 6.30 ms, 2 lanes 7.43 ms for 12,800 captures), so big scanner
    // batches are cut into pipeline stages instead: the front kernels run one after the other (chained by events) and
    // each stage's detector runs underneath the next stage's front kernel.
    // Stage size: at most 640 captures (at least 8 stages) -- a stage of at most 3 x 256 captures lets its detector run as ONE
    // resident round of k_coarse_scan<INL> workgroups (12 800 captures, with the split front launches: 16 two-kernel / 20 / 24 /
    // 32 stages 3.69 / 3.50 / 3.56 / 3.56 ms).  From 1 200 captures on four stages already pay (1 600 captures: 1 / 2 / 3 / 4 stages
    // 0.547 / 0.510 / 0.585 / 0.504 ms; 800: 0.269 / 0.268 / 0.334 / 0.274; 400: 0.146 / 0.163 / 0.217 / 0.194).
    int nl = latency_bound ? c->n_lanes_cfg : (c->scan_stages > 0 ? c->scan_stages : (d >= 1200 ? std::max(d >= 2048 ? 8 : 4, (d + 639) / 640) : 1));
    if (latency_bound && nl > d / c->lane_min) nl = d / c->lane_min;   // a minimum of streams per lane: below that splitting only adds launches
    if (nl > MAX_LANES) nl = MAX_LANES;
    if (nl < 1) nl = 1;
    for (int i = 0; i < nl; ++i) {
        c->lanes[i].lo = (int)(((long)i * d) / nl);
        c->lanes[i].n = (int)(((long)(i + 1) * d) / nl) - c->lanes[i].lo;
    }
    for (int i = nl; i < MAX_LANES; ++i) c->lanes[i].n = 0;
    c->n_lanes_used = nl;
    return nl;
}

int fork_lanes(gsmcal_ctx* c, int nl) {
    if (nl <= 1) return 0;
    if (!c->fork) HIPCHK(c, hipEventCreateWithFlags(&c->fork, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->fork, c->stream));
    for (int i = 1; i < nl; ++i) {
        Lane& L = c->lanes[i];
        if (!L.stream) HIPCHK(c, hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        if (!L.done) HIPCHK(c, hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
        HIPCHK(c, hipStreamWaitEvent(L.stream, c->fork, 0));
    }
    return 0;
}

int join_lanes(gsmcal_ctx* c, int nl) {
    for (int i = 1; i < nl; ++i) {
        HIPCHK(c, hipEventRecord(c->lanes[i].done, c->lanes[i].stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->lanes[i].done, 0));
    }
    c->cur = &c->lanes[0];
    return 0;
}

// Run `enqueue` (which only enqueues work on the context's streams) eagerly, or -- from the second identical
// call on -- as a captured hipGraph replayed with one hipGraphLaunch.  The first call runs eagerly so that every
// workspace buffer, lane stream and event exists before capture starts; the second captures, instantiates and
// replays.  Capture is never attempted where it cannot work -- the legacy NULL stream, or a user stream that is
// itself being captured (e.g. inside torch.cuda.graph) -- and any capture failure falls back to eager launches.
template <class F>
int run_maybe_graph(gsmcal_ctx* c, gsmcal_ctx::GraphSlot& slot, const std::vector<uintptr_t>& key, F enqueue, bool multi_stream) {
    // A plan on one stream (one lane, no pipeline stages) is launched eagerly: nine back-to-back launches ran 1-5 % faster
    // than replaying them as a graph (consecutive graph launches sit 8.6 us apart on the GPU's timeline; 0.238 vs 0.242 ms at
    // 64 streams, 0.125 vs 0.129 at 2, 0.101 vs 0.107 for 200 captures).  Plans that fork onto internal streams replay as a
    // graph: the event choreography costs more launched piecemeal (12 800 captures: 4.06 vs 4.33 ms).  GSMCAL_GRAPH=2: always.
    bool can_graph = c->use_graph && (multi_stream || c->graph_always) && !c->prof && c->stream != nullptr;
    if (can_graph) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(c->stream, &cs) != hipSuccess) { (void)hipGetLastError(); can_graph = false; }
        else if (cs != hipStreamCaptureStatusNone) can_graph = false;       // the caller is capturing: just enqueue
    }
    const bool same = can_graph && slot.key == key && slot.epoch == c->ws_epoch;
    if (same && slot.exec) {
        HIPCHK(c, hipGraphLaunch(slot.exec, c->stream));
        return 0;
    }
    if (!same) {
        if (slot.exec) { (void)hipGraphExecDestroy(slot.exec); slot.exec = nullptr; }
        if (slot.graph) { (void)hipGraphDestroy(slot.graph); slot.graph = nullptr; }
        slot.seen = 0;
    }
    if (same && slot.seen >= 1) {
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
            (void)hipGetLastError();
            c->use_graph = false;                       // this stream cannot be captured: eager launches for good
            return enqueue();
        }
        c->capturing = true;
        const int rc = enqueue();
        c->capturing = false;
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (rc < 0 || e != hipSuccess || !g) {
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            c->use_graph = false;
            if (rc < 0) return rc;
            return enqueue();
        }
        hipGraphExec_t ex = nullptr;
        if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
            (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            c->use_graph = false;
            return enqueue();
        }
        slot.graph = g;
        slot.exec = ex;
        HIPCHK(c, hipGraphLaunch(slot.exec, c->stream));
        return 0;
    }
    const int rc = enqueue();
    if (rc < 0) return rc;
    if (can_graph) {                                    // the call may have allocated / uploaded: remember the state AFTER it
        slot.key = key;
        slot.epoch = c->ws_epoch;
        slot.seen = 1;
    }
    return rc;
}

// the slot holding `key`, else the least recently used one
gsmcal_ctx::GraphSlot& pick_slot(gsmcal_ctx* c, gsmcal_ctx::GraphSlot* slots, const std::vector<uintptr_t>& key) {
    int pick = 0;
    bool hit = false;
    for (int i = 0; i < gsmcal_ctx::GRAPH_SLOTS && !hit; ++i)
        if (slots[i].key == key) { pick = i; hit = true; }
    if (!hit)
        for (int i = 1; i < gsmcal_ctx::GRAPH_SLOTS; ++i)
            if (slots[i].used < slots[pick].used) pick = i;
    slots[pick].used = ++c->g_stamp;
    return slots[pick];
}

int positive_status(const StreamState& st, int stage) {
    if (st.status < 0) return st.status;
    return st.stage_status[stage];
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char* gsmcal_version(void) { return GSMCAL_VERSION; }

void gsmcal_params_default(gsmcal_params* p) {
    if (!p) return;
    p->coarse_th_db = 10.0; p->coarse_mv_factor = 10; p->coarse_max_offset = 5; p->min_hits = 5;
    p->fine_max_offset = 64; p->fine_max_ppm = 4000.0; p->fine_gate_snr_db = 5.0; p->fine_noise_bw_hz = 200e3;
    p->sch_max_offset = 8; p->sch_max_ppm = 400.0; p->post_min_bcch = 4;
    p->scan_min_hits = 3; p->scan_spacing = 12500.0; p->scan_spacing_idle = 12500.0 + 1250.0; p->scan_tol = 50.0;
}

int gsmcal_set_params(gsmcal_ctx* c, const gsmcal_params* p) {
    if (!c || !p) return GSMCAL_E_ARG;
    gsmcal_params d;
    gsmcal_params_default(&d);
    if (p->coarse_mv_factor != d.coarse_mv_factor || p->coarse_max_offset != d.coarse_max_offset ||
        p->fine_max_offset != d.fine_max_offset || p->fine_noise_bw_hz != d.fine_noise_bw_hz || p->sch_max_offset != d.sch_max_offset) {
        c->err = "gsmcal_set_params: a geometry field differs from its default";
        return GSMCAL_E_UNSUPPORTED;
    }
    if (p->min_hits < 2 || p->min_hits > GSMCAL_MAX_HITS || p->scan_min_hits < 1 || p->post_min_bcch < 0) return GSMCAL_E_ARG;
    c->params = *p;
    ++c->params_epoch;
    return 0;
}

int gsmcal_get_params(gsmcal_ctx* c, gsmcal_params* p) {
    if (!c || !p) return GSMCAL_E_ARG;
    *p = c->params;
    return 0;
}

int gsmcal_ctx_create_on_stream(int device_id, void* hip_stream, gsmcal_ctx** out) {
    if (!out) return GSMCAL_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return GSMCAL_E_NO_DEVICE;
    if (device_id < 0 || device_id >= n) return GSMCAL_E_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return GSMCAL_E_HIP;
    // kernels whose dynamic LDS may exceed the 64 KiB default
    (void)hipFuncSetAttribute((const void*)k_gather, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_fir_decim_raw, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_fine_verify, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_fine_cert<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_fine_cert<8, 47>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_fine_chunk, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_front_fused, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_front_fast47_sym, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_front_fast47, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_front_fast31_sym, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_front_fast31, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_coarse_scan_lat, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_coarse_scan_thr, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_coarse_scan_inl, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_coarse_scan_gen, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_coarse_scan_ref, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_burst_tone<0, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_burst_tone<1, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_window_sch<0, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_post_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_post_chain_r<0, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_post_chain_r<8, 512, 47>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_sch_equalise, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_sch_fd_training, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_fft_burst<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_fft_burst<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_burst_tone<0, 8, 47>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_burst_tone<1, 8, 47>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipFuncSetAttribute((const void*)k_window_sch<8, 512, 47>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    (void)hipGetLastError();   // an attribute request the device rejects must not surface at the first launch
    gsmcal_ctx* c = new gsmcal_ctx();
    gsmcal_params_default(&c->params);
    c->device = device_id;
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    c->lanes[0].stream = c->stream;
    c->cur = &c->lanes[0];
    const char* e = getenv("GSMCAL_LANES");
    if (e && atoi(e) >= 1) c->n_lanes_cfg = atoi(e) > MAX_LANES ? MAX_LANES : atoi(e);
    { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && ncu > 0) c->n_cu = ncu; }
    if (const char* e2 = getenv("GSMCAL_SNR_INLINE_MIN")) c->snr_inline_min = atoi(e2);
    if (const char* e2 = getenv("GSMCAL_SNR_INLINE_KEEP")) c->snr_inline_keep = atoi(e2);
    if (const char* e2 = getenv("GSMCAL_FRONT_NT")) c->front_nt = atoi(e2);
    if (const char* e2 = getenv("GSMCAL_SCAN_SPLIT")) c->scan_split = atoi(e2);
    if (const char* e2 = getenv("GSMCAL_SNR_INLINE_PIPE")) c->snr_inline_pipe = atoi(e2);
    const char* sst = getenv("GSMCAL_SCAN_STAGES");
    if (sst && atoi(sst) >= 1) c->scan_stages = atoi(sst);
    const char* lm = getenv("GSMCAL_LANE_MIN");
    if (lm && atoi(lm) >= 1) c->lane_min = atoi(lm);
    const char* ce = getenv("GSMCAL_CERT");
    if (ce) c->certify = atoi(ce) != 0;
    const char* sfe = getenv("GSMCAL_SNR_FULL");
    if (sfe) c->snr_full = atoi(sfe) != 0;
    const char* rle = getenv("GSMCAL_REUSE_L0");
    if (rle) c->reuse_l0 = atoi(rle) != 0;
    const char* sse = getenv("GSMCAL_SNR_SCREEN_DB");
    if (sse) c->snr_screen_db = atof(sse);
    const char* fge = getenv("GSMCAL_FUSE_GATHER");
    if (fge) c->fuse_fine_gather = atoi(fge) != 0;
    const char* pre_ = getenv("GSMCAL_POST_REPL");
    if (pre_) c->post_repl = atoi(pre_) != 0;
    const char* lse = getenv("GSMCAL_LANE_STAGGER");
    if (lse) c->lane_stagger = atoi(lse) != 0 ? 1 : 0;
    const char* fpe = getenv("GSMCAL_FUSE_POST");
    if (fpe) c->fuse_post = atoi(fpe) != 0;
    const char* pse = getenv("GSMCAL_POST_SLOTS");
    if (pse && atoi(pse) >= 1) c->post_slots_cap = atoi(pse);
    const char* pe = getenv("GSMCAL_PRESCREEN");
    if (pe && atoi(pe) == 0) c->prescreen = false;
    const char* f47 = getenv("GSMCAL_FCERT_S47");
    if (f47) c->fcert_s47 = atoi(f47) != 0;
    const char* s47 = getenv("GSMCAL_STREAM_S47");
    if (s47) c->stream_s47 = atoi(s47) != 0;
    const char* fg = getenv("GSMCAL_FRONT_GENERIC");
    if (fg && atoi(fg) != 0) c->front_generic = true;
    const char* ge = getenv("GSMCAL_GRAPH");
    if (ge && atoi(ge) == 0) c->use_graph = false;
    if (ge && atoi(ge) == 2) c->graph_always = true;
    *out = c;
    return 0;
}

int gsmcal_ctx_create(int device_id, gsmcal_ctx** out) {
    int r = gsmcal_ctx_create_on_stream(device_id, nullptr, out);
    if (r != 0) return r;
    gsmcal_ctx* c = *out;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        *out = nullptr;
        return GSMCAL_E_HIP;
    }
    c->own_stream = true;
    c->lanes[0].stream = c->stream;
    return 0;
}

void gsmcal_ctx_destroy(gsmcal_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipDeviceSynchronize();
    DevBuf* bufs[] = {&c->coef, &c->ts, &c->cf, &c->table, &c->snrhit, &c->arr_in, &c->arr_out, &c->posinfo, &c->rlen,
                      &c->misc, &c->tw, &c->csum_head, &c->tw_sch};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (int i = 0; i < MAX_LANES; ++i) {
        Lane& L = c->lanes[i];
        DevBuf* lb[] = {&L.state, &L.dec, &L.win, &L.peaks, &L.snrbuf, &L.x0, &L.chunkrec, &L.openlist, &L.cert, &L.partial, &L.tailctr, &L.postctr, &L.xch, &L.xepoch};
        for (DevBuf* b : lb)
            if (b->p) (void)hipFree(b->p);
        if (L.done) (void)hipEventDestroy(L.done);
        if (L.front_done) (void)hipEventDestroy(L.front_done);
        if (i > 0 && L.stream) (void)hipStreamDestroy(L.stream);
    }
    if (c->fork) (void)hipEventDestroy(c->fork);
    if (c->ag_stream) (void)hipStreamSynchronize(c->ag_stream);
    for (int i = 0; i < gsmcal_ctx::AG_SLOTS; ++i) {
        if (c->ag_ready[i]) (void)hipEventDestroy(c->ag_ready[i]);
        if (c->ag_done[i]) (void)hipEventDestroy(c->ag_done[i]);
    }
    if (c->ag_stream) (void)hipStreamDestroy(c->ag_stream);
    for (int i = 0; i < gsmcal_ctx::GRAPH_SLOTS; ++i)
        for (auto* g : {&c->g_calib[i], &c->g_scan[i]}) {
            if (g->exec) (void)hipGraphExecDestroy(g->exec);
            if (g->graph) (void)hipGraphDestroy(g->graph);
        }
    for (auto& r : c->prof_pending) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int gsmcal_sync(gsmcal_ctx* c) {
    if (!c) return GSMCAL_E_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

const char* gsmcal_last_error(gsmcal_ctx* c) { return c ? c->err.c_str() : "null context"; }

int gsmcal_dev_alloc(gsmcal_ctx* c, size_t bytes, void** dptr) {
    if (!c || !dptr) return GSMCAL_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(dptr, bytes));
    return 0;
}
int gsmcal_dev_free(gsmcal_ctx* c, void* dptr) {
    if (!c) return GSMCAL_E_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(dptr));
    return 0;
}
int gsmcal_memcpy_h2d(gsmcal_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) return GSMCAL_E_ARG;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int gsmcal_memcpy_d2h(gsmcal_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) return GSMCAL_E_ARG;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int gsmcal_profile_enable(gsmcal_ctx* c, int enable) {
    if (!c) return GSMCAL_E_ARG;
    RET_IF(prof_flush(c));
    c->prof = enable != 0;
    if (c->prof && c->ev_pool.size() < 512) {     // event creation is slow: keep it out of the measured launches
        for (int i = 0; i < 512; ++i) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventReleaseToDevice) == hipSuccess) c->ev_pool.push_back(e);
        }
    }
    return 0;
}
int gsmcal_profile_filter(gsmcal_ctx* c, const char* substr) {
    if (!c) return GSMCAL_E_ARG;
    c->prof_filter = substr ? substr : "";
    return 0;
}
int gsmcal_profile_reset(gsmcal_ctx* c) {
    if (!c) return GSMCAL_E_ARG;
    RET_IF(prof_flush(c));
    for (auto& v : c->prof_ms) v = 0.0;
    for (auto& v : c->prof_n) v = 0;
    return 0;
}
int gsmcal_profile_get(gsmcal_ctx* c, int cap, const char** names, double* total_ms, long* launches) {
    if (!c) return GSMCAL_E_ARG;
    RET_IF(prof_flush(c));
    const int n = (int)c->prof_names.size();
    for (int i = 0; i < n && i < cap; ++i) {
        if (names) names[i] = c->prof_names[i].c_str();
        if (total_ms) total_ms[i] = c->prof_ms[i];
        if (launches) launches[i] = c->prof_n[i];
    }
    return n;
}

// ---- a1 raw2iq -----------------------------------------------------------------------------------
int gsmcal_raw2iq_u8(gsmcal_ctx* c, const uint8_t* a, long rows_2n, int d, double* b) {
    if (!c || !a || !b || rows_2n < 2 || (rows_2n & 1) || d < 1) return GSMCAL_E_ARG;
    const long n = rows_2n / 2;
    HIPCHK(c, hipSetDevice(c->device));
    RET_IF(ensure(c, c->misc, (size_t)rows_2n * d));
    RET_IF(ensure(c, c->arr_out, (size_t)n * d * sizeof(cplx)));
    HIPCHK(c, hipMemcpyAsync(c->misc.p, a, (size_t)rows_2n * d, hipMemcpyHostToDevice, c->stream));
    RET_IF(init_states(c, d, n));
    RET_IF(dc_means(c, (const uint8_t*)c->misc.p, d, n));
    int blocks = (int)((n + 256 * 4 - 1) / (256 * 4));
    if (blocks > 2048) blocks = 2048;
    LAUNCH(c, k_raw2iq, dim3(blocks, d), dim3(256), 0, (const uint8_t*)c->misc.p, rows_2n,
           (const StreamState*)c->cur->state.p, (cplx*)c->arr_out.p, n);
    CHECK_LAUNCH(c);
    HIPCHK(c, hipMemcpyAsync(b, c->arr_out.p, (size_t)n * d * sizeof(cplx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int gsmcal_raw2iq(gsmcal_ctx* c, const double* a, long rows_2n, int d, double* b) {
    if (!c || !a || !b || rows_2n < 2 || (rows_2n & 1) || d < 1) return GSMCAL_E_ARG;
    // the doubles hold byte values (fread(...,'uint8'), gsm_sync_demod.m:96): narrow them back
    const size_t tot = (size_t)rows_2n * d;
    std::vector<uint8_t> u(tot);
    for (size_t i = 0; i < tot; ++i) {
        const double v = a[i];
        if (!(v >= 0.0 && v <= 255.0) || v != floor(v)) {
            c->err = "raw2iq: input is not byte-valued (only uint8-valued captures are supported)";
            return GSMCAL_E_UNSUPPORTED;
        }
        u[i] = (uint8_t)v;
    }
    return gsmcal_raw2iq_u8(c, u.data(), rows_2n, d, b);
}

// ---- a2 filters ----------------------------------------------------------------------------------
int gsmcal_filter(gsmcal_ctx* c, const double* coef, int ntaps, const double* s, long n, int d, int decim, double* r) {
    if (!c || !coef || !s || !r || ntaps < 1 || n < 1 || d < 1 || decim < 1) return GSMCAL_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const long nd = (n + decim - 1) / decim;
    RET_IF(upload_cached(c, c->coef, c->h_coef, coef, ntaps));
    RET_IF(upload_array(c, s, (size_t)n * d));
    RET_IF(ensure(c, c->arr_out, (size_t)nd * d * sizeof(cplx)));
    LAUNCH(c, k_fir_arr, dim3((unsigned)((nd + 255) / 256), d), dim3(256), 0, (const cplx*)c->arr_in.p, n, n,
           (const double*)c->coef.p, ntaps, decim, nd, (cplx*)c->arr_out.p, nd);
    CHECK_LAUNCH(c);
    HIPCHK(c, hipMemcpyAsync(r, c->arr_out.p, (size_t)nd * d * sizeof(cplx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int gsmcal_chn_filter_8x_4x(gsmcal_ctx* c, const double* s, long n, int d, const double* num, int ntaps, double* r) {
    if (!num || ntaps <= 0) { num = GSM_CHN_FILTER_8X_NUM; ntaps = 60; }
    return gsmcal_filter(c, num, ntaps, s, n, d, 2, r);   // chn_filter_8x_4x.m:13,15
}

int gsmcal_chn_filter_4x(gsmcal_ctx* c, const double* s, long n, int d, const double* num, int ntaps, double* r) {
    if (!num || ntaps <= 0) { num = GSM_CHN_FILTER_4X_NUM; ntaps = 30; }
    return gsmcal_filter(c, num, ntaps, s, n, d, 1, r);   // chn_filter_4x.m:13: no decimation
}

// ---- a3..a5 coarse detector -----------------------------------------------------------------------
static int coarse_api(gsmcal_ctx* c, const double* s, long len, CoarseArgs a, StreamState* out) {
    HIPCHK(c, hipSetDevice(c->device));
    RET_IF(upload_array(c, s, (size_t)len));
    std::vector<StreamState> v(1);
    host_init_state(v[0], len);
    RET_IF(push_states(c, v));
    c->last_S = 1; c->cur = &c->lanes[0]; c->lanes[0].lo = 0; c->lanes[0].n = 1; c->n_lanes_used = 1;
    a.s = (const cplx*)c->arr_in.p; a.s_stride = len; a.len = len;
    a.th0 = c->params.coarse_th_db; a.min_hits = c->params.min_hits;
    int fft_len = a.fft_len;
    long n_first = len;
    if (a.mode == 0) {
        fft_len = 1 << (int)floor(log2(148.0 / (double)a.decimation_ratio));
        n_first = (long)ceil(23.0 * 1250.0 / (double)a.decimation_ratio);
    }
    if (fft_len < 2 || fft_len > 64) return GSMCAL_E_UNSUPPORTED;
    const long nwin = n_first - (fft_len - 1);
    const size_t lds = coarse_scan_lds(n_first > 0 ? n_first : 0, a.mode == 0 ? 10 * fft_len : a.mv_len);
    if (lds > 159 * 1024) return GSMCAL_E_UNSUPPORTED;
    if (a.mode != 2 && nwin >= 1 && n_first <= len) {
        RET_IF(ensure(c, c->cur->snrbuf, (size_t)nwin * sizeof(double)));
        a.snr_g = (double*)c->cur->snrbuf.p; a.snr_stride = nwin;
        if (fft_len == 16) LAUNCH(c, k_coarse_snr<true>, dim3((unsigned)((nwin + 255) / 256), 1), dim3(256), 0, a);
        else LAUNCH(c, k_coarse_snr<false>, dim3((unsigned)((nwin + 255) / 256), 1), dim3(256), 0, a);
    }
    if (fft_len == 16) LAUNCH(c, k_coarse_scan_lat, dim3(1), dim3(256), lds, (StreamState*)c->cur->state.p, a);
    else LAUNCH(c, k_coarse_scan_gen, dim3(1), dim3(256), lds, (StreamState*)c->cur->state.p, a);
    CHECK_LAUNCH(c);
    RET_IF(fetch_states(c, 1, v));
    *out = v[0];
    return 0;
}

int gsmcal_move_fft_snr_runtime_avg(gsmcal_ctx* c, const double* s, long len, int mv_len, int fft_len, double th,
                                    int* hit_flag, double* hit_idx, double* hit_avg_snr, double* hit_snr) {
    if (!c || !s || len < 1 || mv_len < 1 || fft_len < 2) return GSMCAL_E_ARG;
    CoarseArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = 1; a.mv_len = mv_len; a.fft_len = fft_len; a.th = th; a.decimation_ratio = 8;
    StreamState st;
    RET_IF(coarse_api(c, s, len, a, &st));
    if (st.status < 0) return st.status;
    if (hit_flag) *hit_flag = st.coarse_hit_flag;
    if (hit_idx) *hit_idx = st.coarse_hit_flag ? st.mv_hit_idx : -1.0;
    if (hit_avg_snr) *hit_avg_snr = st.coarse_hit_flag ? st.hit_avg_snr : INFINITY;
    if (hit_snr) *hit_snr = st.coarse_hit_flag ? st.mv_hit_snr : INFINITY;
    return 0;
}

int gsmcal_specific_fft_snr_fix_avg(gsmcal_ctx* c, const double* s, long len, const double target_set[2], int fft_len,
                                    double th, double avg_snr, int* hit_flag, double* hit_idx, double* hit_snr) {
    if (!c || !s || !target_set || len < 1 || fft_len < 2) return GSMCAL_E_ARG;
    CoarseArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = 2; a.fft_len = fft_len; a.th = th; a.avg_snr = avg_snr; a.decimation_ratio = 8; a.mv_len = 1;
    a.t_lo = (long)target_set[0]; a.t_hi = (long)target_set[1];
    StreamState st;
    RET_IF(coarse_api(c, s, len, a, &st));
    if (st.status < 0) return st.status;
    if (hit_flag) *hit_flag = st.coarse_hit_flag;
    if (hit_idx) *hit_idx = st.coarse_hit_flag ? st.mv_hit_idx : -1.0;
    if (hit_snr) *hit_snr = st.coarse_hit_flag ? st.mv_hit_snr : INFINITY;
    return 0;
}

int gsmcal_FCCH_coarse_position(gsmcal_ctx* c, const double* s, long len, int decimation_ratio, double* position,
                                double* snr, int cap, int* count) {
    if (!c || !s || !position || !snr || !count || len < 1 || decimation_ratio < 1 || cap < 1) return GSMCAL_E_ARG;
    if (hits_capacity(len, decimation_ratio) > MAXH) return GSMCAL_E_CAPACITY;
    CoarseArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = 0; a.decimation_ratio = decimation_ratio;
    StreamState st;
    RET_IF(coarse_api(c, s, len, a, &st));
    if (st.status < 0) return st.status;
    if (st.n_coarse == 0) {
        position[0] = -1.0; snr[0] = -1.0; *count = 1;
        return GSMCAL_S_NO_FCCH;
    }
    if (st.n_coarse > cap) return GSMCAL_E_CAPACITY;
    for (int i = 0; i < st.n_coarse; ++i) { position[i] = st.coarse_pos[i]; snr[i] = st.coarse_snr[i]; }
    *count = st.n_coarse;
    return 0;
}

// ---- a6 FCCH_fine_correction ------------------------------------------------------------------------
int gsmcal_FCCH_fine_correction(gsmcal_ctx* c, const double* s, long len, const double* base_position, int num_base,
                                int ov, double carrier_freq, double* fcch_pos, int cap_pos, int* num_pos, double* r,
                                long cap_r, long* len_r, double* sampling_ppm, double* carrier_ppm) {
    if (!c || !s || !base_position || !fcch_pos || !num_pos || len < 1 || num_base < 0 || ov < 1 || cap_pos < 1)
        return GSMCAL_E_ARG;
    if (num_base > MAXH) return GSMCAL_E_CAPACITY;
    HIPCHK(c, hipSetDevice(c->device));
    const Geom g(ov);
    RET_IF(upload_array(c, s, (size_t)len));
    RET_IF(upload_cached(c, c->cf, c->h_cf, &carrier_freq, 1));
    std::vector<StreamState> v(1);
    host_init_state(v[0], len);
    v[0].n_coarse = num_base;
    for (int i = 0; i < num_base; ++i) v[0].coarse_pos[i] = base_position[i];
    RET_IF(push_states(c, v));
    c->last_S = 1; c->cur = &c->lanes[0]; c->lanes[0].lo = 0; c->lanes[0].n = 1; c->n_lanes_used = 1;
    Source src{SRC_ARR, nullptr, 0, (const cplx*)c->arr_in.p, len, nullptr, 0};
    const int H = num_base > 0 ? num_base : 1;
    RET_IF(run_fine(c, 1, src, 0, g, H, false, -1, 0));
    RET_IF(fetch_states(c, 1, v));
    const StreamState& st = v[0];
    if (st.status < 0) return st.status;
    if (sampling_ppm) *sampling_ppm = st.sampling_ppm1;
    if (carrier_ppm) *carrier_ppm = st.carrier_ppm1;
    if (st.fcch_is_sentinel) {
        fcch_pos[0] = -1.0;
        *num_pos = 1;
    } else {
        if (st.n_fcch > cap_pos) return GSMCAL_E_CAPACITY;
        for (int i = 0; i < st.n_fcch; ++i) fcch_pos[i] = st.fcch_pos[i];
        *num_pos = st.n_fcch;
    }
    long lr = -1;
    int level = 0;
    if (st.r1_kind == 1) { lr = len; level = 0; }
    else if (st.r1_kind == 2) { lr = st.op[1].n; level = 1; }
    else if (st.r1_kind == 3) { lr = st.op[2].n; level = 2; }
    if (len_r) *len_r = lr;
    if (r && lr > 0) {
        if (lr > cap_r) return GSMCAL_E_CAPACITY;
        if (level == 0) memcpy(r, s, (size_t)lr * sizeof(cplx));
        else RET_IF(materialise_to_host(c, src, level, lr, r));
    }
    return positive_status(st, 0);
}

// ---- a7 SCH_corr_rate_correction ----------------------------------------------------------------------
int gsmcal_SCH_corr_rate_correction(gsmcal_ctx* c, const double* s, long len, const double* fcch_pos, int num_fcch,
                                    const double* sch_ts, int len_ts, int ov, double* pos_info, int cap_rows,
                                    int* num_rows, double* r, long cap_r, long* len_r, double* sampling_ppm) {
    if (!c || !fcch_pos || !sch_ts || !pos_info || !num_rows || num_fcch < 0 || len_ts < 1 || ov < 1 || cap_rows < 1)
        return GSMCAL_E_ARG;
    if (num_fcch > MAXH) return GSMCAL_E_CAPACITY;
    HIPCHK(c, hipSetDevice(c->device));
    const Geom g(ov);
    const bool have_s = s != nullptr && len >= 1;   // r = -1 from a failed fine stage arrives as s = NULL
    if (have_s) RET_IF(upload_array(c, s, (size_t)len));
    RET_IF(upload_cached(c, c->ts, c->h_ts, sch_ts, (size_t)2 * len_ts));
    std::vector<StreamState> v(1);
    host_init_state(v[0], have_s ? len : 0);
    const bool sentinel_in = (num_fcch == 1 && fcch_pos[0] == -1.0);
    v[0].fcch_is_sentinel = sentinel_in ? 1 : 0;
    v[0].n_fcch = sentinel_in ? 0 : num_fcch;
    for (int i = 0; i < num_fcch; ++i) v[0].fcch_pos[i] = fcch_pos[i];
    if (!have_s && !(sentinel_in || num_fcch < 5)) return GSMCAL_E_ARG;
    RET_IF(push_states(c, v));
    c->last_S = 1; c->cur = &c->lanes[0]; c->lanes[0].lo = 0; c->lanes[0].n = 1; c->n_lanes_used = 1;
    Source src{SRC_ARR, nullptr, 0, (const cplx*)c->arr_in.p, len, nullptr, 0};
    const int H = num_fcch > 0 ? num_fcch : 1;
    RET_IF(run_sch(c, 1, src, 0, g, H, len_ts, false, -1));
    RET_IF(fetch_states(c, 1, v));
    const StreamState& st = v[0];
    if (st.status < 0) return st.status;
    if (sampling_ppm) *sampling_ppm = st.sampling_ppm2;
    if (st.n_rows == 0) {
        // the reference's all -1 sentinel keeps the shape of the exit taken: [-1 -1] (:9, :61) or the -ones(3K,2)
        // pre-allocation of :32 (fewer than 5 SCH :84, spacing failure :106-112) -- gsm_sync_demod.m:130 counts its rows
        const int nr = st.n_sent_rows > 0 ? st.n_sent_rows : 1;
        if (nr > cap_rows) return GSMCAL_E_CAPACITY;
        for (int i = 0; i < nr; ++i) { pos_info[i] = -1.0; pos_info[cap_rows + i] = -1.0; }
        *num_rows = nr;
    } else {
        if (st.n_rows > cap_rows) return GSMCAL_E_CAPACITY;
        for (int i = 0; i < st.n_rows; ++i) {
            pos_info[i] = st.pos_info[i];
            pos_info[cap_rows + i] = st.pos_info[MAXROWS + i];
        }
        *num_rows = st.n_rows;
    }
    long lr = -1;
    int level = 0;
    if (st.r2_kind == 1) { lr = len; level = 0; }
    else if (st.r2_kind == 2) { lr = st.op[1].n; level = st.op[1].type == OP_COPY ? 0 : 1; }
    if (len_r) *len_r = lr;
    if (r && lr > 0) {
        if (lr > cap_r) return GSMCAL_E_CAPACITY;
        if (level == 0) memcpy(r, s, (size_t)lr * sizeof(cplx));
        else RET_IF(materialise_to_host(c, src, level, lr, r));
    }
    return positive_status(st, 1);
}

// ---- a8 carrier_correct_post_SCH -------------------------------------------------------------------------
int gsmcal_carrier_correct_post_SCH(gsmcal_ctx* c, const double* s, long len, const double* pos_info, int rows, int ld,
                                    int ov, double carrier_freq, double* r, long cap_r, long* len_r, double* carrier_ppm) {
    if (!c || !pos_info || rows < 1 || ld < rows || ov < 1) return GSMCAL_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const Geom g(ov);
    bool all_m1 = true;                       // `if pos_info == -1` is true only if every element is -1
    for (int i = 0; i < rows; ++i) all_m1 = all_m1 && pos_info[i] == -1.0 && pos_info[ld + i] == -1.0;
    if (!all_m1 && rows > MAXROWS) return GSMCAL_E_CAPACITY;
    const bool have_s = s != nullptr && len >= 1;
    if (have_s) RET_IF(upload_array(c, s, (size_t)len));
    RET_IF(upload_cached(c, c->cf, c->h_cf, &carrier_freq, 1));
    std::vector<StreamState> v(1);
    host_init_state(v[0], have_s ? len : 0);
    int nfcch = 0;
    if (!all_m1) {
        v[0].n_rows = rows;
        for (int i = 0; i < rows; ++i) {
            v[0].pos_info[i] = pos_info[i];
            v[0].pos_info[MAXROWS + i] = pos_info[ld + i];
            nfcch += pos_info[ld + i] == 0.0;
        }
        if (!have_s) return GSMCAL_E_ARG;
    }
    if (nfcch > MAXH) return GSMCAL_E_CAPACITY;
    RET_IF(push_states(c, v));
    c->last_S = 1; c->cur = &c->lanes[0]; c->lanes[0].lo = 0; c->lanes[0].n = 1; c->n_lanes_used = 1;
    Source src{SRC_ARR, nullptr, 0, (const cplx*)c->arr_in.p, len, nullptr, 0};
    RET_IF(run_post(c, 1, src, 0, g, nfcch > 0 ? nfcch : 1, false, nullptr, nullptr, nullptr));
    RET_IF(fetch_states(c, 1, v));
    const StreamState& st = v[0];
    if (st.status < 0) return st.status;
    if (carrier_ppm) *carrier_ppm = st.carrier_ppm2;
    long lr = st.r3_kind == 3 ? st.op[1].n : -1;
    if (len_r) *len_r = lr;
    if (r && lr > 0) {
        if (lr > cap_r) return GSMCAL_E_CAPACITY;
        RET_IF(materialise_to_host(c, src, 1, lr, r));
    }
    return positive_status(st, 2);
}

// ---- f4 SCH demodulator front end -------------------------------------------------------------------------
int gsmcal_SCH_equalise(gsmcal_ctx* c, const double* s, long len, const double* pos_info, int rows, int ld, const double* sch_ts,
                        int len_ts, int ov, double* x_eq, int cap_bursts, int* num_bursts, int* len_fde_ov) {
    if (!c || !pos_info || !sch_ts || !num_bursts || rows < 1 || ld < rows || ov < 1 || len_ts < 1 || cap_bursts < 0) return GSMCAL_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int L = (148 + 2 * 8 + 30) * ov, N2 = L / DM_N1;      // SCH_demod.m:22,45,53-55: round(156.25 - 8.25) + 2*8 + 30 symbols
    const int sp_t0 = (8 + 42) * ov;                            // :56 sp_of_training (0-based)
    *num_bursts = 0;
    if (len_fde_ov) *len_fde_ov = L;
    bool all_m1 = true;                                         // :8 `if pos_info == -1`: every element
    for (int i = 0; i < rows; ++i) all_m1 = all_m1 && pos_info[i] == -1.0 && pos_info[ld + i] == -1.0;
    if (all_m1) return GSMCAL_S_POST_NO_POS;
    if (!s || len < 1 || !x_eq) return GSMCAL_E_ARG;
    if (sp_t0 + len_ts > L) return GSMCAL_E_ARG;                // the training sequence must fit the window (:58)
    std::vector<long> starts;
    for (int i = 0; i < rows; ++i)
        if (pos_info[ld + i] == 1.0) starts.push_back((long)pos_info[i] - 8L * ov - 1);   // :13-14, :79 (0-based)
    const int nb = (int)starts.size();
    if (nb == 0) return 0;
    if (nb > cap_bursts) return GSMCAL_E_CAPACITY;
    const size_t lds = dm_lds_bytes(L, N2);
    if (lds > 159 * 1024) return GSMCAL_E_UNSUPPORTED;
    c->cur = &c->lanes[0];
    RET_IF(upload_array(c, s, (size_t)len));
    RET_IF(upload_cached(c, c->ts, c->h_ts, sch_ts, (size_t)2 * len_ts));
    if (c->tw_sch_n != L) {
        RET_IF(ensure(c, c->tw_sch, (size_t)L * sizeof(cplx)));
        LAUNCH(c, k_make_twiddles, dim3((L + 255) / 256), dim3(256), 0, (cplx*)c->tw_sch.p, L);
        c->tw_sch_n = L;
    }
    RET_IF(ensure(c, c->misc, (size_t)nb * (sizeof(long) + sizeof(int)) + (size_t)L * sizeof(cplx) + 64));
    cplx* d_ft = (cplx*)c->misc.p;
    long* d_start = (long*)(d_ft + L);
    int* d_status = (int*)(d_start + nb);
    RET_IF(ensure(c, c->arr_out, (size_t)nb * L * sizeof(cplx)));
    HIPCHK(c, hipMemcpyAsync(d_start, starts.data(), (size_t)nb * sizeof(long), hipMemcpyHostToDevice, c->stream));
    LAUNCH(c, k_sch_fd_training, dim3(1), dim3(DM_THREADS), lds, (const cplx*)c->ts.p, len_ts, sp_t0, L, N2, (const cplx*)c->tw_sch.p, d_ft);
    LAUNCH(c, k_sch_equalise, dim3(nb), dim3(DM_THREADS), lds, (const cplx*)c->arr_in.p, len, (const long*)d_start, len_ts, sp_t0, L, N2,
           (const cplx*)c->tw_sch.p, (const cplx*)d_ft, (cplx*)c->arr_out.p, d_status);
    CHECK_LAUNCH(c);
    std::vector<int> st(nb);
    HIPCHK(c, hipMemcpyAsync(st.data(), d_status, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(x_eq, c->arr_out.p, (size_t)nb * L * sizeof(cplx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < nb; ++i)
        if (st[i] != 0) return st[i];                           // MATLAB: index exceeds matrix dimensions at s(sp:ep), :81
    *num_bursts = nb;
    return 0;
}

// ---- a9 total_ppm_calculation ---------------------------------------------------------------------------
int gsmcal_total_ppm_calculation(const double* ppm_in, int n, double* ppm_out) {
    if (!ppm_in || !ppm_out || n < 1) return GSMCAL_E_ARG;
    bool all_inf = true;
    for (int i = 0; i < n; ++i) all_inf = all_inf && ppm_in[i] == INFINITY;
    if (all_inf) { *ppm_out = INFINITY; return GSMCAL_S_ALL_INF; }   // :7-11
    double p = 1.0;
    for (int i = 0; i < n; ++i) p = p * (1.0 + ppm_in[i] * 1e-6);     // :14-18
    *ppm_out = (p - 1.0) * 1e6;                                        // :20-21
    return 0;
}

// ---- batched hot path ---------------------------------------------------------------------------------------
int gsmcal_frontend_batch_dev(gsmcal_ctx* c, const uint8_t* d_raw, int d, long n, const double* coef, int ntaps,
                              int decim, double* d_out) {
    if (!c || !d_raw || !coef || !d_out || d < 1 || n < 1 || ntaps < 1 || decim < 1) return GSMCAL_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    c->cur = &c->lanes[0];
    c->lanes[0].lo = 0; c->lanes[0].n = d; c->n_lanes_used = 1;
    RET_IF(upload_cached(c, c->coef, c->h_coef, coef, ntaps));
    RET_IF(init_states(c, d, n));
    RET_IF(dc_means(c, d_raw, d, n));
    const long nd = (n + decim - 1) / decim;
    return fir_decim_raw(c, d_raw, d, n, (const double*)c->coef.p, ntaps, decim, (cplx*)d_out, nd);
}

int gsmcal_frontend_batch(gsmcal_ctx* c, const uint8_t* raw, int d, long n, const double* coef, int ntaps, int decim,
                          double* out) {
    if (!c || !raw || !out || d < 1 || n < 1 || decim < 1) return GSMCAL_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const long nd = (n + decim - 1) / decim;
    RET_IF(ensure(c, c->misc, (size_t)2 * n * d));
    RET_IF(ensure(c, c->arr_out, (size_t)nd * d * sizeof(cplx)));
    HIPCHK(c, hipMemcpyAsync(c->misc.p, raw, (size_t)2 * n * d, hipMemcpyHostToDevice, c->stream));
    RET_IF(gsmcal_frontend_batch_dev(c, (const uint8_t*)c->misc.p, d, n, coef, ntaps, decim, (double*)c->arr_out.p));
    HIPCHK(c, hipMemcpyAsync(out, c->arr_out.p, (size_t)nd * d * sizeof(cplx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int gsmcal_fcch_scan_batch_dev(gsmcal_ctx* c, const uint8_t* d_raw, int d, long n, const double* coef, int ntaps,
                               double* d_snr_numhit, double* d_positions, double* d_pos_snr, int* d_counts) {
    if (!c || !d_raw || !coef || !d_snr_numhit || d < 1 || n < 1 || ntaps < 1) return GSMCAL_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int ov = 8, dec_ratio = 8, decim = ov * dec_ratio;   // ..FCCH_scanner.m:43-45
    const long nd = (n + decim - 1) / decim;
    if (hits_capacity(nd, dec_ratio) > MAXH) return GSMCAL_E_CAPACITY;
    if (nd < (long)ceil(23.0 * 1250.0 / (double)dec_ratio)) {   // FCCH_coarse_position.m:25 s(1:ceil(23 frames)): MATLAB index error
        c->err = "capture shorter than 23 frames after decimation (FCCH_coarse_position.m:25 would index past the end)";
        return GSMCAL_E_INDEX;
    }
    c->cur = &c->lanes[0];
    c->call_raw_bytes = (size_t)d * 2 * (size_t)n;
    RET_IF(upload_cached(c, c->coef, c->h_coef, coef, ntaps));
    RET_IF(ensure_head(c, decim));
    const std::vector<uintptr_t> key = {(uintptr_t)d_raw, (uintptr_t)d, (uintptr_t)n, (uintptr_t)ntaps,
                                        (uintptr_t)d_snr_numhit, (uintptr_t)d_positions, (uintptr_t)d_pos_snr,
                                        (uintptr_t)d_counts, (uintptr_t)c->n_lanes_cfg, (uintptr_t)c->params_epoch};
    auto enqueue = [&]() -> int {
    const int nl = plan_lanes(c, d, false);
    RET_IF(fork_lanes(c, nl));
    for (int i = 0; i < nl; ++i) {
        Lane& L = c->lanes[i];
        c->cur = &L;
        const int lo = L.lo, S = L.n;
        const uint8_t* raw_i = d_raw + (size_t)lo * 2 * n;
        RET_IF(ensure(c, L.dec, (size_t)S * nd * sizeof(cplx)));
        if (nl > 1) {                                       // pipeline: this front kernel starts when the previous stage's has finished
            if (!L.front_done) HIPCHK(c, hipEventCreateWithFlags(&L.front_done, hipEventDisableTiming));
            if (i > 0) HIPCHK(c, hipStreamWaitEvent(L.stream, c->lanes[i - 1].front_done, 0));
        }
        // The next stage's front kernel waits for this one through an event, and that hand-over leaves the memory system idle for
        // ~12 us per stage (rocprofv3 timeline).  So the stage's front kernel is launched in two parts: the event sits behind the
        // first (GSMCAL_SCAN_SPLIT percent of the captures), and the rest runs on this lane underneath the start of the next stage.
        const int S_a = nl > 1 && i + 1 < nl && c->scan_split > 0 && c->scan_split < 100 ? std::max(1, (int)((long)S * c->scan_split / 100)) : S;
        RET_IF(front_fused(c, raw_i, S_a, n, (const double*)c->coef.p, ntaps, decim, (cplx*)L.dec.p, nd, 0, S));
        if (nl > 1) HIPCHK(c, hipEventRecord(L.front_done, L.stream));
        if (S_a < S) RET_IF(front_fused(c, raw_i, S - S_a, n, (const double*)c->coef.p, ntaps, decim, (cplx*)L.dec.p, nd, S_a, S));
        // the acceptance rule (multi_rtl_sdr_gsm_FCCH_scanner.m:168-185) runs at the end of k_coarse_scan, on the state it just built
        ScanAccept acc;
        acc.snr_numhit = d_snr_numhit + (size_t)2 * lo;
        acc.positions = d_positions ? d_positions + (size_t)lo * MAXH : nullptr;
        acc.pos_snr = d_pos_snr ? d_pos_snr + (size_t)lo * MAXH : nullptr;
        acc.counts = d_counts ? d_counts + lo : nullptr;
        RET_IF(coarse(c, S, (const cplx*)L.dec.p, nd, nd, dec_ratio, 0, true, n, decim, &acc, nl == 1 || (c->snr_inline_pipe && S <= 3 * c->n_cu)));
        CHECK_LAUNCH(c);
    }
    RET_IF(join_lanes(c, nl));
    return 0;
    };
    RET_IF(run_maybe_graph(c, pick_slot(c, c->g_scan, key), key, enqueue, plan_lanes(c, d, false) > 1));
    plan_lanes(c, d, false);
    c->cur = &c->lanes[0];
    c->last_S = d;
    return 0;
}

int gsmcal_fcch_scan_batch(gsmcal_ctx* c, const uint8_t* raw, int d, long n, const double* coef, int ntaps, double* snr,
                           double* num_hit, double* positions, double* pos_snr, int* counts) {
    if (!c || !raw || !snr || !num_hit || d < 1 || n < 1) return GSMCAL_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    RET_IF(ensure(c, c->misc, (size_t)2 * n * d));
    RET_IF(ensure(c, c->snrhit, (size_t)d * (2 + 2 * MAXH) * sizeof(double) + (size_t)d * sizeof(int)));
    HIPCHK(c, hipMemcpyAsync(c->misc.p, raw, (size_t)2 * n * d, hipMemcpyHostToDevice, c->stream));
    double* d_sn = (double*)c->snrhit.p;
    double* d_pos = d_sn + (size_t)2 * d;
    double* d_ps = d_pos + (size_t)d * MAXH;
    int* d_cnt = (int*)(d_ps + (size_t)d * MAXH);
    RET_IF(gsmcal_fcch_scan_batch_dev(c, (const uint8_t*)c->misc.p, d, n, coef, ntaps, d_sn, d_pos, d_ps, d_cnt));
    std::vector<double> sn((size_t)2 * d);
    HIPCHK(c, hipMemcpyAsync(sn.data(), d_sn, sn.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (positions) HIPCHK(c, hipMemcpyAsync(positions, d_pos, (size_t)d * MAXH * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (pos_snr) HIPCHK(c, hipMemcpyAsync(pos_snr, d_ps, (size_t)d * MAXH * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (counts) HIPCHK(c, hipMemcpyAsync(counts, d_cnt, (size_t)d * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < d; ++i) { snr[i] = sn[2 * i]; num_hit[i] = sn[2 * i + 1]; }
    return 0;
}

int gsmcal_calibrate_batch_dev(gsmcal_ctx* c, const uint8_t* d_raw, int d, long n, const double* coef, int ntaps,
                               const double* sch_ts, int len_ts, const double* carrier_freq, double* d_table,
                               double* d_pos_info, double* d_r_correct, long* d_r_len) {
    if (!c || !d_raw || !coef || !sch_ts || !carrier_freq || !d_table || d < 1 || n < 1 || ntaps < 1 || len_ts < 1)
        return GSMCAL_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int ov = 8, dec_ratio = 8, decim = ov * dec_ratio;   // gsm_sync_demod.m:17-19
    const Geom g(ov);
    const long nd = (n + decim - 1) / decim;
    int H = hits_capacity(nd, dec_ratio) + 1;
    if (H > MAXH) return GSMCAL_E_CAPACITY;
    c->cur = &c->lanes[0];
    c->call_raw_bytes = (size_t)d * 2 * (size_t)n;
    RET_IF(upload_cached(c, c->coef, c->h_coef, coef, ntaps));
    RET_IF(upload_cached(c, c->ts, c->h_ts, sch_ts, (size_t)2 * len_ts));
    RET_IF(upload_cached(c, c->cf, c->h_cf, carrier_freq, d));
    RET_IF(ensure_head(c, decim));
    RET_IF(ensure_twiddles(c, g.nfft));
    // independent streams: split over lanes (HIP streams) so latency-bound stages of one group overlap the
    // compute-bound fine search of another; a repeated call is replayed as one hipGraph
    const std::vector<uintptr_t> key = {(uintptr_t)d_raw, (uintptr_t)d, (uintptr_t)n, (uintptr_t)ntaps, (uintptr_t)len_ts,
                                        (uintptr_t)d_table, (uintptr_t)d_pos_info, (uintptr_t)d_r_correct,
                                        (uintptr_t)d_r_len, (uintptr_t)c->n_lanes_cfg, (uintptr_t)c->params_epoch};
    auto enqueue = [&]() -> int {
    const int nl = plan_lanes(c, d);
    RET_IF(fork_lanes(c, nl));
    const double* cf_all = (const double*)c->cf.p;
    for (int i = 0; i < nl; ++i) {
        Lane& L = c->lanes[i];
        c->cur = &L;
        const int lo = L.lo, S = L.n;
        const uint8_t* raw_i = d_raw + (size_t)lo * 2 * n;
        RET_IF(ensure(c, L.dec, (size_t)S * nd * sizeof(cplx)));
        // staggered lanes: this lane's front kernel starts when the previous lane's has finished -- the bandwidth-bound front kernels
        // then follow one another instead of competing, and each runs beside the compute-bound stages of the lanes ahead of it.
        // Measured (round 4, tools/s10.sh): 128 / 256 / 512 / 1 024 / 2 048 streams 0.345 / 0.562 / 1.002 / 1.807 / 3.629 ms staggered
        // against 0.349 / 0.554 / 1.002 / 1.874 / 3.715 together: worth it from 256 streams per lane on.
        const bool stagger = nl > 1 && (c->lane_stagger == 1 || (c->lane_stagger < 0 && d / nl >= 256));   // (one decision for all lanes of the call)
        if (stagger) {
            if (!L.front_done) HIPCHK(c, hipEventCreateWithFlags(&L.front_done, hipEventDisableTiming));
            if (i > 0) HIPCHK(c, hipStreamWaitEvent(L.stream, c->lanes[i - 1].front_done, 0));
        }
        RET_IF(front_fused(c, raw_i, S, n, (const double*)c->coef.p, ntaps, decim, (cplx*)L.dec.p, nd));  // :107,110,117
        if (stagger) HIPCHK(c, hipEventRecord(L.front_done, L.stream));
        RET_IF(coarse(c, S, (const cplx*)L.dec.p, nd, nd, dec_ratio, ov, true, n, decim));         // :117 (+ state init, fine setup)
        Source src{SRC_RAW, raw_i, 2 * n, nullptr, 0, (const double*)c->coef.p, ntaps};
        c->cf_lane = cf_all + lo;
        ChainOut co{d_table + (size_t)lo * GSMCAL_TABLE_COLS, d_pos_info ? d_pos_info + (size_t)lo * 2 * MAXROWS : nullptr,
                    d_r_len ? d_r_len + lo : nullptr, false};
        RET_IF(run_fine(c, S, src, 0, g, H, true, 2, len_ts, &co));                         // :118 (+ SCH window setup; fused: :118-124)
        if (!co.fused) {
            RET_IF(run_sch(c, S, src, 2, g, H, len_ts, true, 3));                           // :119 (+ post-SCH window setup)
            RET_IF(run_post(c, S, src, 3, g, H, true, co.table, co.pos_info_out, co.r_len_out));   // :120, :123-124
        }
        if (d_r_correct) {
            StreamTileArgs ta;
            ta.raw = raw_i; ta.raw_stride = 2 * n; ta.coef = (const double*)c->coef.p; ta.ntaps = ntaps;
            ta.dst = (cplx*)d_r_correct + (size_t)lo * n; ta.dst_stream_stride = n;
            const size_t tlds = stream_tile_lds(ntaps);
            bool sym = (int)c->h_coef.size() == ntaps;         // exactly mirrored taps (what fir1 returns)
            for (int k = 0; sym && k < ntaps / 2; ++k) sym = c->h_coef[k] == c->h_coef[ntaps - 1 - k];
            if (ntaps == 47 && sym && c->stream_s47) {          // the drivers' filter: taps in registers, every sample read once
                LAUNCH(c, k_stream_tile_s47<ST47_TILE>, dim3((unsigned)((n + (long)ST47_TILE * ST_TPB - 1) / ((long)ST47_TILE * ST_TPB)), S), dim3(ST_THREADS), stream_tile_s47_lds(), (const StreamState*)L.state.p, ta);
                CHECK_LAUNCH(c);
            } else if (tlds <= 64 * 1024) {
                LAUNCH_GEOM(ta.ntaps == 47, c, (k_stream_tile<47>), (k_stream_tile<0>), dim3((unsigned)((n + (long)ST_TILE * ST_TPB - 1) / ((long)ST_TILE * ST_TPB)), S), dim3(ST_THREADS), tlds, (const StreamState*)L.state.p, ta);
                CHECK_LAUNCH(c);
            } else {                                        // very long filters: the general tile gather
                const int tiles = (int)((n + TILE - 1) / TILE);
                RET_IF(launch_gather(c, S, src, 4, TILE, true, tiles, (cplx*)d_r_correct + (size_t)lo * n, n, 0));
            }
        }
    }
    c->cf_lane = nullptr;
    RET_IF(join_lanes(c, nl));
    return 0;
    };
    RET_IF(run_maybe_graph(c, pick_slot(c, c->g_calib, key), key, enqueue, plan_lanes(c, d) > 1));
    plan_lanes(c, d);          // lane bookkeeping for gsmcal_last_batch_details (a replay does not run `enqueue`)
    c->cur = &c->lanes[0];
    c->last_S = d;
    return 0;
}

int gsmcal_calibrate_batch(gsmcal_ctx* c, const uint8_t* raw, int d, long n, const double* coef, int ntaps,
                           const double* sch_ts, int len_ts, const double* carrier_freq, double* table,
                           double* pos_info, double* r_correct, long* r_len) {
    if (!c || !raw || !table || d < 1 || n < 1) return GSMCAL_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    RET_IF(ensure(c, c->misc, (size_t)2 * n * d));
    RET_IF(ensure(c, c->table, (size_t)d * GSMCAL_TABLE_COLS * sizeof(double)));
    RET_IF(ensure(c, c->posinfo, (size_t)d * 2 * MAXROWS * sizeof(double)));
    RET_IF(ensure(c, c->rlen, (size_t)d * sizeof(long)));
    if (r_correct) RET_IF(ensure(c, c->arr_out, (size_t)d * n * sizeof(cplx)));
    HIPCHK(c, hipMemcpyAsync(c->misc.p, raw, (size_t)2 * n * d, hipMemcpyHostToDevice, c->stream));
    RET_IF(gsmcal_calibrate_batch_dev(c, (const uint8_t*)c->misc.p, d, n, coef, ntaps, sch_ts, len_ts, carrier_freq,
                                      (double*)c->table.p, (double*)c->posinfo.p,
                                      r_correct ? (double*)c->arr_out.p : nullptr, (long*)c->rlen.p));
    HIPCHK(c, hipMemcpyAsync(table, c->table.p, (size_t)d * GSMCAL_TABLE_COLS * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (pos_info)
        HIPCHK(c, hipMemcpyAsync(pos_info, c->posinfo.p, (size_t)d * 2 * MAXROWS * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (r_len) HIPCHK(c, hipMemcpyAsync(r_len, c->rlen.p, (size_t)d * sizeof(long), hipMemcpyDeviceToHost, c->stream));
    if (r_correct)
        HIPCHK(c, hipMemcpyAsync(r_correct, c->arr_out.p, (size_t)d * n * sizeof(cplx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ---- multi-GPU: RCCL all-gather of the result table ------------------------------------------------------------
// librccl.so is loaded on first use, so single-GPU users of libgsmcal.so do not depend on it.
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi* rccl_api() {
    static RcclApi api;
    if (api.h) return &api;
    // RCCL must belong to the HIP runtime this process runs on: a process whose runtime is the copy bundled with a
    // PyTorch-ROCm wheel and whose RCCL is the system one works until exit and then aborts in the allocator (double free).
    // So: a librccl that is mapped already; else the one lying beside the loaded libamdhip64; else the loader's choice.
    // (RTLD_NODELETE throughout: RCCL registers exit-time clean-up of its own; a process that unloads the library before that
    // runs -- a Python interpreter tearing down its ctypes handles in no particular order -- ends in the allocator with
    // "double free or corruption" after all work is done and checked)
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD | RTLD_NODELETE);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD | RTLD_NODELETE);
    if (!h) {
        Dl_info di;
        if (dladdr((void*)&hipGetDeviceCount, &di) && di.dli_fname) {
            std::string dir(di.dli_fname);
            const size_t cut = dir.rfind('/');
            if (cut != std::string::npos) {
                dir.resize(cut + 1);
                h = dlopen((dir + "librccl.so").c_str(), RTLD_NOW | RTLD_GLOBAL | RTLD_NODELETE);
                if (!h) h = dlopen((dir + "librccl.so.1").c_str(), RTLD_NOW | RTLD_GLOBAL | RTLD_NODELETE);
            }
        }
    }
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NODELETE);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NODELETE);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NODELETE);
    if (!h) return nullptr;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
    api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy) return nullptr;
    api.h = h;
    return &api;
}
struct gsmcal_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
};
static_assert(GSMCAL_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");

int gsmcal_comm_get_unique_id(void* id_out) {
    if (!id_out) return GSMCAL_E_ARG;
    RcclApi* a = rccl_api();
    if (!a) return GSMCAL_E_UNSUPPORTED;
    ncclUniqueId id;
    if (a->GetUniqueId(&id) != ncclSuccess) return GSMCAL_E_HIP;
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int gsmcal_comm_init_rank(gsmcal_ctx* c, const void* idp, int world, int rank, gsmcal_comm** out) {
    if (!c || !idp || !out || world < 1 || rank < 0 || rank >= world) return GSMCAL_E_ARG;
    *out = nullptr;
    RcclApi* a = rccl_api();
    if (!a) { c->err = "librccl.so could not be loaded"; return GSMCAL_E_UNSUPPORTED; }
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, idp, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = a->CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) {
        c->err = std::string("ncclCommInitRank: ") + (a->GetErrorString ? a->GetErrorString(r) : "failed");
        return GSMCAL_E_HIP;
    }
    gsmcal_comm* g = new gsmcal_comm();
    g->comm = comm; g->world = world; g->rank = rank;
    *out = g;
    return 0;
}

// ---- id-file bootstrap: [8 B magic | 8 B nonce | 128 B id], run-specific (see include/gsmcal.h) ----
static const unsigned long long GSMCAL_ID_MAGIC = 0x3144494c41434d47ull;   // "GMCALID1"

int gsmcal_comm_id_file_remove(const char* path) {
    if (!path) return GSMCAL_E_ARG;
    (void)unlink(path);
    (void)unlink((std::string(path) + ".tmp").c_str());
    return 0;
}

// age_test: also reject a record older than the stale window (GSMCAL_COMM_STALE_S, 120 s).  Always on for nonce 0; on as well
// for a nonce that was only DERIVED from the environment (default_launch_nonce): plain torchrun gives every launch the same
// MASTER_ADDR:MASTER_PORT, so a derived nonce may repeat across launches and must not switch the age test off.
static int id_file_exchange(const char* path, unsigned long long nonce, bool age_test, int world, int rank, void* id_inout, double timeout_s) {
    if (!path || !id_inout || world < 1 || rank < 0 || rank >= world) return GSMCAL_E_ARG;
    const size_t rec = 16 + GSMCAL_COMM_ID_BYTES;
    unsigned char buf[16 + GSMCAL_COMM_ID_BYTES];
    if (rank == 0) {
        (void)gsmcal_comm_id_file_remove(path);                           // whatever an earlier (crashed) bootstrap left behind
        memcpy(buf, &GSMCAL_ID_MAGIC, 8);
        memcpy(buf + 8, &nonce, 8);
        memcpy(buf + 16, id_inout, GSMCAL_COMM_ID_BYTES);
        const std::string tmp = std::string(path) + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(buf, 1, rec, f) != rec) { if (f) fclose(f); return GSMCAL_E_ARG; }
        if (fclose(f) != 0) return GSMCAL_E_ARG;
        if (rename(tmp.c_str(), path) != 0) return GSMCAL_E_ARG;          // atomic: readers never see half a record
        return 0;
    }
    double stale_s = 120.0;
    if (const char* e = getenv("GSMCAL_COMM_STALE_S")) { const double v = atof(e); if (v > 0.0) stale_s = v; }
    const long tries = (long)(timeout_s > 0.0 ? timeout_s * 100.0 : 6000.0);
    for (long t = 0; t < tries; ++t) {
        FILE* f = fopen(path, "rb");
        if (f) {
            const size_t got = fread(buf, 1, rec, f);
            const bool more = got == rec && fgetc(f) != EOF;
            struct stat sb;
            const bool have_sb = fstat(fileno(f), &sb) == 0;
            fclose(f);
            unsigned long long magic = 0, fn = 0;
            memcpy(&magic, buf, 8);
            memcpy(&fn, buf + 8, 8);
            bool ok = got == rec && !more && magic == GSMCAL_ID_MAGIC && fn == nonce;
            // no caller-chosen nonce to tell runs apart: a record older than the stale window belongs to a bootstrap that died
            if (ok && (nonce == 0 || age_test)) ok = have_sb && difftime(time(nullptr), sb.st_mtime) <= stale_s;
            if (ok) { memcpy(id_inout, buf + 16, GSMCAL_COMM_ID_BYTES); return 0; }
        }
        usleep(10000);
    }
    return GSMCAL_E_ARG;
}

int gsmcal_comm_id_file_exchange(const char* path, unsigned long long nonce, int world, int rank, void* id_inout, double timeout_s) {
    return id_file_exchange(path, nonce, nonce == 0, world, rank, id_inout, timeout_s);
}

int gsmcal_comm_id_file_exchange_aged(const char* path, unsigned long long nonce, int world, int rank, void* id_inout, double timeout_s) {
    return id_file_exchange(path, nonce, true, world, rank, id_inout, timeout_s);
}

static int comm_init_file(gsmcal_ctx* c, const char* path, unsigned long long nonce, bool age_test, int world, int rank, gsmcal_comm** out) {
    if (!c || !path || !out || world < 1 || rank < 0 || rank >= world) return GSMCAL_E_ARG;
    unsigned char id[GSMCAL_COMM_ID_BYTES];
    if (rank == 0) RET_IF(gsmcal_comm_get_unique_id(id));
    if (id_file_exchange(path, nonce, age_test, world, rank, id, 60.0) != 0) {
        c->err = rank == 0 ? "cannot publish the id file" : "timed out waiting for rank 0's id file (this run's nonce)";
        return GSMCAL_E_ARG;
    }
    const int rc = gsmcal_comm_init_rank(c, id, world, rank, out);
    if (rank == 0) (void)gsmcal_comm_id_file_remove(path);              // every rank has joined (or the bootstrap failed): the id is spent
    return rc;
}

int gsmcal_comm_init_file_nonce(gsmcal_ctx* c, const char* path, unsigned long long nonce, int world, int rank, gsmcal_comm** out) {
    return comm_init_file(c, path, nonce, nonce == 0, world, rank, out);
}

// The nonce gsmcal_comm_init_file uses when the caller names none: GSMCAL_COMM_NONCE if set, else a hash of what identifies
// this LAUNCH to every one of its ranks -- the launcher's run id (TORCHELASTIC_RUN_ID, unless it is torchrun's literal default
// "none") with its restart count, a batch scheduler's job id, and the rendezvous address (MASTER_ADDR:MASTER_PORT).  0 when the
// environment offers none of these.  *strong = the caller chose it (GSMCAL_COMM_NONCE): only then may readers skip the age
// test.  A derived nonce can repeat -- plain `torchrun` has RUN_ID "none" and the static 127.0.0.1:29500 in every launch -- so
// records carrying it are still held to the stale window (GSMCAL_COMM_STALE_S): an id file a crashed bootstrap left behind is
// rejected by the nonce when the launcher tells launches apart and by its age when it does not.
static unsigned long long default_launch_nonce(bool* strong = nullptr) {
    if (strong) *strong = false;
    if (const char* e = getenv("GSMCAL_COMM_NONCE")) { if (strong) *strong = true; return strtoull(e, nullptr, 0); }
    std::string id;
    for (const char* name : {"TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "SLURM_JOB_ID", "SLURM_STEP_ID", "PBS_JOBID", "LSB_JOBID"})
        if (const char* v = getenv(name)) {
            if (!*v) continue;
            if (!strcmp(name, "TORCHELASTIC_RUN_ID") && !strcmp(v, "none")) continue;       // torch.distributed.run's default: identifies nothing
            if (!strcmp(name, "TORCHELASTIC_RESTART_COUNT") && !strcmp(v, "0") && id.empty()) continue;   // (a first attempt without a run id says nothing either)
            id += name; id += '='; id += v; id += ';';
        }
    const char* ma = getenv("MASTER_ADDR");
    const char* mp = getenv("MASTER_PORT");
    if (ma && mp && *ma && *mp) { id += ma; id += ':'; id += mp; }
    if (id.empty()) return 0;
    unsigned long long h = 0xcbf29ce484222325ull;           // FNV-1a, 64 bit
    for (unsigned char ch : id) { h ^= ch; h *= 0x100000001b3ull; }
    return h ? h : 1;
}

unsigned long long gsmcal_comm_default_nonce(void) { return default_launch_nonce(); }

int gsmcal_comm_init_file(gsmcal_ctx* c, const char* path, int world, int rank, gsmcal_comm** out) {
    bool strong = false;
    const unsigned long long nonce = default_launch_nonce(&strong);
    return comm_init_file(c, path, nonce, !strong, world, rank, out);
}

void gsmcal_comm_destroy(gsmcal_comm* g) {
    if (!g) return;
    RcclApi* a = rccl_api();
    if (a && g->comm) (void)a->CommDestroy(g->comm);
    delete g;
}

int gsmcal_allgather_table(gsmcal_ctx* c, gsmcal_comm* g, const double* d_local, int rows_per_rank, int cols, double* d_all) {
    if (!c || !g || !d_local || !d_all || rows_per_rank < 1 || cols < 1) return GSMCAL_E_ARG;
    RcclApi* a = rccl_api();
    if (!a) return GSMCAL_E_UNSUPPORTED;
    HIPCHK(c, hipSetDevice(c->device));
    const ncclResult_t r = a->AllGather(d_local, d_all, (size_t)rows_per_rank * cols, ncclDouble, g->comm, c->stream);
    if (r != ncclSuccess) {
        c->err = std::string("ncclAllGather: ") + (a->GetErrorString ? a->GetErrorString(r) : "failed");
        return GSMCAL_E_HIP;
    }
    return 0;
}

// The same collective OFF the chain's critical path: RCCL runs on a side stream of the context, ordered
// behind an event recorded on the context's stream now; the context's stream itself does not wait, so the next batch's
// kernels start at once and the gather of batch i travels under the kernels of batch i+1.
int gsmcal_allgather_table_async(gsmcal_ctx* c, gsmcal_comm* g, const double* d_local, int rows_per_rank, int cols, double* d_all, int slot) {
    if (!c || !g || !d_local || !d_all || rows_per_rank < 1 || cols < 1 || slot < 0 || slot >= gsmcal_ctx::AG_SLOTS) return GSMCAL_E_ARG;
    RcclApi* a = rccl_api();
    if (!a) return GSMCAL_E_UNSUPPORTED;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->ag_stream) {
        int lo = 0, hi = 0;                                     // lowest priority: the collective never delays the chain's kernels
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&c->ag_stream, hipStreamNonBlocking, lo) != hipSuccess) {
            (void)hipGetLastError();
            HIPCHK(c, hipStreamCreateWithFlags(&c->ag_stream, hipStreamNonBlocking));
        }
    }
    if (!c->ag_ready[slot]) {
        // device-scope release: the table only has to be visible to the collective's kernel on this device; a default event
        // flushes to system scope at every record (see get_event())
        unsigned fl = hipEventDisableTiming | hipEventReleaseToDevice;
        if (const char* e = getenv("GSMCAL_AG_EVENT_FLAGS")) fl = (unsigned)strtoul(e, nullptr, 0);
        if (hipEventCreateWithFlags(&c->ag_ready[slot], fl) != hipSuccess) {
            (void)hipGetLastError();
            HIPCHK(c, hipEventCreateWithFlags(&c->ag_ready[slot], hipEventDisableTiming));
        }
    }
    if (!c->ag_done[slot]) HIPCHK(c, hipEventCreateWithFlags(&c->ag_done[slot], hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ag_ready[slot], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->ag_stream, c->ag_ready[slot], 0));
    const ncclResult_t r = a->AllGather(d_local, d_all, (size_t)rows_per_rank * cols, ncclDouble, g->comm, c->ag_stream);
    if (r != ncclSuccess) {
        c->err = std::string("ncclAllGather: ") + (a->GetErrorString ? a->GetErrorString(r) : "failed");
        return GSMCAL_E_HIP;
    }
    HIPCHK(c, hipEventRecord(c->ag_done[slot], c->ag_stream));
    c->ag_posted[slot] = true;
    return 0;
}

int gsmcal_allgather_wait(gsmcal_ctx* c, int slot) {
    if (!c || slot < 0 || slot >= gsmcal_ctx::AG_SLOTS) return GSMCAL_E_ARG;
    if (c->ag_posted[slot]) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ag_done[slot], 0));
    return 0;
}

int gsmcal_allgather_sync(gsmcal_ctx* c, int slot) {
    if (!c || slot < 0 || slot >= gsmcal_ctx::AG_SLOTS) return GSMCAL_E_ARG;
    if (c->ag_posted[slot]) HIPCHK(c, hipEventSynchronize(c->ag_done[slot]));
    return 0;
}

// ---- ingest ring ---------------------------------------------------------------------------------------------------
struct gsmcal_ring {
    gsmcal_ctx* c = nullptr;
    size_t bytes = 0;
    int n = 0;
    hipStream_t copy = nullptr;
    std::vector<void*> host, dev;
    std::vector<hipEvent_t> copied, consumed;      // H2D of the slot done / consumer kernels of the slot done
    std::vector<char> has_consumed;
};

int gsmcal_ring_create(gsmcal_ctx* c, size_t batch_bytes, int slots, gsmcal_ring** out) {
    if (!c || !out || batch_bytes < 1 || slots < 2 || slots > 16) return GSMCAL_E_ARG;
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    gsmcal_ring* r = new gsmcal_ring();
    r->c = c; r->bytes = batch_bytes; r->n = slots;
    r->host.assign(slots, nullptr); r->dev.assign(slots, nullptr);
    r->copied.assign(slots, nullptr); r->consumed.assign(slots, nullptr); r->has_consumed.assign(slots, 0);
    bool ok = hipStreamCreateWithFlags(&r->copy, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < slots; ++i) {
        ok = hipHostMalloc(&r->host[i], batch_bytes, hipHostMallocDefault) == hipSuccess &&
             hipMalloc(&r->dev[i], batch_bytes) == hipSuccess &&
             hipEventCreateWithFlags(&r->copied[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&r->consumed[i], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { c->err = "ring allocation failed"; gsmcal_ring_destroy(r); return GSMCAL_E_HIP; }
    *out = r;
    return 0;
}

void gsmcal_ring_destroy(gsmcal_ring* r) {
    if (!r) return;
    (void)hipSetDevice(r->c->device);
    if (r->copy) (void)hipStreamSynchronize(r->copy);
    (void)hipStreamSynchronize(r->c->stream);
    for (int i = 0; i < r->n; ++i) {
        if (r->host[i]) (void)hipHostFree(r->host[i]);
        if (r->dev[i]) (void)hipFree(r->dev[i]);
        if (r->copied[i]) (void)hipEventDestroy(r->copied[i]);
        if (r->consumed[i]) (void)hipEventDestroy(r->consumed[i]);
    }
    if (r->copy) (void)hipStreamDestroy(r->copy);
    delete r;
}

void* gsmcal_ring_host(gsmcal_ring* r, int slot) { return (r && slot >= 0 && slot < r->n) ? r->host[slot] : nullptr; }

int gsmcal_ring_submit(gsmcal_ring* r, int slot, size_t bytes) {
    if (!r || slot < 0 || slot >= r->n || bytes > r->bytes) return GSMCAL_E_ARG;
    gsmcal_ctx* c = r->c;
    HIPCHK(c, hipSetDevice(c->device));
    if (r->has_consumed[slot]) HIPCHK(c, hipStreamWaitEvent(r->copy, r->consumed[slot], 0));   // the device twin is free again
    HIPCHK(c, hipMemcpyAsync(r->dev[slot], r->host[slot], bytes ? bytes : r->bytes, hipMemcpyHostToDevice, r->copy));
    HIPCHK(c, hipEventRecord(r->copied[slot], r->copy));
    return 0;
}

void* gsmcal_ring_acquire(gsmcal_ring* r, int slot) {
    if (!r || slot < 0 || slot >= r->n) return nullptr;
    if (hipStreamWaitEvent(r->c->stream, r->copied[slot], 0) != hipSuccess) return nullptr;
    return r->dev[slot];
}

int gsmcal_ring_release(gsmcal_ring* r, int slot) {
    if (!r || slot < 0 || slot >= r->n) return GSMCAL_E_ARG;
    HIPCHK(r->c, hipEventRecord(r->consumed[slot], r->c->stream));
    r->has_consumed[slot] = 1;
    return 0;
}

int gsmcal_ring_host_ready(gsmcal_ring* r, int slot) {
    if (!r || slot < 0 || slot >= r->n) return GSMCAL_E_ARG;
    HIPCHK(r->c, hipEventSynchronize(r->copied[slot]));
    return 0;
}

// ---- synthetic-input utility ---------------------------------------------------------------------------
int gsmcal_synth_expand_dev(gsmcal_ctx* c, const uint8_t* d_base, int k, long n, uint8_t* d_out, long d, long first_unit,
                            unsigned long long seed) {
    if (!c || !d_base || !d_out || k < 1 || n < 1 || d < 1 || first_unit < 0) return GSMCAL_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    c->cur = &c->lanes[0];
    for (long lo = 0; lo < d; lo += 32768) {            // grid.y limit
        const long cnt = d - lo < 32768 ? d - lo : 32768;
        long bx = (n / 8 + 255) / 256;
        if (bx > 64) bx = 64;
        LAUNCH(c, k_synth_expand, dim3((unsigned)bx, (unsigned)cnt), dim3(256), 0, d_base, k, n, d_out + (size_t)lo * 2 * n,
               first_unit + lo, seed);
    }
    CHECK_LAUNCH(c);
    return 0;
}

int gsmcal_last_batch_details(gsmcal_ctx* c, int d, double* coarse_pos, double* coarse_snr, double* fine_first,
                              double* fcch_pos, double* sch_first, int* counts) {
    if (!c || d < 1 || d > c->last_S) return GSMCAL_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<StreamState> v((size_t)c->last_S);
    if (c->n_lanes_used <= 1 && c->lanes[0].n == 0) { c->lanes[0].lo = 0; c->lanes[0].n = c->last_S; }
    for (int i = 0; i < c->n_lanes_used; ++i) {
        const Lane& L = c->lanes[i];
        if (L.n <= 0 || L.lo + L.n > c->last_S) continue;
        HIPCHK(c, hipMemcpy(v.data() + L.lo, L.state.p, (size_t)L.n * sizeof(StreamState), hipMemcpyDeviceToHost));
    }
    for (int s = 0; s < d; ++s) {
        const StreamState& st = v[s];
        for (int i = 0; i < MAXH; ++i) {
            if (coarse_pos) coarse_pos[(size_t)s * MAXH + i] = i < st.n_coarse ? st.coarse_pos[i] : 0.0;
            if (coarse_snr) coarse_snr[(size_t)s * MAXH + i] = i < st.n_coarse ? st.coarse_snr[i] : 0.0;
            if (fine_first) fine_first[(size_t)s * MAXH + i] = i < st.n_fine ? st.fine_first[i] : 0.0;
            if (fcch_pos) fcch_pos[(size_t)s * MAXH + i] = i < st.n_fcch ? st.fcch_pos[i] : 0.0;
            if (sch_first) sch_first[(size_t)s * MAXH + i] = i < st.n_sch_first ? st.sch_first[i] : 0.0;
        }
        if (counts) {
            counts[5 * s + 0] = st.n_coarse;
            counts[5 * s + 1] = st.n_fine;
            counts[5 * s + 2] = st.fcch_is_sentinel ? -1 : st.n_fcch;
            counts[5 * s + 3] = st.n_sch_first;
            counts[5 * s + 4] = st.n_rows;
        }
    }
    return 0;
}

int gsmcal_last_batch_snr(gsmcal_ctx* c, int stream, double* snr, long cap, long* n_table, long* n_moving) {
    if (!c || stream < 0 || stream >= c->last_S || !snr || cap < 1) return GSMCAL_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < c->n_lanes_used; ++i) {
        const Lane& L = c->lanes[i];
        if (stream < L.lo || stream >= L.lo + L.n) continue;
        if (L.snr_stride <= 0 && L.snr_nmove > 0) {
            c->err = "the last batch kept no SNR table (throughput batches compute the window SNRs inside the scan kernel; GSMCAL_SNR_INLINE_KEEP=1 writes it out)";
            return GSMCAL_E_UNSUPPORTED;
        }
        if (!L.snrbuf.p || L.snr_stride <= 0) continue;
        const long n = L.snr_stride < cap ? L.snr_stride : cap;
        HIPCHK(c, hipMemcpy(snr, (const double*)L.snrbuf.p + (size_t)(stream - L.lo) * L.snr_stride, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        if (n_table) *n_table = L.snr_stride;
        if (n_moving) *n_moving = L.snr_nmove;
        return 0;
    }
    return GSMCAL_E_ARG;
}

#ifdef GSMCAL_DEVTIMING
// Development build only: in-kernel phase timestamps (state.h DEV_STAMP).  begin() arms a zeroed buffer, report()
// prints, per kernel, the span of the launch and the mean time between consecutive stamps of a block.
static void* g_stamp_buf = nullptr;
int gsmcal_devtiming_begin(gsmcal_ctx* c) {
    if (!c) return GSMCAL_E_ARG;
    const size_t bytes = (size_t)KID_N * DEV_STAMP_BLOCKS * 16 * sizeof(unsigned long long);
    HIPCHK(c, hipDeviceSynchronize());
    if (!g_stamp_buf) HIPCHK(c, hipMalloc(&g_stamp_buf, bytes));
    HIPCHK(c, hipMemset(g_stamp_buf, 0, bytes));
    HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &g_stamp_buf, sizeof(g_stamp_buf)));
    return 0;
}
int gsmcal_devtiming_report(gsmcal_ctx* c) {
    if (!c || !g_stamp_buf) return GSMCAL_E_ARG;
    static const char* names[KID_N] = {"coarse_snr", "coarse_scan", "gather|post_chain barriers", "cert", "chunk", "verify", "burst_tone<1>", "window_sch", "burst_tone<0>", "front"};
    HIPCHK(c, hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)KID_N * DEV_STAMP_BLOCKS * 16);
    HIPCHK(c, hipMemcpy(h.data(), g_stamp_buf, h.size() * 8, hipMemcpyDeviceToHost));
    void* z = nullptr;
    HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &z, sizeof(z)));
    if (const char* dump = getenv("GSMCAL_DEVTIMING_DUMP")) {          // raw stamps: kernel id, block, stamp, 10 ns ticks
        if (FILE* f = fopen(dump, "w")) {
            for (int k = 0; k < KID_N; ++k)
                for (int b = 0; b < DEV_STAMP_BLOCKS; ++b)
                    for (int i = 0; i < 16; ++i) {
                        const unsigned long long t = h[((size_t)k * DEV_STAMP_BLOCKS + b) * 16 + i];
                        if (t) fprintf(f, "%d,%d,%d,%llu\n", k, b, i, t);
                    }
            fclose(f);
        }
    }
    for (int k = 0; k < KID_N; ++k) {
        unsigned long long t0 = ~0ull, t1 = 0;
        double ph[15] = {0}; int pc[15] = {0}; int nb = 0;
        for (int b = 0; b < DEV_STAMP_BLOCKS; ++b) {
            const unsigned long long* r = &h[((size_t)k * DEV_STAMP_BLOCKS + b) * 16];
            if (!r[0]) continue;
            ++nb;
            if (r[0] < t0) t0 = r[0];
            unsigned long long prev = r[0];
            for (int i = 1; i < 16; ++i) {
                if (!r[i]) continue;
                if (r[i] > t1) t1 = r[i];
                ph[i - 1] += (double)(r[i] - prev) / 100.0; ++pc[i - 1];
                prev = r[i];
            }
        }
        if (!nb) continue;
        if (k == KID_VERIFY) {
            for (int b = 0; b < DEV_STAMP_BLOCKS; ++b) {
                const unsigned long long* r = &h[((size_t)k * DEV_STAMP_BLOCKS + b) * 16];
                if (!r[0] || !r[4]) continue;
                fprintf(stderr, "verify block %d: items %llu open %llu | list %.1f anchors %.1f slides %.1f\n", b, (r[15] / 100ull) % 100000ull, r[15] / 10000000ull,
                        (double)(r[3] - r[0]) / 100.0, (double)(r[4] - r[3]) / 100.0, (double)(r[5] - r[4]) / 100.0);
            }
        }
        std::vector<double> st0, en0;
        for (int b = 0; b < DEV_STAMP_BLOCKS; ++b) {
            const unsigned long long* r = &h[((size_t)k * DEV_STAMP_BLOCKS + b) * 16];
            if (!r[0]) continue;
            unsigned long long e = r[0];
            for (int i = 1; i < 16; ++i) if (r[i] > e) e = r[i];
            st0.push_back((double)(r[0] - t0) / 100.0);
            en0.push_back((double)(e - t0) / 100.0);
        }
        std::sort(st0.begin(), st0.end());
        std::sort(en0.begin(), en0.end());
        static unsigned long long g0 = 0;
        if (k == 0 || !g0) g0 = t0;
        fprintf(stderr, "devtiming abs [%7.1f .. %7.1f] ", ((double)t0 - (double)g0) / 100.0, ((double)t1 - (double)g0) / 100.0);
        fprintf(stderr, "devtiming %-14s blocks %4d span %7.1f us | start p50 %.1f p90 %.1f max %.1f | end p50 %.1f p90 %.1f | phases:", names[k], nb,
                t1 > t0 ? (double)(t1 - t0) / 100.0 : 0.0, st0[st0.size() / 2], st0[st0.size() * 9 / 10], st0.back(),
                en0[en0.size() / 2], en0[en0.size() * 9 / 10]);
        for (int i = 0; i < 15; ++i) if (pc[i]) fprintf(stderr, " [%d->%d] %.1f (n=%d)", i, i + 1, ph[i] / pc[i], pc[i]);
        fprintf(stderr, "\n    mean time of stamp i after stamp 0:");
        for (int i = 1; i < 16; ++i) {
            double a = 0.0; int n = 0;
            for (int b = 0; b < DEV_STAMP_BLOCKS; ++b) {
                const unsigned long long* r = &h[((size_t)k * DEV_STAMP_BLOCKS + b) * 16];
                if (r[0] && r[i]) { a += ((double)r[i] - (double)r[0]) / 100.0; ++n; }
            }
            if (n) fprintf(stderr, " %d:%.1f", i, a / n);
        }
        fprintf(stderr, "\n");
    }
    // what the certificate of the last batch left open, and what the chunk sweep handed on
    Lane& L = c->lanes[0];
    if (L.cert.p && L.chunkrec.p && L.state.p && c->last_S > 0 && L.win_l0_H > 0) {
        const int S = c->last_S, H = L.win_l0_H;
        std::vector<FineCert> fc((size_t)S * H);
        std::vector<StreamState> st(S);
        HIPCHK(c, hipMemcpy(fc.data(), L.cert.p, fc.size() * sizeof(FineCert), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(st.data(), L.state.p, st.size() * sizeof(StreamState), hipMemcpyDeviceToHost));
        const Geom g(8);
        const int nchunk = (g.fine_nshift - 1 + FS_CHUNK - 1) / FS_CHUNK;
        std::vector<ChunkRec> rec((size_t)S * H * nchunk);
        if (L.chunkrec.cap >= rec.size() * sizeof(ChunkRec)) {
            HIPCHK(c, hipMemcpy(rec.data(), L.chunkrec.p, rec.size() * sizeof(ChunkRec), hipMemcpyDeviceToHost));
            int hist[20] = {0}, nwin = 0, cand_hist[8] = {0}, win_with_cand = 0, n_open_chunks = 0;
            for (int s = 0; s < S; ++s)
                for (int w = 0; w < H && w < st[s].n_fine_ws; ++w) {
                    const FineCert& f = fc[(size_t)s * H + w];
                    ++nwin; ++hist[f.nch < 19 ? f.nch : 19]; n_open_chunks += f.nch;
                    int tot = 0;
                    for (int k = 0; k < f.nch; ++k) { const int cnt = rec[((size_t)s * H + w) * nchunk + (k < f.nch - f.nsuf ? k : nchunk - f.nch + k)].count; tot += cnt < 0 ? 100 : cnt; }
                    if (f.nch > 0) { ++cand_hist[tot < 7 ? tot : 7]; if (tot) ++win_with_cand; }
                }
            fprintf(stderr, "certificate: %d windows; open chunks per window:", nwin);
            for (int i = 0; i < 20; ++i) if (hist[i]) fprintf(stderr, " %d:%d", i, hist[i]);
            fprintf(stderr, "  (%d open chunks in all)", n_open_chunks);
            fprintf(stderr, "\n  candidates the chunk sweep handed to the exact pass, per window with open chunks:");
            for (int i = 0; i < 8; ++i) if (cand_hist[i]) fprintf(stderr, " %d%s:%d", i, i == 7 ? "+" : "", cand_hist[i]);
            fprintf(stderr, "  (%d windows with any)\n", win_with_cand);
        }
    }
    return 0;
}
#endif

}  // extern "C"
