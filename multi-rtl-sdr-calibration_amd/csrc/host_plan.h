// host_plan.h -- context, workspace, lanes, launch sequences and graph replay of libgsmcal.so (included by gsmcal.hip only).
//
// The calibration chain is enqueued as a fixed sequence of kernels on one HIP stream; all data-dependent control lives in
// StreamState on the device (state.h).  The same building blocks serve the per-function MATLAB-signature entry points
// (level 0 = a complex array handed in) and the batched hot path (level 0 = FIR of the raw bytes, evaluated lazily).
#pragma once
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
    DevBuf state, dec, win, peaks, snrbuf, x0, chunkrec, openlist, cert, partial, tailctr, xch, xepoch;
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
    std::string report;             // console text the reference's function would have printed in the last per-function call (abi_report.h)
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
        int fused_streams = 0;      // streams of fused tails inside the captured plan (fused_gate: checked at every replay)
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
    const void* last_raw = nullptr; // raw buffer of the previous batch call: one that comes again may still sit in the Infinity Cache
    bool call_raw_fresh = false;    // the call in progress reads another buffer than the previous call did
    int scan_split = 88;            // GSMCAL_SCAN_SPLIT: percent of a pipeline stage's captures in the first of its two front-kernel launches (0: one launch;
                                    // 12 800 captures: 0 / 70 / 80 / 88 / 94 -> 3.64 / 3.58 / 3.525 / 3.515 / 3.57 ms)
    int scan_stages = 0;            // GSMCAL_SCAN_STAGES: pipeline stages of a big scanner batch (0: by batch size)
    int lane_min = 64;              // GSMCAL_LANE_MIN: fewest streams a lane is worth forking for
    bool certify = true;            // GSMCAL_CERT=0: no Parseval certificate, every chunk of every window is swept
    bool snr_full = true;           // GSMCAL_SNR_FULL=0: the hop walk of FCCH_coarse_position computes its own 16-point spectra
    double snr_screen_db = 5.0;     // GSMCAL_SNR_SCREEN_DB: level below which k_coarse_snr proves windows instead of computing them
                                    // (typical thresholds hit_avg_snr + th sit at 6.5 .. 7.5 dB; ~95 % of the windows are below 5)
    bool fuse_fine_gather = true;   // GSMCAL_FUSE_GATHER=0: a k_gather launch writes the fine windows, k_fine_cert reads them back
    // fused tail bookkeeping (fused_gate), per stream index of lane 0: fused tails this context has enqueued over the stream /
    // the launch count workgroup 0 of the stream stored at the end of its last k_post_chain_r (pinned host words)
    static constexpr int FUSED_MAX_STREAMS = 1024;
    std::vector<unsigned> fused_expected;
    unsigned* fused_done = nullptr;
    int fused_hi = 0;               // streams [0, fused_hi) may have a fused tail in flight
    unsigned long long n_fused_launches = 0, n_gate_fallbacks = 0;   // gsmcal_fused_tail_stats
    int capture_fused_streams = 0;  // streams of the fused tail enqueued during the stream capture in progress
    // A fused tail whose workgroups gave up waiting for a peer (another PROCESS's fused tail holding the slots: outside the gate) tells
    // the host through a pinned word; gsmcal_sync / the host-buffer entry points then run the recorded calls again with the
    // four-launch tail (fused_recover in abi_calls.h).
    struct FusedCall {
        const uint8_t* d_raw; int d; long n; int ntaps, len_ts;
        std::vector<double> coef, ts, cf;
        double* d_table; double* d_pos_info; double* d_r_correct; long* d_r_len;
    };
    std::vector<FusedCall> fused_calls;    // calls that took the fused tail since the last synchronisation that found no time-out (at most 16)
    double fused_poll_s = 3.0;             // GSMCAL_FUSED_POLL_S: how long a workgroup waits for its stream's peers
    int test_stall = 0;                    // GSMCAL_TEST_FUSED_STALL=<1..4> (test hook): one workgroup never publishes that stage's result
    unsigned long long n_tail_reruns = 0;  // gsmcal_fused_tail_reruns
    bool recovering = false;
    bool fuse_post = true;          // GSMCAL_FUSE_POST=0: k_fine_verify, k_burst_tone<1>, k_window_sch, k_burst_tone<0> as four launches
    bool front_generic = false;     // GSMCAL_FRONT_GENERIC=1: the any-geometry front kernel also for the 47/31-tap production geometry
    bool capturing = false;
    // ---- calls in flight (gsmcal_ctx_set_pipeline_depth; gsm_sync_demod.m:107-124 over consecutive batches) ----
    // Up to `pipe_depth` consecutive single-lane batch calls in flight, each in a workspace of its own (pipe[slot]): WHOLE calls side
    // by side, call i on internal stream i mod depth behind an event on the context's stream, every calibration call with the
    // FOUR-LAUNCH tail -- its kernels wait for nobody and hold no slot while idle, so the kernels of up to four calls interleave
    // workgroup by workgroup as slots free up: 64 streams 0.177 -> 0.148 / 0.138 ms per call at depth 3 / 4 (more streams than the
    // device's four hardware queues gain nothing).  The forms measured and dropped -- a call cut into stages on stage streams with
    // the fused tail kept (no gain: the fused tail and the certificate each fill every CU's registers and LDS), fused tails chained
    // or mixed in among side-by-side calls (slower) -- are profiles/experiments_r06/staged_pipeline_and_side_fused.patch, NOTES_r06.
    static constexpr int PIPE_MAX_DEPTH = 8;
    int pipe_depth = 1;             // 1: a call is complete in stream order when it returns (the semantics of every earlier release)
    Lane pipe[PIPE_MAX_DEPTH];
    hipStream_t side_stream[PIPE_MAX_DEPTH] = {};     // call i runs on side_stream[i mod depth]; entries repeat when fewer streams run side by side
    std::vector<hipStream_t> side_owned;              // the distinct streams behind side_stream[] (pipe_pick_streams)
    int n_side = 0;                                   // how many of them were seen running side by side (hardware queues they sit on)
    hipEvent_t side_in[PIPE_MAX_DEPTH] = {};          // the context's stream at the moment the slot's call was enqueued
    hipEvent_t pipe_done[PIPE_MAX_DEPTH] = {};        // end of the slot's call (plus a collective enqueued right behind it)
    bool pipe_pending[PIPE_MAX_DEPTH] = {};           // the slot's call has not been joined into the context's stream yet
    unsigned long pipe_calls = 0;
    int pipe_last_slot = -1;        // slot of the most recent call in flight, -1: none since the last join
    Lane* detail_lane = nullptr;    // the workspace gsmcal_last_batch_details / _snr read (call in flight), nullptr: lanes[]
    bool no_fuse_now = false;             // the call being enqueued takes the four-launch tail whatever the batch size (calls in flight)
                                          // ... and the SNR table of the moving search only (the hop walk on its own spectra): with calls in
                                          // flight the full table's 27 MB and its screening pass cost more than the walk's latency saves
                                          // (four deep, 100 steps: 0.1315 -> 0.1287 ms per call; one call at a time it is the other way round)
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
    // in-line collectives behind pipelined calls sit on those calls' streams -- different ones: an event chain keeps the collectives of
    // one communicator one behind the other on the GPU (RCCL does not allow two of them to run at once)
    hipEvent_t ag_chain[2] = {nullptr, nullptr};
    int ag_chain_n = 0;
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

// ---- at most ONE fused tail of this process in flight per device ----------------------------------------------------
// k_post_chain_r's workgroups wait for each other inside the launch.  One such launch always advances (in-order dispatch puts
// its oldest unfinished stream first in line for every free slot); two of them, from two contexts on two streams, could in
// principle each hold the slots the other's next workgroups need.  So a context takes the fused tail only while no OTHER
// context of the process has one enqueued and unfinished on the same device -- else this call uses the four-launch tail
// (same results, ~10 us slower at 64 streams).  "Unfinished" without any event or synchronisation: workgroup 0 of every stream
// stores the stream's launch count (the device-side counter that also selects the exchange block's parity) into the context's
// pinned host word for that stream as its last act -- a posted store, like the table row's -- and the host counts the fused tails
// it enqueued per stream; a context is busy while any of its words lags.  (One shared word bumped by an atomic per stream
// serialised on PCIe: +56 us per 64-stream step.)
struct FusedGate { std::mutex mu; std::vector<gsmcal_ctx*> ctxs; };
inline FusedGate& fused_gate() { static FusedGate g; return g; }

void fused_gate_register(gsmcal_ctx* c) {
    const size_t bytes = ((size_t)gsmcal_ctx::FUSED_MAX_STREAMS + 16) * sizeof(unsigned);     // (+ the time-out word behind the per-stream words)
    if (hipHostMalloc((void**)&c->fused_done, bytes, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); c->fused_done = nullptr; }
    if (c->fused_done) memset(c->fused_done, 0, bytes);
    c->fused_expected.assign(gsmcal_ctx::FUSED_MAX_STREAMS, 0u);
    std::lock_guard<std::mutex> lk(fused_gate().mu);
    fused_gate().ctxs.push_back(c);
}

void fused_gate_unregister(gsmcal_ctx* c) {
    {
        std::lock_guard<std::mutex> lk(fused_gate().mu);
        auto& v = fused_gate().ctxs;
        v.erase(std::remove(v.begin(), v.end(), c), v.end());
    }
    if (c->fused_done) (void)hipHostFree(c->fused_done);
    c->fused_done = nullptr;
}

// (the launch counters on the device were re-created, all zero: the host's view follows -- called with the device idle)
void fused_gate_reset(gsmcal_ctx* c) {
    std::lock_guard<std::mutex> lk(fused_gate().mu);
    if (c->fused_done) memset(c->fused_done, 0, (size_t)gsmcal_ctx::FUSED_MAX_STREAMS * sizeof(unsigned));
    c->fused_expected.assign(gsmcal_ctx::FUSED_MAX_STREAMS, 0u);
    c->fused_hi = 0;
}

// (gate mutex held) does context o have a fused tail enqueued that the GPU has not finished?
bool fused_busy(gsmcal_ctx* o) {
    if (!o->fused_done) return false;
    const volatile unsigned* d = o->fused_done;
    while (o->fused_hi > 0 && d[o->fused_hi - 1] == o->fused_expected[o->fused_hi - 1]) --o->fused_hi;   // (streams finish in any order: trim from the top)
    for (int s = 0; s < o->fused_hi; ++s)
        if (d[s] != o->fused_expected[s]) return true;
    o->fused_hi = 0;
    return false;
}

// (gate mutex held) may context c enqueue a fused tail over streams [0, streams) now?  true: counted as enqueued.
bool fused_gate_pass(gsmcal_ctx* c, int streams) {
    if (!c->fused_done || streams > gsmcal_ctx::FUSED_MAX_STREAMS) return false;   // (no pinned words: the gate cannot see this context -- never fuse)
    for (gsmcal_ctx* o : fused_gate().ctxs)
        if (o != c && o->device == c->device && fused_busy(o)) return false;
    for (int s = 0; s < streams; ++s) ++c->fused_expected[s];
    if (streams > c->fused_hi) c->fused_hi = streams;
    ++c->n_fused_launches;
    return true;
}

// (gate mutex NOT held) a fused tail that was counted by fused_gate_pass() never reached the queue (failed launch / graph launch):
// take it out of the count again, or this context would read as busy for good and every other context of the device would lose
// the fused tail for the rest of the process
void fused_gate_rollback(gsmcal_ctx* c, int streams) {
    std::lock_guard<std::mutex> lk(fused_gate().mu);
    for (int s = 0; s < streams && s < gsmcal_ctx::FUSED_MAX_STREAMS; ++s)
        if (c->fused_expected[s] > 0) --c->fused_expected[s];
    if (c->n_fused_launches > 0) --c->n_fused_launches;
}

// May context c enqueue a fused tail over `streams` streams now?  false: take the four-launch tail.
bool fused_gate_enter(gsmcal_ctx* c, int streams) {
    if (c->capturing) {
        // our own capture: run_maybe_graph passes the gate at every replay, counting ONE fused tail over capture_fused_streams
        // streams per replay -- a second fused tail inside the same captured plan takes the four-launch form
        if (c->capture_fused_streams > 0 || !c->fused_done) return false;
        c->capture_fused_streams = streams;
        return true;
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (c->stream && (hipStreamIsCapturing(c->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)) {
        (void)hipGetLastError();                          // the CALLER is capturing: the library will not see the replays -- never fuse
        ++c->n_gate_fallbacks;
        return false;
    }
    std::lock_guard<std::mutex> lk(fused_gate().mu);
    const bool ok = fused_gate_pass(c, streams);
    if (!ok) ++c->n_gate_fallbacks;
    return ok;
}

// Workgroups of the fused tail (k_post_chain_r<8,512,47> | k_post_chain_r<0,0,0>) that fit one CU at this dynamic LDS size, from
// the occupancy calculator of the runtime (registers, LDS granules, wave slots of the compiled kernel); cached per
// (variant, LDS size).  0: the query failed -- the four-launch tail is used.
int post_chain_blocks_per_cu(gsmcal_ctx* c, int variant, size_t lds) {
    for (const auto& e : c->occ_cache)
        if (e.variant == variant && e.lds == lds) return e.blocks;
    int nb = 0;
    hipError_t r;
    if (variant == 0) r = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_post_chain_r<8, 512, 47>, PC_THREADS, lds);
    else r = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_post_chain_r<0, 0, 0>, PC_THREADS, lds);
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
                bool sym = src.ntaps == 47 && (int)c->h_coef.size() == 47;
                for (int k = 0; sym && k < 23; ++k) sym = c->h_coef[k] == c->h_coef[46 - k];
                fg.sym47 = sym ? 1 : 0;
                break;
            }
        }
    }
    if (!fg.raw) RET_IF(launch_gather(c, S, src, lvl, g.fine_wlen, false, H, win, sstride, wstride));
    // from here on the lane's window buffer holds level 0 of every fine window (nothing later in a batch call writes it)
    c->cur->win_l0_len = (src.kind == SRC_RAW && lvl == 0) ? g.fine_wlen : 0;
#if defined(GSMCAL_AB_NO_REUSE_L0) || defined(GSMCAL_AB_LAZY_WIN)
    c->cur->win_l0_len = 0;      // A/B builds of tools/ab_traffic.sh only: every per-burst gather filters its raw bytes again (no read of the window buffer)
#endif
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
        Lane* const X = c->cur;
        if (chain) {
            // ---- the fused tail of the chain: verify -> bursts -> SCH windows -> post-SCH bursts in one launch ----
            const int wl_sch = g.sch_nshift - 1 + len_ts;
            const size_t sch_scratch = (size_t)(len_ts + g.sch_nshift * SCH_PARTS) * sizeof(cplx) + (size_t)g.sch_nshift * sizeof(double);
            // replicated decisions (k_post_chain_r): the state copy stays in LDS in front of the stages' work area, and
            // the burst stages stage their (rare) raw-byte fallback without bank padding so that three workgroups still fit a CU
            size_t lds = vlds;
            lds = std::max(lds, fused_lds(src, lvl + 1, g.nfft, burst_scratch(g), true));
            lds = std::max(lds, fused_lds(src, lvl + 2, wl_sch, sch_scratch));
            lds = std::max(lds, fused_lds(src, lvl + 3, g.nfft, burst_scratch(g), true));
            lds = std::max(lds, (sizeof(StreamState) + 15) & ~(size_t)15);
            lds += PCR_STATE_BYTES;
            const bool ref_geom = g.ov == 8 && g.nfft == 148 * 8 && g.fine_nshift == 128 * 8 + 1 && g.sch_nshift == 11 * 8 + 1 &&
                                  len_ts == 512 && wl_sch == 11 * 8 + 512 && src.ntaps == 47;
            // Used while every workgroup of the launch fits the chip at once -- a PERFORMANCE choice: a workgroup waiting at a
            // stream's exchange holds its slot, which costs nothing in the latency regime (64 streams: 0.234 vs 0.237 ms per
            // step) and a fifth of the throughput beyond it (256 streams: 0.73 vs 0.63 ms; 1024: 2.53 vs 2.13); and only for a
            // call that runs on ONE lane (four lanes of 64 streams each would put 3 072 waiting workgroups on 768 slots --
            // measured 0.70 against 0.63 ms at 256 streams).  The slots per CU come from the occupancy of the very
            // instantiation and LDS size that would be launched, not from a literal.  What the kernel's forward
            // progress rests on is not this calculation but in-order dispatch + fused_gate_enter() below (k_post_chain_r's header).
            const int variant = ref_geom ? 0 : 1;
            const int per_cu = lds <= 159 * 1024 && H <= MAXH ? post_chain_blocks_per_cu(c, variant, lds) : 0;
            chain->fused = c->fuse_post && !c->no_fuse_now && cert_ok && src.kind == SRC_RAW && lvl == 0 && next_sch_lvl == 2 && per_cu > 0 &&
                           (long)H * S <= (long)per_cu * c->n_cu && c->n_lanes_used == 1;
            if (chain->fused) {
                // k_post_chain_r's exchange block [S][2 parities][4 stages][MAXH][2] and launch counters [S]: the layout
                // does not depend on the batch geometry (fixed MAXH stride per sub-block, one counter per stream in a buffer of its
                // own), and every launch leaves the parity it did not use EMPTY for all MAXH windows -- so launches of any
                // (S, H), eager or replayed from a graph captured here or by the caller, may follow each other.
                // Only growth re-creates the pair (all granules EMPTY, all counters zero); ensure() bumps ws_epoch then.
                const size_t need_x = (size_t)S * 2 * 4 * 2 * MAXH * sizeof(unsigned long long), need_e = (size_t)S * sizeof(unsigned);
                if (X->xch.cap < need_x || X->xepoch.cap < need_e) {
                    if (c->capturing) return GSMCAL_E_HIP;      // (cannot happen: the eager call before a capture sized both)
                    RET_IF(ensure(c, X->xch, need_x));
                    RET_IF(ensure(c, X->xepoch, need_e));
                    HIPCHK(c, hipMemsetAsync(X->xch.p, 0xFF, X->xch.cap, c->cur->stream));
                    HIPCHK(c, hipMemsetAsync(X->xepoch.p, 0, X->xepoch.cap, c->cur->stream));
                    fused_gate_reset(c);                      // (ensure() left the device idle: nothing of this context is in flight)
                }
                chain->fused = fused_gate_enter(c, S);      // (behind the re-creation above: it resets the gate's view of this context)
            }
            if (chain->fused) {
                PostChainArgs pa;
                memset(&pa, 0, sizeof(pa));
                pa.ga1 = gather_args(src, lvl + 1, g.nfft);
                pa.ga_sch = gather_args(src, lvl + 2, wl_sch);
                pa.ga0 = gather_args(src, lvl + 3, g.nfft);
                if (c->cur->win_l0_len > 0) {
                    for (GatherArgs* ga : {&pa.ga1, &pa.ga0}) { ga->l0 = win; ga->l0_stream_stride = sstride; ga->l0_win_stride = wstride; ga->l0_len = c->cur->win_l0_len; }
                }
                pa.ga1.pad = 1; pa.ga0.pad = 1;
                pa.sa = sa_fine;
                pa.sa.table = chain->table; pa.sa.pos_info_out = chain->pos_info_out; pa.sa.r_len_out = chain->r_len_out;
                pa.lvl_fine = lvl; pa.lvl_sch = lvl + 2; pa.lvl_post = lvl + 3;
                pa.nfft = g.nfft; pa.ov = g.ov; pa.len_ts = len_ts; pa.sch_nshift = g.sch_nshift; pa.fine_nshift = g.fine_nshift; pa.H = H;
                pa.tw_g = (const cplx*)c->tw.p; pa.ts = (const cplx*)c->ts.p;
                pa.win = win; pa.win_stream_stride = sstride; pa.win_stride = wstride;
                pa.rec = (const ChunkRec*)c->cur->chunkrec.p; pa.cert = certp; pa.peaks = peaks; pa.n_open = n_open;
                pa.with_totals = chain->table ? 1 : 0;
                pa.done = c->fused_done;
                pa.timed_out = c->fused_done ? c->fused_done + gsmcal_ctx::FUSED_MAX_STREAMS : nullptr;
                pa.poll_ticks = (unsigned long long)(c->fused_poll_s * 1e8);
                pa.test_stall = c->recovering ? 0 : c->test_stall;
                if (ref_geom) LAUNCH(c, (k_post_chain_r<8, 512, 47>), dim3(H, S), dim3(PC_THREADS), lds, st, pa, (unsigned long long*)X->xch.p, (unsigned*)X->xepoch.p);
                else LAUNCH(c, (k_post_chain_r<0, 0, 0>), dim3(H, S), dim3(PC_THREADS), lds, st, pa, (unsigned long long*)X->xch.p, (unsigned*)X->xepoch.p);
                {
                    const hipError_t le = hipGetLastError();
                    if (le != hipSuccess) {
                        if (!c->capturing) fused_gate_rollback(c, S);     // (counted by fused_gate_enter above, never enqueued)
                        c->err = std::string("k_post_chain_r launch: ") + hipGetErrorString(le);
                        return GSMCAL_E_HIP;
                    }
                }
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
        // Round 6: ... and so are the raw bytes of a call that reads ANOTHER buffer than the previous call did (a service fed by the
        // ingest ring, bench.py's batch rotated over four buffers: nothing of it is in the Infinity Cache) -- 64 streams rotated
        // over four buffers 0.1785 -> 0.1752 ms per step; the same buffer again keeps plain loads.
        const int nt = c->front_nt >= 0 ? c->front_nt : ((c->call_raw_bytes > ((size_t)256 << 20) || c->call_raw_fresh) ? 1 : 0);
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
    if (fft_len == 16 && 2 * S <= c->n_cu && c->snr_full && !c->no_fuse_now && len - (fft_len - 1) > nwin) {
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

// ---- calls in flight: streams, events, joining ---------------------------------------------------------------------------
// Two one-thread kernels that tell whether two streams sit on different hardware queues: the first spins (at most `ticks` of the
// 100 MHz wall clock) until the second has raised the flag.  On one queue the second is dispatched behind the first and the spin
// times out; on two queues it runs at once.
__global__ void k_probe_spin(unsigned* flag, unsigned* seen, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    unsigned v = 0;
    while ((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    *seen = v;
}
__global__ void k_probe_set(unsigned* flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

int streams_side_by_side(gsmcal_ctx* c, hipStream_t a, hipStream_t b, unsigned* d_words, bool* yes) {
    HIPCHK(c, hipMemsetAsync(d_words, 0, 2 * sizeof(unsigned), a));
    HIPCHK(c, hipStreamSynchronize(a));
    hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(1), 0, a, d_words, d_words + 1, 20000ull);     // 200 us at most
    hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(1), 0, b, d_words);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(a));
    HIPCHK(c, hipStreamSynchronize(b));
    unsigned seen = 0;
    HIPCHK(c, hipMemcpy(&seen, d_words + 1, sizeof(unsigned), hipMemcpyDeviceToHost));
    *yes = seen != 0;
    return 0;
}

// The streams calls in flight run on.  The runtime hands its (four) hardware queues to new streams least-used first, and what
// "least used" means depends on every stream the process has created so far: four streams created in a row can land on three
// queues (measured under PyTorch: two of them on one queue, 0.158 instead of 0.127 ms per 64-stream call four deep).  So the
// context creates a dozen candidates, keeps those it SEES running side by side with all it has kept so far (a 200-us probe per
// pair, once per context) and gives the rest back; slot s runs on kept stream s mod n_side.
int pipe_pick_streams(gsmcal_ctx* c) {
    constexpr int NC = 12;
    hipStream_t cand[NC] = {};
    // (work still queued on the context's stream would hold up a probe kernel that shares its hardware queue and make two good
    // streams look like one: the first call in flight of a context waits for it -- once)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < NC; ++i) HIPCHK(c, hipStreamCreateWithFlags(&cand[i], hipStreamNonBlocking));   // (default priority: a higher one for some slots measured no better, NOTES_r06)
    unsigned* d_words = nullptr;
    int rc = hipMalloc((void**)&d_words, 2 * sizeof(unsigned)) == hipSuccess ? 0 : GSMCAL_E_HIP;
    std::vector<int> kept;
    for (int i = 0; rc >= 0 && i < NC && (int)kept.size() < gsmcal_ctx::PIPE_MAX_DEPTH; ++i) {
        bool ok = true;
        for (int k : kept) {
            rc = streams_side_by_side(c, cand[k], cand[i], d_words, &ok);
            if (rc < 0 || !ok) break;
        }
        if (rc >= 0 && ok) kept.push_back(i);
    }
    if (d_words) (void)hipFree(d_words);
    if (rc < 0 || kept.empty()) {                          // (the probe itself failed: the first streams, as created)
        (void)hipGetLastError();
        kept.clear();
        for (int i = 0; i < 4; ++i) kept.push_back(i);
    }
    for (int i = 0; i < NC; ++i) {
        bool keep = false;
        for (int k : kept) keep = keep || k == i;
        if (keep) c->side_owned.push_back(cand[i]); else (void)hipStreamDestroy(cand[i]);
    }
    c->n_side = (int)c->side_owned.size();
    for (int s = 0; s < gsmcal_ctx::PIPE_MAX_DEPTH; ++s) c->side_stream[s] = c->side_owned[s % c->n_side];
    return 0;
}

int pipe_prepare(gsmcal_ctx* c) {
    if (c->n_side == 0) RET_IF(pipe_pick_streams(c));
    for (int s = 0; s < gsmcal_ctx::PIPE_MAX_DEPTH; ++s) {
        if (!c->side_in[s] && hipEventCreateWithFlags(&c->side_in[s], hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess) {
            (void)hipGetLastError();
            HIPCHK(c, hipEventCreateWithFlags(&c->side_in[s], hipEventDisableTiming));
        }
        // the end of a call also orders the table row stored in host memory -- a plain event
        if (!c->pipe_done[s]) HIPCHK(c, hipEventCreateWithFlags(&c->pipe_done[s], hipEventDisableTiming));
    }
    return 0;
}

// The stream the outputs of the most recent batch call are ordered on: the slot's stream while that call is in flight, else the
// context's stream.
hipStream_t pipe_out_stream(gsmcal_ctx* c) {
    if (c->pipe_last_slot < 0 || !c->pipe_pending[c->pipe_last_slot]) return c->stream;
    return c->side_stream[c->pipe_last_slot];
}

// The context's stream waits for every call still in flight (GPU-side waits, the host does not block): from here on the context
// behaves as at depth 1.  Every entry point but the batch calls that go in flight themselves starts with this.
int pipe_join(gsmcal_ctx* c) {
    for (int s = 0; s < gsmcal_ctx::PIPE_MAX_DEPTH; ++s)
        if (c->pipe_pending[s]) {
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->pipe_done[s], 0));
            c->pipe_pending[s] = false;
        }
    c->pipe_last_slot = -1;
    return 0;
}

// ... and the host waits too (gsmcal_sync, destruction).
int pipe_drain(gsmcal_ctx* c) {
    bool any = false;
    for (int s = 0; s < gsmcal_ctx::PIPE_MAX_DEPTH; ++s) any = any || c->pipe_pending[s];
    if (any)
        for (hipStream_t st : c->side_owned) HIPCHK(c, hipStreamSynchronize(st));
    return pipe_join(c);
}

#define ENTER(c)                                   \
    do {                                           \
        HIPCHK(c, hipSetDevice((c)->device));      \
        RET_IF(pipe_join(c));                      \
    } while (0)

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
    // a captured plan with a fused tail in it passes the gate at every replay; when another context's fused tail is in flight this
    // call is enqueued eagerly instead (run_fine then takes the four-launch tail)
    auto replay = [&]() -> int {
        if (slot.fused_streams > 0) {
            bool ok;
            {
                std::lock_guard<std::mutex> lk(fused_gate().mu);
                ok = fused_gate_pass(c, slot.fused_streams);
            }
            if (!ok) return enqueue();
        }
        const hipError_t ge = hipGraphLaunch(slot.exec, c->stream);
        if (ge != hipSuccess) {
            if (slot.fused_streams > 0) fused_gate_rollback(c, slot.fused_streams);     // (counted above, never enqueued)
            c->err = std::string("hipGraphLaunch: ") + hipGetErrorString(ge);
            return GSMCAL_E_HIP;
        }
        return 0;
    };
    if (same && slot.exec) return replay();
    if (!same) {
        if (slot.exec) { (void)hipGraphExecDestroy(slot.exec); slot.exec = nullptr; }
        if (slot.graph) { (void)hipGraphDestroy(slot.graph); slot.graph = nullptr; }
        slot.seen = 0;
        slot.fused_streams = 0;
    }
    if (same && slot.seen >= 1) {
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
            (void)hipGetLastError();
            c->use_graph = false;                       // this stream cannot be captured: eager launches for good
            return enqueue();
        }
        c->capturing = true;
        c->capture_fused_streams = 0;
        const int rc = enqueue();
        c->capturing = false;
        slot.fused_streams = c->capture_fused_streams;
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
        return replay();
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

