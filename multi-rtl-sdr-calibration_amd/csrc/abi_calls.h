// abi_calls.h -- the C ABI entry points of include/gsmcal.h for the DSP chain: context, parameters, the per-function
// MATLAB-signature calls (a1..a9, f4) and the batched hot path.  Included by gsmcal.hip inside its extern "C" block.
#pragma once

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
    RET_IF(pipe_join(c));
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
    const char* sst = getenv("GSMCAL_SCAN_STAGES");
    if (sst && atoi(sst) >= 1) c->scan_stages = atoi(sst);
    const char* lm = getenv("GSMCAL_LANE_MIN");
    if (lm && atoi(lm) >= 1) c->lane_min = atoi(lm);
    const char* ce = getenv("GSMCAL_CERT");
    if (ce) c->certify = atoi(ce) != 0;
    const char* sfe = getenv("GSMCAL_SNR_FULL");
    if (sfe) c->snr_full = atoi(sfe) != 0;
    const char* sse = getenv("GSMCAL_SNR_SCREEN_DB");
    if (sse) c->snr_screen_db = atof(sse);
    const char* fge = getenv("GSMCAL_FUSE_GATHER");
    if (fge) c->fuse_fine_gather = atoi(fge) != 0;
    const char* fpe = getenv("GSMCAL_FUSE_POST");
    if (fpe) c->fuse_post = atoi(fpe) != 0;
    const char* pse = getenv("GSMCAL_POST_SLOTS");
    if (pse && atoi(pse) >= 1) c->post_slots_cap = atoi(pse);
    const char* pe = getenv("GSMCAL_PRESCREEN");
    if (pe && atoi(pe) == 0) c->prescreen = false;
    const char* fg = getenv("GSMCAL_FRONT_GENERIC");
    if (fg && atoi(fg) != 0) c->front_generic = true;
    if (const char* e2 = getenv("GSMCAL_FUSED_POLL_S")) { if (atof(e2) > 0.0) c->fused_poll_s = atof(e2); }
    if (const char* e2 = getenv("GSMCAL_TEST_FUSED_STALL")) c->test_stall = atoi(e2);
    const char* ge = getenv("GSMCAL_GRAPH");
    if (ge && atoi(ge) == 0) c->use_graph = false;
    if (ge && atoi(ge) == 2) c->graph_always = true;
    fused_gate_register(c);
    *out = c;
    return 0;
}

int gsmcal_ctx_create(int device_id, gsmcal_ctx** out) {
    int r = gsmcal_ctx_create_on_stream(device_id, nullptr, out);
    if (r != 0) return r;
    gsmcal_ctx* c = *out;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        fused_gate_unregister(c);
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
    (void)pipe_drain(c);
    (void)hipStreamSynchronize(c->stream);
    (void)hipDeviceSynchronize();
    fused_gate_unregister(c);
    DevBuf* bufs[] = {&c->coef, &c->ts, &c->cf, &c->table, &c->snrhit, &c->arr_in, &c->arr_out, &c->posinfo, &c->rlen,
                      &c->misc, &c->tw, &c->csum_head, &c->tw_sch};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (int i = 0; i < MAX_LANES; ++i) {
        Lane& L = c->lanes[i];
        DevBuf* lb[] = {&L.state, &L.dec, &L.win, &L.peaks, &L.snrbuf, &L.x0, &L.chunkrec, &L.openlist, &L.cert, &L.partial, &L.tailctr, &L.xch, &L.xepoch};
        for (DevBuf* b : lb)
            if (b->p) (void)hipFree(b->p);
        if (L.done) (void)hipEventDestroy(L.done);
        if (L.front_done) (void)hipEventDestroy(L.front_done);
        if (i > 0 && L.stream) (void)hipStreamDestroy(L.stream);
    }
    for (int i = 0; i < gsmcal_ctx::PIPE_MAX_DEPTH; ++i) {
        Lane& L = c->pipe[i];
        DevBuf* lb[] = {&L.state, &L.dec, &L.win, &L.peaks, &L.snrbuf, &L.x0, &L.chunkrec, &L.openlist, &L.cert, &L.partial, &L.tailctr, &L.xch, &L.xepoch};
        for (DevBuf* b : lb)
            if (b->p) (void)hipFree(b->p);
        if (c->pipe_done[i]) (void)hipEventDestroy(c->pipe_done[i]);
        if (c->side_in[i]) (void)hipEventDestroy(c->side_in[i]);
    }
    for (hipStream_t st : c->side_owned) (void)hipStreamDestroy(st);
    if (c->fork) (void)hipEventDestroy(c->fork);
    for (hipEvent_t e : c->ag_chain)
        if (e) (void)hipEventDestroy(e);
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

int gsmcal_fused_tail_stats(gsmcal_ctx* c, unsigned long long* fused_launches, unsigned long long* gate_fallbacks) {
    if (!c) return GSMCAL_E_ARG;
    if (fused_launches) *fused_launches = c->n_fused_launches;
    if (gate_fallbacks) *gate_fallbacks = c->n_gate_fallbacks;
    return 0;
}

// The context's streams have drained.  If a fused tail of the calls since the last check gave up waiting for a peer (the kernel's
// poll limit: another process's fused tail held the slots its workgroups needed), run those calls again -- same inputs, which the
// caller leaves untouched until it has synchronised -- with the four-launch tail, whose kernels wait for nobody.
static int fused_recover(gsmcal_ctx* c) {
    if (!c->fused_done || c->recovering) return 0;
    volatile unsigned* flag = c->fused_done + gsmcal_ctx::FUSED_MAX_STREAMS;
    if (*flag == 0u) { c->fused_calls.clear(); return 0; }
    *flag = 0u;
    std::vector<gsmcal_ctx::FusedCall> calls;
    calls.swap(c->fused_calls);
    const bool fp = c->fuse_post;
    const int depth = c->pipe_depth;
    c->fuse_post = false; c->pipe_depth = 1; c->recovering = true;
    int rc = 0;
    for (auto& k : calls) {
        rc = gsmcal_calibrate_batch_dev(c, k.d_raw, k.d, k.n, k.coef.data(), k.ntaps, k.ts.data(), k.len_ts, k.cf.data(), k.d_table, k.d_pos_info, k.d_r_correct, k.d_r_len);
        if (rc < 0) break;
    }
    hipError_t e = hipSuccess;
    if (rc >= 0) e = hipStreamSynchronize(c->stream);
    c->fuse_post = fp; c->pipe_depth = depth; c->recovering = false;
    ++c->n_tail_reruns;
    if (rc < 0) return rc;
    if (e != hipSuccess) { c->err = std::string("hipStreamSynchronize (fused tail re-run): ") + hipGetErrorString(e); return GSMCAL_E_HIP; }
    return 0;
}

int gsmcal_sync(gsmcal_ctx* c) {
    if (!c) return GSMCAL_E_ARG;
    RET_IF(pipe_drain(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return fused_recover(c);
}

long long gsmcal_fused_tail_reruns(gsmcal_ctx* c) { return c ? (long long)c->n_tail_reruns : (long long)GSMCAL_E_ARG; }

// ---- pipelined batch calls ----------------------------------------------------------------------------------------
int gsmcal_ctx_set_pipeline_depth(gsmcal_ctx* c, int depth) {
    if (!c || depth < 1 || depth > gsmcal_ctx::PIPE_MAX_DEPTH) return GSMCAL_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    RET_IF(pipe_drain(c));
    if (depth > 1) RET_IF(pipe_prepare(c));
    c->pipe_depth = depth;
    return 0;
}

int gsmcal_ctx_get_pipeline_depth(gsmcal_ctx* c) { return c ? c->pipe_depth : GSMCAL_E_ARG; }
int gsmcal_ctx_pipeline_queues(gsmcal_ctx* c) { return c ? c->n_side : GSMCAL_E_ARG; }

const char* gsmcal_last_error(gsmcal_ctx* c) { return c ? c->err.c_str() : "null context"; }

long gsmcal_last_call_report(gsmcal_ctx* c, char* buf, size_t cap) {
    if (!c) return GSMCAL_E_ARG;
    if (buf && cap > 0) {
        const size_t n = c->report.size() < cap - 1 ? c->report.size() : cap - 1;
        memcpy(buf, c->report.data(), n);
        buf[n] = 0;
    }
    return (long)c->report.size();
}

long gsmcal_num2str(const double* x, int n, char* buf, size_t cap) {
    if (!x || n < 0) return GSMCAL_E_ARG;
    const std::string s = rep_num2str(x, n);
    if (buf && cap > 0) {
        const size_t m = s.size() < cap - 1 ? s.size() : cap - 1;
        memcpy(buf, s.data(), m);
        buf[m] = 0;
    }
    return (long)s.size();
}

int gsmcal_dev_alloc(gsmcal_ctx* c, size_t bytes, void** dptr) {
    if (!c || !dptr) return GSMCAL_E_ARG;
    ENTER(c);
    HIPCHK(c, hipMalloc(dptr, bytes));
    return 0;
}
int gsmcal_dev_free(gsmcal_ctx* c, void* dptr) {
    if (!c) return GSMCAL_E_ARG;
    RET_IF(pipe_drain(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(dptr));
    return 0;
}
int gsmcal_memcpy_h2d(gsmcal_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) return GSMCAL_E_ARG;
    RET_IF(pipe_join(c));
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int gsmcal_memcpy_d2h(gsmcal_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) return GSMCAL_E_ARG;
    RET_IF(pipe_join(c));
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int gsmcal_profile_enable(gsmcal_ctx* c, int enable) {
    if (!c) return GSMCAL_E_ARG;
    RET_IF(pipe_drain(c));
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
    ENTER(c);
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
    ENTER(c);
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
    ENTER(c);
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
    c->report.clear();
    RET_IF(coarse_api(c, s, len, a, &st));
    if (st.status < 0) return st.status;
    c->report = report_coarse(st);
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
    ENTER(c);
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
    c->report.clear();
    RET_IF(run_fine(c, 1, src, 0, g, H, false, -1, 0));
    RET_IF(fetch_states(c, 1, v));
    const StreamState& st = v[0];
    if (st.status < 0) return st.status;
    c->report = report_fine(st, ov, c->params.fine_max_ppm, c->params.fine_gate_snr_db);
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
    ENTER(c);
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
    c->report.clear();
    RET_IF(run_sch(c, 1, src, 0, g, H, len_ts, false, -1));
    RET_IF(fetch_states(c, 1, v));
    const StreamState& st = v[0];
    if (st.status < 0) return st.status;
    c->report = report_sch(st, ov, c->params.sch_max_ppm);
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
    ENTER(c);
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
    c->report.clear();
    RET_IF(run_post(c, 1, src, 0, g, nfcch > 0 ? nfcch : 1, false, nullptr, nullptr, nullptr));
    RET_IF(fetch_states(c, 1, v));
    const StreamState& st = v[0];
    if (st.status < 0) return st.status;
    c->report = report_post(st);
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
    ENTER(c);
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
    ENTER(c);
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
    ENTER(c);
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
    // Calls in flight (gsmcal_ctx_set_pipeline_depth > 1): a single-stage scanner batch (up to 1 199 captures)
    // runs on internal stream i mod depth in workspace i mod depth, so the detector of call i -- one resident round of latency-bound
    // workgroups -- sits underneath the bandwidth-bound front kernel of call i+1 (multi_rtl_sdr_gsm_FCCH_scanner.m:132-135,163-186
    // over consecutive sweeps).  Same semantics as for calibration calls: outputs complete at call i + depth / gsmcal_sync / any other
    // entry point.
    bool pipelined = c->pipe_depth > 1 && !c->prof && c->stream != nullptr && plan_lanes(c, d, false) == 1;
    if (pipelined) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(c->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); pipelined = false; }
    }
    const bool same_taps = (int)c->h_coef.size() == ntaps && memcmp(c->h_coef.data(), coef, (size_t)ntaps * sizeof(double)) == 0 && c->head_epoch == c->coef_epoch;
    if (!pipelined || !same_taps) RET_IF(pipe_join(c));
    c->cur = &c->lanes[0];
    c->detail_lane = nullptr;
    c->call_raw_bytes = (size_t)d * 2 * (size_t)n;
    c->call_raw_fresh = (const void*)d_raw != c->last_raw;
    c->last_raw = d_raw;
    RET_IF(upload_cached(c, c->coef, c->h_coef, coef, ntaps));
    RET_IF(ensure_head(c, decim));
    if (pipelined) {
        RET_IF(pipe_prepare(c));
        const int slot = (int)(c->pipe_calls % (unsigned long)c->pipe_depth);
        Lane& L = c->pipe[slot];
        if (c->pipe_pending[slot]) {
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->pipe_done[slot], 0));
            c->pipe_pending[slot] = false;
        }
        HIPCHK(c, hipEventRecord(c->side_in[slot], c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->side_stream[slot], c->side_in[slot], 0));
        L.stream = c->side_stream[slot];
        L.lo = 0; L.n = d;
        c->cur = &L;
        c->n_lanes_used = 1;
        int rc = ensure(c, L.dec, (size_t)d * nd * sizeof(cplx));
        if (rc >= 0) rc = front_fused(c, d_raw, d, n, (const double*)c->coef.p, ntaps, decim, (cplx*)L.dec.p, nd, 0, d);
        if (rc >= 0) {
            ScanAccept acc;
            acc.snr_numhit = d_snr_numhit; acc.positions = d_positions; acc.pos_snr = d_pos_snr; acc.counts = d_counts;
            rc = coarse(c, d, (const cplx*)L.dec.p, nd, nd, dec_ratio, 0, true, n, decim, &acc, true);
        }
        if (hipEventRecord(c->pipe_done[slot], L.stream) != hipSuccess) { c->err = "hipEventRecord (calls in flight)"; rc = GSMCAL_E_HIP; }
        c->pipe_pending[slot] = true;
        c->pipe_last_slot = slot;
        ++c->pipe_calls;
        c->detail_lane = &L;
        c->cur = &c->lanes[0];
        c->last_S = d;
        if (rc < 0) { (void)pipe_drain(c); return rc; }
        return 0;
    }
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
        RET_IF(coarse(c, S, (const cplx*)L.dec.p, nd, nd, dec_ratio, 0, true, n, decim, &acc, nl == 1 || S <= 3 * c->n_cu));
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
    ENTER(c);
    RET_IF(ensure(c, c->misc, (size_t)2 * n * d));
    RET_IF(ensure(c, c->snrhit, (size_t)d * (2 + 2 * MAXH) * sizeof(double) + (size_t)d * sizeof(int)));
    HIPCHK(c, hipMemcpyAsync(c->misc.p, raw, (size_t)2 * n * d, hipMemcpyHostToDevice, c->stream));
    double* d_sn = (double*)c->snrhit.p;
    double* d_pos = d_sn + (size_t)2 * d;
    double* d_ps = d_pos + (size_t)d * MAXH;
    int* d_cnt = (int*)(d_ps + (size_t)d * MAXH);
    RET_IF(gsmcal_fcch_scan_batch_dev(c, (const uint8_t*)c->misc.p, d, n, coef, ntaps, d_sn, d_pos, d_ps, d_cnt));
    RET_IF(pipe_join(c));                      // (a pipelined context: the copies below wait for this call like for any other)
    std::vector<double> sn((size_t)2 * d);
    HIPCHK(c, hipMemcpyAsync(sn.data(), d_sn, sn.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (positions) HIPCHK(c, hipMemcpyAsync(positions, d_pos, (size_t)d * MAXH * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (pos_snr) HIPCHK(c, hipMemcpyAsync(pos_snr, d_ps, (size_t)d * MAXH * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (counts) HIPCHK(c, hipMemcpyAsync(counts, d_cnt, (size_t)d * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < d; ++i) { snr[i] = sn[2 * i]; num_hit[i] = sn[2 * i + 1]; }
    return 0;
}

// r_correct of the S streams of the current lane (gsm_sync_demod.m:118-120 hand it to SCH_demod): one launch on the lane's stream
static int launch_r_correct(gsmcal_ctx* c, Lane& L, const Source& src, const uint8_t* raw_i, int S, long n, int ntaps, double* d_r_correct_lane) {
    StreamTileArgs ta;
    ta.raw = raw_i; ta.raw_stride = 2 * n; ta.coef = (const double*)c->coef.p; ta.ntaps = ntaps;
    ta.dst = (cplx*)d_r_correct_lane; ta.dst_stream_stride = n;
    const size_t tlds = stream_tile_lds(ntaps);
    bool sym = (int)c->h_coef.size() == ntaps;         // exactly mirrored taps (what fir1 returns)
    for (int k = 0; sym && k < ntaps / 2; ++k) sym = c->h_coef[k] == c->h_coef[ntaps - 1 - k];
    if (ntaps == 47 && sym) {                           // the drivers' filter: taps in registers, every sample read once
        LAUNCH(c, k_stream_tile_s47<ST47_TILE>, dim3((unsigned)((n + (long)ST47_TILE * ST_TPB - 1) / ((long)ST47_TILE * ST_TPB)), S), dim3(ST_THREADS), stream_tile_s47_lds(), (const StreamState*)L.state.p, ta);
        CHECK_LAUNCH(c);
    } else if (tlds <= 64 * 1024) {
        LAUNCH_GEOM(ta.ntaps == 47, c, (k_stream_tile<47>), (k_stream_tile<0>), dim3((unsigned)((n + (long)ST_TILE * ST_TPB - 1) / ((long)ST_TILE * ST_TPB)), S), dim3(ST_THREADS), tlds, (const StreamState*)L.state.p, ta);
        CHECK_LAUNCH(c);
    } else {                                        // very long filters: the general tile gather
        const int tiles = (int)((n + TILE - 1) / TILE);
        RET_IF(launch_gather(c, S, src, 4, TILE, true, tiles, (cplx*)d_r_correct_lane, n, 0));
    }
    return 0;
}

// (a call that took the fused tail: what fused_recover needs to run it again; the inputs are the context's cached copies)
static void record_fused_call(gsmcal_ctx* c, const uint8_t* d_raw, int d, long n, int ntaps, int len_ts, double* d_table, double* d_pos_info,
                              double* d_r_correct, long* d_r_len) {
    if (c->recovering) return;
    if (c->fused_calls.size() >= 16) c->fused_calls.erase(c->fused_calls.begin());      // (a caller that never synchronises through the library)
    c->fused_calls.push_back({d_raw, d, n, ntaps, len_ts, c->h_coef, c->h_ts, c->h_cf, d_table, d_pos_info, d_r_correct, d_r_len});
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
    // Calls in flight (gsmcal_ctx_set_pipeline_depth > 1): calls that run on one lane.  Anything that would touch what the calls in
    // flight still read (new taps / training sequence / carrier frequencies, a workspace that must grow, the twiddle table) joins
    // them into the context's stream first.
    const bool want_pipe = c->pipe_depth > 1 && c->stream != nullptr && plan_lanes(c, d) == 1;
    bool pipelined = want_pipe && !c->prof;
    if (pipelined) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(c->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); pipelined = false; }
    }
    // (events on every dispatch -- gsmcal_profile_enable -- or the caller's own stream capture: one call at a time, but with the
    // kernels the calls in flight run, so that a per-kernel profile of a depth-4 context describes what the depth-4 loop launches)
    const bool same_kernels_unpipelined = want_pipe && !pipelined;
    const bool same_inputs = (int)c->h_coef.size() == ntaps && memcmp(c->h_coef.data(), coef, (size_t)ntaps * sizeof(double)) == 0 &&
                             c->h_ts.size() == (size_t)2 * len_ts && memcmp(c->h_ts.data(), sch_ts, (size_t)2 * len_ts * sizeof(double)) == 0 &&
                             (int)c->h_cf.size() == d && memcmp(c->h_cf.data(), carrier_freq, (size_t)d * sizeof(double)) == 0 &&
                             c->tw_n == g.nfft && c->head_epoch == c->coef_epoch;
    if (!pipelined || !same_inputs) RET_IF(pipe_join(c));
    c->cur = &c->lanes[0];
    c->detail_lane = nullptr; c->cf_lane = nullptr;       // (also what an earlier call that failed half-way may have left set)
    c->no_fuse_now = same_kernels_unpipelined;
    c->call_raw_bytes = (size_t)d * 2 * (size_t)n;
    c->call_raw_fresh = (const void*)d_raw != c->last_raw;
    c->last_raw = d_raw;
    RET_IF(upload_cached(c, c->coef, c->h_coef, coef, ntaps));
    RET_IF(upload_cached(c, c->ts, c->h_ts, sch_ts, (size_t)2 * len_ts));
    RET_IF(upload_cached(c, c->cf, c->h_cf, carrier_freq, d));
    RET_IF(ensure_head(c, decim));
    RET_IF(ensure_twiddles(c, g.nfft));
    if (pipelined) {
        // slot = workspace and stream of this call; the context's stream first waits for the call that used it `depth` calls ago
        // (that call's outputs are complete in the context's stream order from here on), then the slot's stream takes the call behind
        // whatever the context's stream holds now: front end + coarse detector (:107,110,117), fine search (:118), four-launch tail
        RET_IF(pipe_prepare(c));
        const int slot = (int)(c->pipe_calls % (unsigned long)c->pipe_depth);
        Lane& L = c->pipe[slot];
        if (c->pipe_pending[slot]) {
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->pipe_done[slot], 0));
            c->pipe_pending[slot] = false;
        }
        c->cur = &L;
        c->n_lanes_used = 1;
        L.lo = 0; L.n = d;
        HIPCHK(c, hipEventRecord(c->side_in[slot], c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->side_stream[slot], c->side_in[slot], 0));
        L.stream = c->side_stream[slot];
        c->no_fuse_now = true;
        RET_IF(ensure(c, L.dec, (size_t)d * nd * sizeof(cplx)));
        RET_IF(front_fused(c, d_raw, d, n, (const double*)c->coef.p, ntaps, decim, (cplx*)L.dec.p, nd));
        RET_IF(coarse(c, d, (const cplx*)L.dec.p, nd, nd, dec_ratio, ov, true, n, decim));
        Source src{SRC_RAW, d_raw, 2 * n, nullptr, 0, (const double*)c->coef.p, ntaps};
        c->cf_lane = (const double*)c->cf.p;
        ChainOut co{d_table, d_pos_info, d_r_len, false};
        int rc = run_fine(c, d, src, 0, g, H, true, 2, len_ts, &co);
        if (rc >= 0 && !co.fused) {
            rc = run_sch(c, d, src, 2, g, H, len_ts, true, 3);
            if (rc >= 0) rc = run_post(c, d, src, 3, g, H, true, co.table, co.pos_info_out, co.r_len_out);
        }
        if (rc >= 0 && d_r_correct) rc = launch_r_correct(c, L, src, d_raw, d, n, ntaps, d_r_correct);
        c->cf_lane = nullptr; c->no_fuse_now = false;
        if (hipEventRecord(c->pipe_done[slot], L.stream) != hipSuccess) { c->err = "hipEventRecord (calls in flight)"; rc = GSMCAL_E_HIP; }
        c->pipe_pending[slot] = true;
        c->pipe_last_slot = slot;
        ++c->pipe_calls;
        c->detail_lane = &L;
        c->cur = &c->lanes[0];
        c->last_S = d;
        if (rc < 0) { (void)pipe_drain(c); return rc; }
        return 0;
    }
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
        // Measured (round 4, NOTES_r04.md): 128 / 256 / 512 / 1 024 / 2 048 streams 0.345 / 0.562 / 1.002 / 1.807 / 3.629 ms staggered
        // against 0.349 / 0.554 / 1.002 / 1.874 / 3.715 together: worth it from 256 streams per lane on.
        const bool stagger = nl > 1 && d / nl >= 256;   // (one decision for all lanes of the call)
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
        if (d_r_correct) RET_IF(launch_r_correct(c, L, src, raw_i, S, n, ntaps, d_r_correct + (size_t)2 * lo * n));
    }
    c->cf_lane = nullptr;
    RET_IF(join_lanes(c, nl));
    return 0;
    };
    const unsigned long long fused_before = c->n_fused_launches;
    RET_IF(run_maybe_graph(c, pick_slot(c, c->g_calib, key), key, enqueue, plan_lanes(c, d) > 1));
    c->no_fuse_now = false;
    if (c->n_fused_launches != fused_before) record_fused_call(c, d_raw, d, n, ntaps, len_ts, d_table, d_pos_info, d_r_correct, d_r_len);
    plan_lanes(c, d);          // lane bookkeeping for gsmcal_last_batch_details (a replay does not run `enqueue`)
    c->cur = &c->lanes[0];
    c->last_S = d;
    return 0;
}

int gsmcal_calibrate_batch(gsmcal_ctx* c, const uint8_t* raw, int d, long n, const double* coef, int ntaps,
                           const double* sch_ts, int len_ts, const double* carrier_freq, double* table,
                           double* pos_info, double* r_correct, long* r_len) {
    if (!c || !raw || !table || d < 1 || n < 1) return GSMCAL_E_ARG;
    ENTER(c);
    RET_IF(ensure(c, c->misc, (size_t)2 * n * d));
    RET_IF(ensure(c, c->table, (size_t)d * GSMCAL_TABLE_COLS * sizeof(double)));
    RET_IF(ensure(c, c->posinfo, (size_t)d * 2 * MAXROWS * sizeof(double)));
    RET_IF(ensure(c, c->rlen, (size_t)d * sizeof(long)));
    if (r_correct) RET_IF(ensure(c, c->arr_out, (size_t)d * n * sizeof(cplx)));
    HIPCHK(c, hipMemcpyAsync(c->misc.p, raw, (size_t)2 * n * d, hipMemcpyHostToDevice, c->stream));
    RET_IF(gsmcal_calibrate_batch_dev(c, (const uint8_t*)c->misc.p, d, n, coef, ntaps, sch_ts, len_ts, carrier_freq,
                                      (double*)c->table.p, (double*)c->posinfo.p,
                                      r_correct ? (double*)c->arr_out.p : nullptr, (long*)c->rlen.p));
    RET_IF(pipe_join(c));                      // (a pipelined context: the copies below wait for this call like for any other)
    if (!c->fused_calls.empty()) {             // (the fused tail ran: had it timed out, the call runs again as four launches before anything is copied)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        RET_IF(fused_recover(c));
    }
    HIPCHK(c, hipMemcpyAsync(table, c->table.p, (size_t)d * GSMCAL_TABLE_COLS * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (pos_info)
        HIPCHK(c, hipMemcpyAsync(pos_info, c->posinfo.p, (size_t)d * 2 * MAXROWS * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (r_len) HIPCHK(c, hipMemcpyAsync(r_len, c->rlen.p, (size_t)d * sizeof(long), hipMemcpyDeviceToHost, c->stream));
    if (r_correct)
        HIPCHK(c, hipMemcpyAsync(r_correct, c->arr_out.p, (size_t)d * n * sizeof(cplx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

