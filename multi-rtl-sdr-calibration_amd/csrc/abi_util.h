// abi_util.h -- synthetic-input expansion on the device (bench / tests) and the development-build phase timing.
// Included by gsmcal.hip inside its extern "C" block.
#pragma once
// ---- synthetic-input utility ---------------------------------------------------------------------------
int gsmcal_synth_expand_dev(gsmcal_ctx* c, const uint8_t* d_base, int k, long n, uint8_t* d_out, long d, long first_unit,
                            unsigned long long seed) {
    if (!c || !d_base || !d_out || k < 1 || n < 1 || d < 1 || first_unit < 0) return GSMCAL_E_ARG;
    ENTER(c);
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
    ENTER(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<StreamState> v((size_t)c->last_S);
    if (c->n_lanes_used <= 1 && c->lanes[0].n == 0) { c->lanes[0].lo = 0; c->lanes[0].n = c->last_S; }
    for (int i = 0; i < c->n_lanes_used; ++i) {
        const Lane& L = c->detail_lane ? *c->detail_lane : c->lanes[i];     // (a pipelined call: the workspace of its slot)
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
    ENTER(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < c->n_lanes_used; ++i) {
        const Lane& L = c->detail_lane ? *c->detail_lane : c->lanes[i];
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
                if (!r[i] || r[i] < r[0]) continue;          // absent, or left by an earlier launch that used this block's slot
                if (r[i] > t1) t1 = r[i];
                // stamps are numbered by code path, not by time: one taken EARLIER than its predecessor starts a new chain
                // (no phase is reported across the break -- the unsigned difference used to print as 1.8e17 us)
                if (r[i] >= prev) { ph[i - 1] += (double)(r[i] - prev) / 100.0; ++pc[i - 1]; }
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
                if (r[0] && r[i] && r[i] >= r[0]) { a += ((double)r[i] - (double)r[0]) / 100.0; ++n; }
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

