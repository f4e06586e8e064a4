// abi_report.h -- the reference's console diagnostics (FCCH_coarse_position.m:6,28,92-94, FCCH_fine_correction.m:6,13,66,
// 96-99,116,156-161,190-193, SCH_corr_rate_correction.m:6,12,60,80,107-110,118, carrier_correct_post_SCH.m:6,11,17,73-79) as TEXT:
// the per-function entry points of abi_calls.h have the stream's state on the host anyway and leave the lines the .m file would
// have disp()ed in the context; gsmcal_last_call_report() hands them to the MEX gateway (mexPrintf) or the Python mirror.
// Included by gsmcal.hip ahead of abi_calls.h (host code only).
#pragma once
namespace {

// MATLAB's num2str for a real row vector: integers as %{w}d with w = digits + 2 (+ 1 if any is negative), everything else as
// %{d+7}.{d}g with d = max(floor(log10(max|x|)) + 5, 5) capped at 16; the pieces are concatenated and the result trimmed.
std::string rep_num2str(const double* x, int n) {
    if (n <= 0) return std::string();
    bool ints = true, neg = false;
    double mx = 0.0;
    for (int i = 0; i < n; ++i) {
        if (x[i] != floor(x[i]) || !std::isfinite(x[i])) ints = false;
        if (x[i] < 0.0) neg = true;
        if (std::isfinite(x[i]) && fabs(x[i]) > mx) mx = fabs(x[i]);
    }
    char fmt[32], buf[64];
    std::string out;
    const int lg = mx > 0.0 ? (int)floor(log10(mx)) : 0;
    if (ints) {
        const int w = (lg > 0 ? lg : 0) + (neg ? 1 : 0) + 3;
        snprintf(fmt, sizeof(fmt), "%%%d.0f", w);
    } else {
        int d = lg + 5;
        if (d < 5) d = 5;
        if (d > 16) d = 16;
        snprintf(fmt, sizeof(fmt), "%%%d.%dg", d + 7 + (neg ? 1 : 0), d);
    }
    for (int i = 0; i < n; ++i) {
        if (std::isnan(x[i])) snprintf(buf, sizeof(buf), "%*s", ints ? 5 : 12, "NaN");
        else if (std::isinf(x[i])) snprintf(buf, sizeof(buf), "%*s", ints ? 5 : 12, x[i] > 0 ? "Inf" : "-Inf");
        else snprintf(buf, sizeof(buf), fmt, x[i]);
        out += buf;
    }
    const size_t a = out.find_first_not_of(' ');
    const size_t b = out.find_last_not_of(' ');
    return a == std::string::npos ? std::string() : out.substr(a, b - a + 1);
}
std::string rep_num2str(double v) { return rep_num2str(&v, 1); }
std::string rep_diff(const double* x, int n) {
    std::vector<double> d(n > 1 ? n - 1 : 0);
    for (int i = 0; i + 1 < n; ++i) d[i] = x[i + 1] - x[i];
    return rep_num2str(d.data(), (int)d.size());
}

// the "Kinds of pos diff more than 2" block (FCCH_fine_correction.m:96-99 / SCH_corr_rate_correction.m:107-110)
void rep_spacing(std::string& r, const char* who, const double* pos, int n, int ov, double max_ppm) {
    const double d = 10.0 * 1250.0 * ov, d1 = 11.0 * 1250.0 * ov;
    const double max_th = floor(d * max_ppm * 1e-6), max_th1 = floor(d1 * max_ppm * 1e-6);
    std::vector<double> a(n > 1 ? n - 1 : 0), b(a.size());
    double na = 0, nb = 0;
    for (int i = 0; i + 1 < n; ++i) {
        a[i] = fabs(pos[i + 1] - pos[i] - d); b[i] = fabs(pos[i + 1] - pos[i] - d1);
        na += a[i] < max_th; nb += b[i] < max_th1;
    }
    const double cnt[2] = {na, nb};
    r += std::string(who) + " Warning! Kinds of pos diff more than 2!\n";
    r += "Expected len " + rep_num2str((double)(n - 1)) + ". Actual " + rep_num2str(cnt, 2) + "\n";
    r += "diff intra multiframe max th " + rep_num2str(max_th) + " actual " + rep_num2str(a.data(), (int)a.size()) + "\n";
    r += "diff inter multiframe max th " + rep_num2str(max_th1) + " actual " + rep_num2str(b.data(), (int)b.size()) + "\n";
}

// the tone estimates of nb bursts (FCCH_fine_correction.m:156-161 / carrier_correct_post_SCH.m:73-79)
void rep_tone(std::string& r, const char* who, const double* fo, int nb, double carrier_ppm) {
    double m = 0.0;
    for (int i = 0; i < nb; ++i) m += fo[i];
    m /= (double)nb;
    r += std::string(who) + " FCCH freq " + rep_num2str(fo, nb) + "\n";
    r += std::string(who) + " mean FCCH freq " + rep_num2str(m) + "\n";
    r += std::string(who) + " carrier error ppm " + rep_num2str(carrier_ppm) + "\n";
}

std::string report_coarse(const StreamState& st) {
    std::string r = " \n";
    if (st.n_coarse == 0) return r + "FCCH coarse: No FCCH found!\n";
    r += "FCCH coarse: hit successive " + rep_num2str((double)st.n_coarse) + " FCCH. pos " + rep_num2str(st.coarse_pos, st.n_coarse) + "\n";
    r += "FCCH coarse: pos diff " + rep_diff(st.coarse_pos, st.n_coarse) + "\n";
    r += "FCCH coarse: SNR " + rep_num2str(st.coarse_snr, st.n_coarse) + "\n";
    return r;
}

std::string report_fine(const StreamState& st, int ov, double max_ppm, double gate_db) {
    std::string r = " \n";
    const int code = st.stage_status[0];
    if (code == GSMCAL_S_FEW_HITS) return r + "FCCH fine: Warning! Length of hits is smaller than 5!\n";
    r += "FCCH fine: first round diff " + rep_diff(st.fine_first, st.n_fine) + "\n";
    if (code == GSMCAL_S_FINE_FEW) return r;
    if (code == GSMCAL_S_FINE_SPACING) { rep_spacing(r, "FCCH fine:", st.fine_first, st.n_fine, ov, max_ppm); return r; }
    r += "FCCH fine: sampling error ppm " + rep_num2str(st.sampling_ppm1) + "\n";
    if (st.r1_kind != 3) return r;                               // fewer than 5 bursts left: the carrier block is skipped (:142)
    int nb = st.fcch_is_sentinel ? 0 : st.n_fcch;
    if (nb == 0) for (nb = 0; nb < st.n_fine && nb < MAXH && st.fo_burst[nb] != 0.0; ++nb) {}     // (the gate zeroed n_fcch: the bursts that were estimated)
    rep_tone(r, "FCCH fine:", st.fo_burst, nb, st.carrier_ppm1);
    r += "FCCH fine: SNR " + rep_num2str(st.snr_burst, nb) + "\n";
    bool low = false;
    for (int i = 0; i < nb; ++i) low = low || st.snr_burst[i] < gate_db;
    if (low) r += "FCCH fine: Warning! Some FCCH SNR seems pretty low!\n";
    return r;
}

std::string report_sch(const StreamState& st, int ov, double max_ppm) {
    std::string r = " \n";
    const int code = st.stage_status[1];
    if (code == GSMCAL_S_FEW_HITS) return r + "SCH: Warning! Length of hits is smaller than 5!\n";
    if (code == GSMCAL_S_SCH_EDGE) return r + "SCH:  Warning! No peak around base position is found!\n";
    r += "SCH: first round diff " + rep_diff(st.sch_first, st.n_sch_first) + "\n";
    if (code == GSMCAL_S_SCH_FEW) return r;
    if (code == GSMCAL_S_SCH_SPACING) { rep_spacing(r, "SCH:", st.sch_first, st.n_sch_first, ov, max_ppm); return r; }
    r += "SCH: sampling error ppm " + rep_num2str(st.sampling_ppm2) + "\n";
    return r;
}

std::string report_post(const StreamState& st) {
    std::string r = " \n";
    const int code = st.stage_status[2];
    if (code == GSMCAL_S_POST_NO_POS) return r + "post SCH: Warning! No valid position information!\n";
    if (code == GSMCAL_S_POST_FEW_BCCH) return r + "post SCH: Warning! The number of BCCH bursts is less than 4!\n";
    if (st.r3_kind != 3) return r;
    int nb = 0;
    for (int i = 0; i < st.n_rows && i < MAXROWS; ++i) nb += st.pos_info[MAXROWS + i] == 0.0;
    if (nb > MAXH) nb = MAXH;
    if (nb > 0) rep_tone(r, "post SCH:", st.fo_burst, nb, st.carrier_ppm2);
    return r;
}

}  // namespace
