function make_reference_vectors(ref_dir, cap_dir, out_file)
% MAKE_REFERENCE_VECTORS  Run the REFERENCE's own functions on this repository's seeded captures and dump per-stage outputs.
%
%   make_reference_vectors('/path/to/multi-rtl-sdr-calibration')            % captures in ./captures
%   make_reference_vectors(ref_dir, cap_dir, 'reference_vectors.json')
%
% This file belongs to the gsmcal-mi355x repository and contains no reference code: it addpath()s a checkout of
% JiaoXianjun/multi-rtl-sdr-calibration and calls its functions by name, in the order the reference's drivers do
% (gsm_sync_demod.m:107-124 per dongle; multi_rtl_sdr_gsm_FCCH_scanner.m:132-135 and :164-185 per capture).  The inputs are
% written by `python tests/golden/refvec.py export` (uint8 interleaved I,Q captures + manifest.txt).  The output is read by
% tests/test_reference_vectors_cpu.py, which compares both CPU oracles with it: with that file committed the oracle is
% pinned to the reference itself instead of "parity unpinned".
%
% Works without jsonencode / strings (MATLAB R2008b+ or Octave).  Toolboxes: fir1 (Signal Processing) and
% comm.GMSKModulator (Communications, inside gsm_SCH_training_sequence_gen) are used when present -- otherwise the exported
% taps / template are read, and the file says which (`coef_source`, `sch_template_source`).  Note that
% gsm_SCH_training_sequence_gen caches its result as gsm_SCH_training_sequence_8x.mat in the current directory.
if nargin < 2 || isempty(cap_dir), cap_dir = 'captures'; end
if nargin < 3 || isempty(out_file), out_file = 'reference_vectors.json'; end
addpath(ref_dir);

symbol_rate = (1625/6)*1e3;
ov = 8;                       % oversampling_ratio
dec = 8;                      % decimation_ratio_for_FCCH_rough_position
fs = symbol_rate*ov;
freq = 957.4e6;
nprobe = 32;

coef_source = 'fir1';
try
    coef46 = fir1(46, 200e3/fs);
    coef30 = fir1(30, 200e3/fs);
catch
    coef_source = 'exported (refvec.py)';
    coef46 = read_f64(fullfile(cap_dir, 'fir1_46.f64'), 1).';
    coef30 = read_f64(fullfile(cap_dir, 'fir1_30.f64'), 1).';
end
ts_source = 'gsm_SCH_training_sequence_gen';
try
    sch_ts = gsm_SCH_training_sequence_gen(ov);
catch
    ts_source = 'exported (refvec.py)';
    t = read_f64(fullfile(cap_dir, 'sch_training_sequence_8x.f64'), 2);
    sch_ts = complex(t(:,1), t(:,2));
end

fid = fopen(fullfile(cap_dir, 'manifest.txt'), 'r');
m = textscan(fid, '%s %s %f');
fclose(fid);
names = m{1}; kinds = m{2};

out = fopen(out_file, 'w');
fprintf(out, '{"generator": "tests/golden/make_reference_vectors.m", "interpreter": "%s", "coef_source": "%s", "sch_template_source": "%s",\n', ...
        strrep(version, '"', ''''), coef_source, ts_source);
fprintf(out, ' "coef46": '); put_vec(out, coef46); fprintf(out, ',\n "coef30": '); put_vec(out, coef30);
fprintf(out, ',\n "sch_training_sequence": '); put_cplx(out, sch_ts, 1:numel(sch_ts));
fprintf(out, ',\n "captures": {\n');
for c = 1:numel(names)
    f = fopen(fullfile(cap_dir, [names{c} '.bin']), 'r');
    s = fread(f, inf, 'uint8');               % a column of doubles holding byte values, as fread(tcp_obj, ..., 'uint8') gives
    fclose(f);
    is_sync = strcmp(kinds{c}, 'sync');
    if is_sync, coef = coef46; else, coef = coef30; end
    fprintf(out, '  "%s": {', names{c});
    % ---- front end: raw2iq, then the drivers' inline channel filter ----
    r = raw2iq(s);
    n = numel(r);
    fprintf(out, '"raw2iq": {"n": %d, "sum_re": %s, "sum_im": %s, "sum_abs2": %s, "first16": ', n, num(sum(real(r))), num(sum(imag(r))), num(sum(real(r).^2 + imag(r).^2)));
    put_cplx(out, r, 1:16); fprintf(out, ', "last16": '); put_cplx(out, r, n-15:n); fprintf(out, '},\n');
    rf = filter(coef, 1, r);
    fprintf(out, '   "filter": '); put_cplx(out, rf, probes(n, nprobe));
    % ---- coarse detector on every 64th filtered sample ----
    [cpos, csnr] = FCCH_coarse_position(rf(1:ov*dec:end, 1), dec);
    fprintf(out, ',\n   "coarse_pos": '); put_vec(out, cpos); fprintf(out, ', "coarse_snr": '); put_vec(out, csnr);
    if ~is_sync
        % acceptance rule of the scanner (its lines 168-185): at least three hits, every spacing within 50 of 12500 or of 13750
        snr = 0; num_hit = 0;
        if numel(cpos) >= 3
            d = diff(cpos);
            off = abs(d - 12500) > 50;
            if ~any(off) || ~any(abs(d(off) - (12500 + 1250)) > 50)
                snr = mean(csnr); num_hit = numel(cpos);
            end
        end
        fprintf(out, ', "snr": %s, "num_hit": %s}', num(snr), num(num_hit));
    else
        % ---- the per-dongle body of the sync driver ----
        sampling_ppm = zeros(1, 2); carrier_ppm = zeros(1, 2);
        [fpos, r1, sampling_ppm(1), carrier_ppm(1)] = FCCH_fine_correction(rf(:, 1), cpos, ov, freq);
        fprintf(out, ',\n   "fcch_pos": '); put_vec(out, fpos);
        fprintf(out, ', "sampling_ppm1": %s, "carrier_ppm1": %s, "r1_len": %d', num(sampling_ppm(1)), num(carrier_ppm(1)), len_or_m1(r1));
        if numel(r1) > 1, fprintf(out, ', "r1_probe": '); put_cplx(out, r1, probes(numel(r1), nprobe)); end
        [pos_info, r2, sampling_ppm(2)] = SCH_corr_rate_correction(r1, fpos, sch_ts, ov);
        fprintf(out, ',\n   "pos_info_rows": %d, "pos_info": ', size(pos_info, 1)); put_vec(out, pos_info(:));
        fprintf(out, ', "sampling_ppm2": %s, "r2_len": %d', num(sampling_ppm(2)), len_or_m1(r2));
        [r3, carrier_ppm(2)] = carrier_correct_post_SCH(r2, pos_info, ov, freq);
        fprintf(out, ',\n   "carrier_ppm2": %s, "r3_len": %d', num(carrier_ppm(2)), len_or_m1(r3));
        if numel(r3) > 1, fprintf(out, ', "r3_probe": '); put_cplx(out, r3, probes(numel(r3), nprobe)); end
        fprintf(out, ',\n   "total_sampling_ppm": %s, "total_carrier_ppm": %s}', num(total_ppm_calculation(sampling_ppm)), num(total_ppm_calculation(carrier_ppm)));
    end
    if c < numel(names), fprintf(out, ',\n'); else, fprintf(out, '\n'); end
end
fprintf(out, ' }\n}\n');
fclose(out);
disp(['wrote ' out_file]);
end

function idx = probes(n, count)
% 1-based probe positions spread over 1..n: floor(k*(n-1)/(count-1)) + 1, k = 0..count-1 (tests/golden/refvec.py probe_indices)
idx = floor(((0:count-1) .* (n - 1)) ./ (count - 1)) + 1;
end

function v = read_f64(path, ncol)
f = fopen(path, 'r', 'ieee-le');
v = fread(f, inf, 'float64');
fclose(f);
v = reshape(v, ncol, []).';
end

function n = len_or_m1(r)
if numel(r) > 1, n = numel(r); else, n = -1; end
end

function s = num(x)
if isnan(x), s = '"nan"';
elseif isinf(x) && x > 0, s = '"inf"';
elseif isinf(x), s = '"-inf"';
else, s = sprintf('%.17g', x);
end
end

function put_vec(out, v)
v = v(:).';
fprintf(out, '[');
for i = 1:numel(v)
    if i > 1, fprintf(out, ', '); end
    fprintf(out, '%s', num(v(i)));
end
fprintf(out, ']');
end

function put_cplx(out, v, idx)
fprintf(out, '{"idx": '); put_vec(out, idx);
fprintf(out, ', "re": '); put_vec(out, real(v(idx)));
fprintf(out, ', "im": '); put_vec(out, imag(v(idx)));
fprintf(out, '}');
end
