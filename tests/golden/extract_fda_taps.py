"""Extract the channel-filter numerators from the reference's FDATool sessions.

The reference's chn_filter_8x_4x.m:9-10 loads `Num` from gsm_chn_filter_8x.mat, which is not in
the reference repo; the FDATool session gsm_chn_filter_8x.fda (a MAT-v5 file) is, and holds the
same design as a dfilt.dffir object whose `refnum` is the float64 numerator.  This script was run
once in the build container (where /root/reference exists) to write the two *_num.txt fixtures.
"""
import sys
import scipy.io

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
for nm in ("8x", "4x"):
    d = scipy.io.loadmat(f"{REF}/gsm_chn_filter_{nm}.fda", struct_as_record=False, squeeze_me=True)
    num = d["s"].current_filt[0][3][3].refnum
    with open(f"gsm_chn_filter_{nm}_num.txt", "w") as f:
        f.write(f"# Numerator 'Num' of the FDATool session gsm_chn_filter_{nm}.fda (dfilt.dffir refnum), {len(num)} taps, float64 %.17g\n")
        f.write("# data extracted by tests/golden/extract_fda_taps.py; consumed by chn_filter_8x_4x (reference chn_filter_8x_4x.m:9-10 loads the same values from a .mat)\n")
        for v in num:
            f.write("%.17g\n" % v)
