import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gsmcal
g = gsmcal
coef = g.synth.fir1(46, 200e3 / g.synth.FS); ts = g.synth.sch_training_sequence()
raw = np.stack([g.synth.make_stream(dongle=d, num_frames=102)[0] for d in range(40, 48)])
for k, v in [a.split("=") for a in sys.argv[1:]]:
    os.environ[k] = v
ctx = g.Context(0)
print("calling", flush=True)
out = g.calibrate_batch(raw, coef, ts, 957.4e6, ctx=ctx)
print("ok", out["table"][:, 9], flush=True)
