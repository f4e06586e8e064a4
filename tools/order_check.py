import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
if mode == "torch_first":
    import torch
import gsmcal
ctx = gsmcal.default_context(0)
print("ctx ok", flush=True)
import torch
try:
    x = torch.zeros(4).cuda()
    print(mode, "torch cuda ok", x.device, flush=True)
except Exception as e:
    print(mode, "torch cuda FAILED:", e, flush=True)
for l in open("/proc/self/maps"):
    if "libamdhip64" in l and " r-xp " in l:
        print(l.split()[-1])
