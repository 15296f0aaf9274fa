import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import vfidkr_amd
from vfidkr_amd import cabi, synthetic as S
dev = torch.device("cuda:0")
h, w = S.padded_size(1080, 1920)
gen = S.generator()
filt = S.filters(1, h, w, gen).to(dev)
flow = S.flow(1, h, w, 8.0, gen, "smooth").to(dev)
full = S.context(1, 196, h, w, gen).to(dev)
def timed(fn, n=20):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
timed(lambda: cabi.filterinterp_forward_ori(full, flow, filt, torch.empty_like(full)))
for groups in (1, 2):
    cabi.lib().vfi_dev_filterinterp(0, groups)
    for C in (12, 24, 49, 98, 196):
        x = full[:, :C].contiguous(); o = torch.empty_like(x)
        ms = timed(lambda: cabi.filterinterp_forward_ori(x, flow, filt, o))
        print("groups %d C=%3d  %8.4f ms  per channel %7.2f us" % (groups, C, ms, ms * 1e3 / C), flush=True)
