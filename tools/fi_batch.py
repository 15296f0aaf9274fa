import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vfidkr_amd import cabi, synthetic as S
dev = torch.device("cuda:0"); gen = S.generator()
h, w = S.padded_size(1080, 1920)
def timed(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for B in (1, 2, 3):
    ctx = S.context(B, 196, h, w, gen).to(dev); filt = S.filters(B, h, w, gen).to(dev)
    flow = S.flow(B, h, w, 8.0, gen, "smooth").to(dev); out = torch.empty_like(ctx)
    ms = timed(lambda: cabi.filterinterp_forward_ori(ctx, flow, filt, out))
    print("B=%d  %.4f ms  %.4f ms per image  frac %.3f" % (B, ms, ms / B, 1640.0 * h * w * B / ms / 1e6 / 8000.0), flush=True)
    del ctx, filt, flow, out
