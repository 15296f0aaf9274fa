"""Where does the gain of two concurrent C=196 launches come from?  Six launches: in a row on one stream; three + three on
two streams (different tensors / the same input tensors); as two batches of three in one launch each."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vfidkr_amd import cabi, synthetic as S
dev = torch.device("cuda:0"); gen = S.generator()
h, w = S.padded_size(1080, 1920)
ctx = [S.context(1, 196, h, w, gen).to(dev) for _ in range(2)]
filt = [S.filters(1, h, w, gen).to(dev) for _ in range(2)]
flow = [[S.flow(1, h, w, 8.0 * t, gen, "smooth").to(dev) for t in (0.25, 0.5, 0.75)] for _ in range(2)]
out = [torch.empty_like(ctx[0]) for _ in range(2)]
side = torch.cuda.Stream(dev)

def wall(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def serial():
    for t in range(3):
        for d in range(2): cabi.filterinterp_forward_ori(ctx[d], flow[d][t], filt[d], out[d])
def two(same):
    main = torch.cuda.current_stream(dev); side.wait_stream(main)
    for d in range(2):
        with torch.cuda.stream(main if d == 0 else side):
            s = 0 if same else d
            for t in range(3): cabi.filterinterp_forward_ori(ctx[s], flow[s][t], filt[s], out[d])
    main.wait_stream(side)
print("six launches in a row            %.3f ms" % wall(serial), flush=True)
print("3 + 3 on two streams             %.3f ms" % wall(lambda: two(False)), flush=True)
print("3 + 3, both on direction 0's data %.3f ms" % wall(lambda: two(True)), flush=True)
print("six launches in a row            %.3f ms" % wall(serial), flush=True)
