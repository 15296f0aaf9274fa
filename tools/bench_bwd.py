"""Times the backward entry points that matter for training (FilterInterpolation, Interpolation) at padded 1080p, C = 3,
on a smooth flow; run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vfidkr_amd  # noqa: E402
if "--lib" in sys.argv:         # a development build of the library (tools/mkvariant.sh)
    vfidkr_amd.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from vfidkr_amd import cabi  # noqa: E402

H, W = 1152, 1984
gen = torch.Generator().manual_seed(5)
img = torch.rand((1, 3, H, W), generator=gen).cuda()
yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
flow = torch.stack([6.0 * torch.sin(xx / 97.0 + yy / 131.0) + 1.3, 5.0 * torch.cos(xx / 113.0 - yy / 71.0) - 0.7])[None].cuda()
filt = torch.rand((1, 16, H, W), generator=gen).cuda()
gout = torch.randn((1, 3, H, W), generator=gen).cuda()


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


g1, g2, g3 = torch.zeros_like(img), torch.zeros_like(flow), torch.zeros_like(filt)
print("filterinterp bwd ori C=3   %8.4f ms" % timed(lambda: cabi.filterinterp_backward_ori(img, flow, filt, gout, g1, g2, g3)), flush=True)
gi, gf = torch.zeros_like(img), torch.zeros_like(flow)
print("interpolation bwd C=3      %8.4f ms" % timed(lambda: cabi.interpolation_backward(img, flow, gout, gi, gf)), flush=True)
# deformable kernel-region variants (fs = 4): offsets of a pixel or so
off = (torch.rand((1, 32, H, W), generator=gen) * 2.0 - 1.0).cuda()
go = torch.zeros_like(off)
for variant, name in ((0, "offset"), (1, "region (deforconv)")):
    print("defor bwd %-20s %8.4f ms" % (name, timed(lambda: cabi.filterinterp_backward_defor(variant, img, flow, filt, off, gout, g1, g2, g3, go), 5)), flush=True)
print("defor fwd region C=3           %8.4f ms" % timed(lambda: cabi.filterinterp_forward_defor(1, img, flow, filt, off, g1)), flush=True)
