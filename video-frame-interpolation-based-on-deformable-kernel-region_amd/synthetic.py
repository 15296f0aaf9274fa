"""Seeded synthetic inputs of the hot path.

Everything is generated on the host with `torch.Generator('cpu').manual_seed(seed)`
and returned as CPU float32 tensors; callers move them to the GPU.

Flow models (what `FlowProjection` / `FilterInterpolation` receive in DAIN is
`Upsample(x4, bilinear)(20 * t * PWCNet(...))`, networks/DAIN.py:306-308):
  "quarter"  SURVEY 8(d) to the letter: N(0, sigma^2) at [B,2,H/4,W/4], bilinear x4.
             Neighbouring quarter-res samples are independent, so the field changes
             by ~sigma*sqrt(2)/4 px per pixel -- far rougher than any estimated flow.
  "smooth"   N(0, sigma^2) at 1/64 resolution, bicubic to quarter resolution, then the
             same bilinear x4: piecewise-smooth motion with the same sigma, which is
             what 8(d) describes in words ("smooth, sub-pixel-varying").
  "uniform1" U(-1, 1) per pixel (my_package/test_module.py:1018).
  "wild"     U(-W/2, W/2) per pixel: adversarial, every tap uncoalesced.
"""
import torch
import torch.nn.functional as F

SEED = 1234
FLOW_SIGMA = {"vimeo": 2.0, "480p": 4.0, "1080p": 8.0, "4k": 16.0}


def generator(seed=SEED):
    return torch.Generator(device="cpu").manual_seed(seed)


def padded_size(h, w):
    """demo_MiddleBury.py:294-310: next multiple of 128, or +64 if already a multiple."""
    def one(v):
        return v + 64 if v % 128 == 0 else (v // 128 + 1) * 128
    return one(h), one(w)


def frames(b, h, w, gen, c=3):
    return torch.rand((b, c, h, w), generator=gen, dtype=torch.float32)


def context(b, c, h, w, gen):
    return torch.randn((b, c, h, w), generator=gen, dtype=torch.float32)


def filters(b, h, w, gen, fs=4, normalised=False):
    f = torch.rand((b, fs * fs, h, w), generator=gen, dtype=torch.float32)
    if normalised:
        f = torch.softmax(f, dim=1)
    return f.contiguous()


def depth_weight(b, h, w, gen):
    """U(0.1, 1.0): 'must be larger than zero' (my_package/test_module.py:1019)."""
    return (torch.rand((b, 1, h, w), generator=gen, dtype=torch.float32) * 0.9 + 0.1).contiguous()


def flow(b, h, w, sigma, gen, model="smooth"):
    if model == "uniform1":
        return (torch.rand((b, 2, h, w), generator=gen, dtype=torch.float32) * 2.0 - 1.0).contiguous()
    if model == "wild":
        return ((torch.rand((b, 2, h, w), generator=gen, dtype=torch.float32) - 0.5) * float(w)).contiguous()
    qh, qw = (h + 3) // 4, (w + 3) // 4
    if model == "quarter":
        q = torch.randn((b, 2, qh, qw), generator=gen, dtype=torch.float32) * sigma
    elif model == "smooth":
        # control points every 1/31 of the frame width (64 px at the padded 1080p width of 1984), so that the field
        # has the same shape at every resolution: sigma and the control spacing scale together, the flow gradient
        # (what sizes the staged windows) does not
        cell = max(1.0, w / 31.0)
        ch, cw = max(2, int(round(h / cell)) + 1), max(2, int(round(w / cell)) + 1)
        coarse = torch.randn((b, 2, ch, cw), generator=gen, dtype=torch.float32) * sigma
        q = F.interpolate(coarse, size=(qh, qw), mode="bicubic", align_corners=True)
    else:
        raise ValueError("unknown flow model %r" % (model,))
    full = F.interpolate(q, scale_factor=4, mode="bilinear", align_corners=False)
    return full[:, :, :h, :w].contiguous()


# feature pyramid PWC-Net feeds to the correlation layer (PWCNet/PWCNet.py:221-300):
# (channels, downscale) from the coarsest level to the finest
PWC_LEVELS = ((196, 64), (128, 32), (96, 16), (64, 8), (32, 4))


def correlation_features(b, h, w, gen):
    """List of (f1, f2) pairs ~ N(0,1), one per pyramid level of a padded HxW frame."""
    out = []
    for c, s in PWC_LEVELS:
        shape = (b, c, h // s, w // s)
        out.append((torch.randn(shape, generator=gen, dtype=torch.float32),
                    torch.randn(shape, generator=gen, dtype=torch.float32)))
    return out
