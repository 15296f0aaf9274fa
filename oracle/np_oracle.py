"""Second, independent formulation of the hot-path ops in vectorised numpy.

TEST INFRASTRUCTURE ONLY.  Purpose: catch transcription errors in
oracle/vfi_oracle.c.  Where the arithmetic allows it the float32 operations are
issued in the reference's order, so the result is bit-identical to the C
oracle in strict mode (fmad=0); scatter ops are formulated with np.add.at in
float64 and compared with a tolerance (or bit-exactly on dyadic inputs whose
sums are exact in any order).

Semantics follow SURVEY.md section 9 (reference lines cited there and in
vfi_oracle.h); nothing here is derived from the C file.
"""
import numpy as np

f32 = np.float32


def _grid(B, H, W):
    xs = np.arange(W, dtype=f32)[None, None, :]
    ys = np.arange(H, dtype=f32)[None, :, None]
    return np.broadcast_to(xs, (B, H, W)), np.broadcast_to(ys, (B, H, W))


def _gather(img, cj, ci):
    """img [B,C,H,W]; cj,ci int [B,H,W] -> img[b,:,cj,ci] as [B,C,H,W]."""
    B, C, H, W = img.shape
    idx = (cj.astype(np.int64) * W + ci.astype(np.int64)).reshape(B, 1, H * W)
    idx = np.broadcast_to(idx, (B, C, H * W))
    return np.take_along_axis(img.reshape(B, C, H * W), idx, axis=2).reshape(B, C, H, W)


def _fi_geometry(flow, H, W, fs):
    B = flow.shape[0]
    xs, ys = _grid(B, H, W)
    fx, fy = flow[:, 0].astype(f32), flow[:, 1].astype(f32)
    x2 = xs + fx
    y2 = ys + fy
    valid = ((x2 >= 0) & (y2 >= 0) & (x2 <= f32(W - 1)) & (y2 <= f32(H - 1)) &
             (np.abs(fx) < f32(W) / f32(2)) & (np.abs(fy) < f32(H) / f32(2)))
    x2s = np.where(valid, x2, f32(0))
    y2s = np.where(valid, y2, f32(0))
    ix = np.trunc(x2s).astype(np.int32)
    iy = np.trunc(y2s).astype(np.int32)
    alpha = x2s - ix.astype(f32)
    beta = y2s - iy.astype(f32)
    L = ix + 1 - fs // 2
    T = iy + 1 - fs // 2
    return valid, x2s, y2s, ix, iy, alpha, beta, L, T


def _blend(alpha, beta, TL, TR, BL, BR):
    a = alpha[:, None]
    b = beta[:, None]
    one = f32(1)
    t = ((one - a) * (one - b)) * TL
    t = t + (a * (one - b)) * TR
    t = t + ((one - a) * b) * BL
    t = t + (a * b) * BR
    return t.astype(f32)


def filterinterp_ori_fwd(img, flow, filt):
    img, flow, filt = img.astype(f32), flow.astype(f32), filt.astype(f32)
    B, C, H, W = img.shape
    fs = int(np.sqrt(f32(filt.shape[1])))
    valid, x2, y2, ix, iy, alpha, beta, L, T = _fi_geometry(flow, H, W, fs)
    q = [np.zeros((B, C, H, W), f32) for _ in range(4)]
    for dj in range(fs):
        for di in range(fs):
            j = T + dj
            i = L + di
            quad = (j > iy).astype(np.int32) * 2 + (i > ix).astype(np.int32)
            val = _gather(img, np.clip(j, 0, H - 1), np.clip(i, 0, W - 1))
            prod = val * filt[:, dj * fs + di][:, None]
            for k in range(4):
                q[k] = np.where((quad == k)[:, None], q[k] + prod, q[k])
    out = _blend(alpha, beta, *q)
    return np.where(valid[:, None], out, img).astype(f32)


def _defor_tap(img, fracY, fracX):
    B, C, H, W = img.shape
    top = np.trunc(fracY).astype(np.int32)
    left = np.trunc(fracX).astype(np.int32)
    phiY = (fracY - top.astype(f32))[:, None]
    phiX = (fracX - left.astype(f32))[:, None]
    one = f32(1)
    t, bo = np.clip(top, 0, H - 1), np.clip(top + 1, 0, H - 1)
    l, r = np.clip(left, 0, W - 1), np.clip(left + 1, 0, W - 1)
    s = ((one - phiX) * (one - phiY)) * _gather(img, t, l)
    s = s + (phiX * (one - phiY)) * _gather(img, t, r)
    s = s + ((one - phiX) * phiY) * _gather(img, bo, l)
    s = s + (phiY * phiX) * _gather(img, bo, r)
    return s.astype(f32)


def filterinterp_defor_fwd(variant, img, flow, filt, off):
    img, flow, off = img.astype(f32), flow.astype(f32), off.astype(f32)
    B, C, H, W = img.shape
    if variant == 2:
        fs = int(np.sqrt(f32(off.shape[1] // 2)))
    else:
        filt = filt.astype(f32)
        fs = int(np.sqrt(f32(filt.shape[1])))
    if variant == 0 and fs not in (4, 6):
        return np.zeros_like(img)
    fs2 = fs * fs
    valid, x2, y2, ix, iy, alpha, beta, L, T = _fi_geometry(flow, H, W, fs)
    q = [np.zeros((B, C, H, W), f32) for _ in range(4)]
    for dj in range(fs):
        for di in range(fs):
            k = dj * fs + di
            j = T + dj
            i = L + di
            fracY = np.clip(j, 0, H - 1).astype(f32) + off[:, k]
            fracX = np.clip(i, 0, W - 1).astype(f32) + off[:, fs2 + k]
            v = _defor_tap(img, fracY, fracX)
            if variant != 2:
                v = v * filt[:, k][:, None]
            if variant == 0:
                masks = [((j <= iy) & (i <= ix)), ((j <= iy) & (i > ix)),
                         ((j > iy) & (i <= ix)), ((j > iy) & (i > ix))]
            else:
                masks = [((fracX <= x2) & (fracY <= y2)), ((fracX > x2) & (fracY <= y2)),
                         ((fracX <= x2) & (fracY > y2)), ((fracX > x2) & (fracY > y2))]
            for n in range(4):
                q[n] = np.where(masks[n][:, None], q[n] + v, q[n])
    out = _blend(alpha, beta, *q)
    return np.where(valid[:, None], out, img).astype(f32)


def filterinterp_defor_bwd(variant, img, flow, filt, off, gout):
    """float64 formulation of the deformable backwards; returns (gimg, gflow, gfilt or None, goff)."""
    img, flow, off, gout = (a.astype(np.float64) for a in (img, flow, off, gout))
    B, C, H, W = img.shape
    if variant == 2:
        fs = int(np.sqrt(f32(off.shape[1] // 2)))
        filt = None
    else:
        filt = filt.astype(np.float64)
        fs = int(np.sqrt(f32(filt.shape[1])))
    fs2 = fs * fs
    valid, x2, y2, ix, iy, alpha, beta, L, T = _fi_geometry(flow.astype(f32), H, W, fs)
    x2, y2, alpha, beta = (a.astype(np.float64) for a in (x2, y2, alpha, beta))
    kq = [(1 - alpha) * (1 - beta), alpha * (1 - beta), (1 - alpha) * beta, alpha * beta]
    gimg = np.zeros((B, C, H * W))
    gfilt = None if filt is None else np.zeros_like(filt)
    goff = np.zeros_like(off)
    q = [np.zeros((B, C, H, W)) for _ in range(4)]
    bidx = np.arange(B)[:, None, None]
    for dj in range(fs):
        for di in range(fs):
            k = dj * fs + di
            j, i = T + dj, L + di
            cj, ci = np.clip(j, 0, H - 1), np.clip(i, 0, W - 1)
            fracY = (cj.astype(f32) + off[:, k].astype(f32)).astype(np.float64)
            fracX = (ci.astype(f32) + off[:, fs2 + k].astype(f32)).astype(np.float64)
            if variant == 0:
                quad = (j > iy) * 2 + (i > ix)
            else:
                quad = np.where(fracY <= y2, 0, 2) + np.where(fracX <= x2, 0, 1)
            top, left = np.trunc(fracY).astype(np.int64), np.trunc(fracX).astype(np.int64)
            phiY, phiX = (fracY - top)[:, None], (fracX - left)[:, None]
            t, bo = np.clip(top, 0, H - 1), np.clip(top + 1, 0, H - 1)
            l, r = np.clip(left, 0, W - 1), np.clip(left + 1, 0, W - 1)
            vTL, vTR, vBL, vBR = _gather(img, t, l), _gather(img, t, r), _gather(img, bo, l), _gather(img, bo, r)
            v = (1 - phiX) * (1 - phiY) * vTL + phiX * (1 - phiY) * vTR + (1 - phiX) * phiY * vBL + phiX * phiY * vBR
            dY = -(1 - phiX) * vTL + (1 - phiX) * vBL - phiX * vTR + phiX * vBR
            dX = -(1 - phiY) * vTL + (1 - phiY) * vTR - phiY * vBL + phiY * vBR
            wq = np.choose(quad, kq) * valid                         # [B,H,W]
            wgt = 1.0 if filt is None else filt[:, k]
            gw = gout * wq[:, None]                                   # [B,C,H,W]
            idx = (cj * W + ci).reshape(B, 1, H * W)
            for b in range(B):
                for c in range(C):
                    np.add.at(gimg[b, c], idx[b, 0], (gw[b, c] * (wgt if filt is None else wgt[b])).reshape(-1))
            if filt is not None:
                gfilt[:, k] = (gw * v).sum(1)
            goff[:, k] = (gw * dY).sum(1) * wgt
            goff[:, fs2 + k] = (gw * dX).sum(1) * wgt
            for n in range(4):
                q[n] += np.where((quad == n)[:, None], v * (wgt if filt is None else wgt[:, None]), 0.0)
    a, b_ = alpha[:, None], beta[:, None]
    gx = (gout * ((1 - b_) * (q[1] - q[0]) + b_ * (q[3] - q[2]))).sum(1) * valid
    gy = (gout * ((1 - a) * (q[2] - q[0]) + a * (q[3] - q[1]))).sum(1) * valid
    gflow = np.stack([gx, gy], 1)
    return gimg.reshape(B, C, H, W), gflow, gfilt, goff


def _project_targets(flow, H, W):
    B = flow.shape[0]
    xs, ys = _grid(B, H, W)
    fx, fy = flow[:, 0].astype(f32), flow[:, 1].astype(f32)
    x2 = xs + fx
    y2 = ys + fy
    valid = (x2 >= 0) & (y2 >= 0) & (x2 <= f32(W - 1)) & (y2 <= f32(H - 1))
    L = np.trunc(np.where(valid, x2, 0)).astype(np.int64)
    T = np.trunc(np.where(valid, y2, 0)).astype(np.int64)
    R = np.minimum(L + 1, W - 1)
    Bm = np.minimum(T + 1, H - 1)
    return valid, fx, fy, L, T, R, Bm


def _fillhole(count, out):
    """Nearest non-hole in -x, +x, -y, +y; mean of those found (python loops: small inputs only)."""
    B, _, H, W = out.shape
    res = out.copy()
    for b in range(B):
        cnt = count[b, 0]
        for y in range(H):
            for x in range(W):
                if cnt[y, x] > 0:
                    continue
                found = []
                for dy, dx in ((0, -1), (0, 1), (-1, 0), (1, 0)):
                    yy, xx = y + dy, x + dx
                    while 0 <= yy < H and 0 <= xx < W and cnt[yy, xx] == 0:
                        yy += dy
                        xx += dx
                    if 0 <= yy < H and 0 <= xx < W and cnt[yy, xx] > 0:
                        found.append((yy, xx))
                if found:
                    for ch in range(2):
                        s = f32(0)
                        for (yy, xx) in found:
                            s = f32(s + out[b, ch, yy, xx])
                        res[b, ch, y, x] = f32(s / f32(len(found)))
    return res


def flowproj_fwd(flow, fillhole=1, depth=None):
    """Returns (out, count); accumulation in float64, rounded to float32 once."""
    flow = flow.astype(f32)
    B, _, H, W = flow.shape
    valid, fx, fy, L, T, R, Bm = _project_targets(flow, H, W)
    d = np.ones((B, H, W), f32) if depth is None else depth[:, 0].astype(f32)
    acc = np.zeros((B, 3, H * W), np.float64)
    ax = (-d * fx).astype(f32) if depth is not None else -fx
    ay = (-d * fy).astype(f32) if depth is not None else -fy
    for b in range(B):
        m = valid[b]
        for (ty, tx) in ((T, L), (T, R), (Bm, L), (Bm, R)):
            idx = (ty[b][m] * W + tx[b][m])
            np.add.at(acc[b, 0], idx, ax[b][m].astype(np.float64))
            np.add.at(acc[b, 1], idx, ay[b][m].astype(np.float64))
            np.add.at(acc[b, 2], idx, d[b][m].astype(np.float64))
    acc = acc.reshape(B, 3, H, W)
    count = acc[:, 2:3].astype(f32)
    s = acc[:, 0:2].astype(f32)
    out = np.where(count > 0, s / np.where(count > 0, count, f32(1)), s).astype(f32)
    if fillhole:
        out = _fillhole(count, out)
    return out, count


def mindepthflowproj_fwd(flow, weight, fillhole=1):
    """Independent formulation of the defined MinDepthFlowProjection result: per target, the source with the
    largest positive weight, lowest raster index on ties (np.lexsort instead of a sequential sweep).
    Returns (out, count)."""
    flow = flow.astype(f32)
    B, _, H, W = flow.shape
    valid, fx, fy, L, T, _, _ = _project_targets(flow, H, W)
    wgt = weight[:, 0].astype(f32)
    out = np.zeros((B, 2, H * W), f32)
    count = np.zeros((B, 1, H * W), f32)
    for b in range(B):
        m = valid[b] & (wgt[b] > 0)
        src = np.flatnonzero(m.ravel())
        tgt = (T[b] * W + L[b]).ravel()[src]
        wv = wgt[b].ravel()[src]
        order = np.lexsort((src, -wv.astype(np.float64), tgt))      # by target, then weight descending, then index
        tgt_s, src_s = tgt[order], src[order]
        first = np.ones(len(order), bool)
        first[1:] = tgt_s[1:] != tgt_s[:-1]
        win_t, win_s = tgt_s[first], src_s[first]
        out[b, 0, win_t] = -fx[b].ravel()[win_s]
        out[b, 1, win_t] = -fy[b].ravel()[win_s]
        count[b, 0, win_t] = wgt[b].ravel()[win_s]
    out = out.reshape(B, 2, H, W)
    count = count.reshape(B, 1, H, W)
    if fillhole:
        out = _fillhole(count, out)
    return out, count


def interp_fwd(img, flow):
    img, flow = img.astype(f32), flow.astype(f32)
    B, C, H, W = img.shape
    xs, ys = _grid(B, H, W)
    x2 = xs + flow[:, 0]
    y2 = ys + flow[:, 1]
    valid = (x2 >= 0) & (y2 >= 0) & (x2 < f32(W)) & (y2 < f32(H))
    x2 = np.where(valid, x2, f32(0))
    y2 = np.where(valid, y2, f32(0))
    L = np.trunc(x2).astype(np.int32)
    T = np.trunc(y2).astype(np.int32)
    R = np.minimum(L + 1, W - 1)
    Bm = np.minimum(T + 1, H - 1)
    alpha = x2 - L.astype(f32)
    beta = y2 - T.astype(f32)
    out = _blend(alpha, beta, _gather(img, T, L), _gather(img, T, R), _gather(img, Bm, L), _gather(img, Bm, R))
    return np.where(valid[:, None], out, f32(0)).astype(f32)


def sepconv_fwd(img, v, h):
    img, v, h = img.astype(f32), v.astype(f32), h.astype(f32)
    B, C, H, W = img.shape
    fs = v.shape[1]
    oH, oW = H - fs + 1, W - fs + 1
    out = np.zeros((B, C, oH, oW), f32)
    for fy in range(fs):
        for fx in range(fs):
            t1 = img[:, :, fy:fy + oH, fx:fx + oW]
            out = out + (t1 * v[:, fy][:, None]) * h[:, fx][:, None]
    return out.astype(f32)


def sepconvflow_fwd(v, h):
    v, h = v.astype(f32), h.astype(f32)
    B, fs = v.shape[:2]
    out = np.zeros((B, 2) + v.shape[2:], f32)
    centre = (np.float64(f32(fs)) - 1.0) / 2.0
    for ch, k in ((1, v), (0, h)):
        m = np.zeros(k.shape[:1] + k.shape[2:], f32)
        s = np.zeros_like(m)
        for f in range(fs):
            m = m + f32(f) * k[:, f]
            s = s + k[:, f]
        with np.errstate(divide="ignore", invalid="ignore"):
            val = ((m / s).astype(np.float64) - centre).astype(f32)
        out[:, ch] = np.where(np.abs(s) > 0, val, f32(-2000))
    return out


def correlation_fwd(f1, f2, pad=4, k=1, md=4, s1=1, s2=1, dtype=np.float32):
    """Sequential channel order in `dtype` (float32: bit-identical to C order=1 strict)."""
    f1, f2 = f1.astype(dtype), f2.astype(dtype)
    B, C, H, W = f1.shape
    kr = (k - 1) // 2
    border = kr + md
    pH, pW = H + 2 * pad, W + 2 * pad
    oH = int(np.ceil(f32(pH - 2 * border) / f32(s1)))
    oW = int(np.ceil(f32(pW - 2 * border) / f32(s1)))
    dr = md // s2
    dsz = 2 * dr + 1
    p1 = np.zeros((B, C, pH, pW), dtype)
    p2 = np.zeros((B, C, pH, pW), dtype)
    p1[:, :, pad:pad + H, pad:pad + W] = f1
    p2[:, :, pad:pad + H, pad:pad + W] = f2
    out = np.zeros((B, dsz * dsz, oH, oW), dtype)
    ys = np.arange(oH) * s1 + md
    xs = np.arange(oW) * s1 + md
    for tj in range(-dr, dr + 1):
        for ti in range(-dr, dr + 1):
            acc = np.zeros((B, oH, oW), dtype)
            for j in range(-kr, kr + 1):
                for i in range(-kr, kr + 1):
                    a = p1[:, :, (ys + j)[:, None], (xs + i)[None, :]]
                    b = p2[:, :, (ys + tj * s2 + j)[:, None], (xs + ti * s2 + i)[None, :]]
                    for c in range(C):
                        acc = acc + a[:, c] * b[:, c]
            out[:, (tj + dr) * dsz + (ti + dr)] = acc / dtype(k * k * C)
    return out
