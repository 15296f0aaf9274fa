#!/usr/bin/env python3
"""bench.py -- throughput of the frame-synthesis hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json metric "interpolated frames/sec at 1080p", configs[2]): one STEP is
the hot path of DAIN_slowmotion x4 for ONE 1080p frame pair (padded 1152x1984,
demo_MiddleBury.py:294-310), i.e. exactly the native calls networks/DAIN_slowmotion.py:147-183
and PWCNet/PWCNet.py:230-300 make for it:

    10 x correlation forward   (5 pyramid levels x 2 directions; pad=4,k=1,md=4,s1=s2=1)
     6 x DepthFlowProjection   (2 directions x t in {0.25,0.5,0.75}; fillhole=1)
     6 x FilterInterpolation   on the 196-channel context tensor
     6 x FilterInterpolation   on the 3-channel frame

and yields 3 interpolated frames.  The convolutional sub-networks around it (PWC-Net convs,
MegaDepth, context/rectify nets: stock MIOpen work) are outside this repo's scope and are
NOT part of the step -- `value` is hot-path frames/s, not end-to-end model frames/s.
All inputs are synthetic (vfidkr_amd/synthetic.py), float32, resident in HBM before timing.
With N ranks every rank processes its own pair per step (weak scaling, no collective).

Every native call goes through the C ABI of libvfi_hip.so (ctypes, vfidkr_amd/cabi.py).
The JSON line also carries:
  roofline      dominant kernel (FilterInterpolation, C=196): algorithmic bytes per launch
                (1640 B/pixel x 2,285,568 pixels) / mean launch time measured with
                HIP events on the launch stream inside the timed region, vs 8 TB/s
  gate          north-star gate: 2 x FilterInterpolation(C=3) + 2 x FlowProjection at 1080p
                (530 MB algorithmic) timed the same way
  cpu_baseline  the CPU oracle (oracle/, a port: kind "port") timed on the host cores on a
                bounded sample of the same workload, rank 0 at N=1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
TIMES = (0.25, 0.5, 0.75)       # x4 slow motion: numFrames = 3 (DAIN_slowmotion.py:29)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--flow-model", default="smooth", choices=["smooth", "quarter", "uniform1"],
                    help="synthetic flow field (vfidkr_amd/synthetic.py); 'quarter' = SURVEY 8d to the letter")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(16, cores)")
    ap.add_argument("--direct", action="store_true", help="force the direct-gather FilterInterpolation kernel")
    return ap.parse_args()


class Workload:
    """All tensors of one frame pair, resident on `dev`."""

    def __init__(self, torch, S, dev, h, w, flow_model, seed):
        gen = S.generator(seed)
        self.h, self.w = h, w
        self.px = h * w
        sigma = 8.0 * (w / 1984.0)
        self.frames = [S.frames(1, h, w, gen).to(dev) for _ in range(2)]
        self.ctx = [S.context(1, 196, h, w, gen).to(dev) for _ in range(2)]
        self.filters = [S.filters(1, h, w, gen).to(dev) for _ in range(2)]
        self.depth = [S.depth_weight(1, h, w, gen).to(dev) for _ in range(2)]
        base = [S.flow(1, h, w, sigma, gen, flow_model) for _ in range(2)]
        # forward_flownets: one flow per time offset, scaled by t (DAIN_slowmotion.py:214-215)
        self.flows = [[(base[d] * (2.0 * t)).contiguous().to(dev) for t in TIMES] for d in range(2)]
        self.corr = [[(a.to(dev), b.to(dev)) for a, b in S.correlation_features(1, h, w, gen)] for _ in range(2)]
        e = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)      # noqa: E731
        self.count = e(1, 1, h, w)
        self.proj = e(1, 2, h, w)
        self.out_ctx = e(1, 196, h, w)
        self.out_img = e(1, 3, h, w)
        self.host = dict(frame=self.frames[0].cpu(), filt=self.filters[0].cpu(), depth=self.depth[0].cpu(),
                         flow=self.flows[0][1].cpu())


def main():
    args = parse()
    import torch
    import vfidkr_amd  # noqa: F401
    from vfidkr_amd import cabi, runner, synthetic as S

    rank, local_rank, world = runner.init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    dev = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    cabi.lib()                                   # OSError here if libvfi_hip.so is missing

    h, w = S.padded_size(args.height, args.width)
    wl = Workload(torch, S, dev, h, w, args.flow_model, S.SEED + rank)
    px = wl.px
    fi196_events = []

    def fi(img, flow, filt, out):
        err = cabi.filterinterp_forward_ori(img, flow, filt, out, direct=args.direct)
        assert err == 0, err

    def step(i, record=False):
        for d in range(2):
            for a, b in wl.corr[d]:
                cabi.correlation_forward(a, b, 4, 1, 4, 1, 1)
            for ti in range(len(TIMES)):
                err = cabi.depthflowprojection_forward(wl.flows[d][ti], wl.depth[d], wl.count, wl.proj, 1)
                assert err == 0, err
                if record:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                fi(wl.ctx[d], wl.proj, wl.filters[d], wl.out_ctx)
                if record:
                    e1.record()
                    fi196_events.append((e0, e1))
                fi(wl.frames[d], wl.proj, wl.filters[d], wl.out_img)

    for i in range(args.warmup):
        step(i)
    elapsed = runner.timed_region(lambda i: step(i, record=True), args.steps, dev)
    frames_total = runner.total_units(len(TIMES) * args.steps, dev)
    value = frames_total / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel, from HIP events recorded inside the timed region
    torch.cuda.synchronize(dev)
    fi196_ms = sum(a.elapsed_time(b) for a, b in fi196_events) / max(1, len(fi196_events))
    fi196_bytes = 1640.0 * px
    achieved = fi196_bytes / (fi196_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "fi196_traffic.json")
    if os.path.exists(tpath) and not args.direct and args.flow_model == "smooth":
        with open(tpath) as fh:
            traffic = json.load(fh).get("hbm_bytes_per_launch")     # rocprofv3 --pmc passes, see profiles/README.md
    roofline = {"kernel": "fi_forward_ori_lds (FilterInterpolation _ori forward, C=196, fs=4)"
                if not args.direct else "fi_forward_ori_direct<true> (C=196)",
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": fi196_bytes, "avg_launch_ms": round(fi196_ms, 4),
                "launches_timed": len(fi196_events)}

    out = {
        "metric": "interpolated frames/sec at 1080p (hot path only: correlation + DepthFlowProjection + "
                  "FilterInterpolation of DAIN_slowmotion x4)",
        "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "DAIN_slowmotion x4 hot path, one %dx%d pair padded to %dx%d per step per GPU: "
                               "10 correlation(pad4,k1,md4) + 6 DepthFlowProjection(fillhole) + "
                               "6 FilterInterpolation(C=196) + 6 FilterInterpolation(C=3); 3 frames/step"
                               % (args.height, args.width, h, w),
                   "flow_model": args.flow_model, "filter_size": 4, "batch": 1,
                   "parallelism": "replicas x%d (one pair per GPU, no collective)" % world},
        "roofline": roofline,
    }

    if rank == 0:
        out["gate"] = gate_measurement(torch, cabi, wl, dev, args)
        out["fp16_storage"] = fp16_storage_measurement(torch, cabi, wl, dev, fi196_ms, ms_per_step)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], out["parity"] = cpu_baseline(torch, cabi, wl, dev, args)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def gate_measurement(torch, cabi, wl, dev, args, iters=50):
    """North-star gate: 2 x FilterInterpolation(C=3) + 2 x FlowProjection at 1080p, kernel time by
    HIP events on the launch stream; algorithmic bytes 96 B/px and 20 B/px."""
    px = wl.px

    def timed(fn, n=iters):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n

    flow = wl.flows[0][1]

    def fi3():
        assert cabi.filterinterp_forward_ori(wl.frames[0], flow, wl.filters[0], wl.out_img, direct=args.direct) == 0

    def fp_as_called():
        assert cabi.flowprojection_forward(flow, wl.count, wl.proj, 1) == 0

    fi3_ms = timed(fi3)
    fp_ms = timed(fp_as_called)
    total_ms = 2 * fi3_ms + 2 * fp_ms
    gbytes = (2 * 96.0 + 2 * 20.0) * px / 1e9
    return {"what": "2 x FilterInterpolation(C=3) + 2 x FlowProjection(fillhole; all its launches), %dx%d"
                    % (wl.h, wl.w),
            "fi_c3_ms": round(fi3_ms, 4), "fi_c3_GBps": round(96.0 * px / fi3_ms / 1e6, 1),
            "flowproj_ms": round(fp_ms, 4), "flowproj_GBps": round(20.0 * px / fp_ms / 1e6, 1),
            "total_ms": round(total_ms, 4), "algorithmic_GB": round(gbytes, 4),
            "achieved_GBps": round(gbytes / (total_ms * 1e-3), 1),
            "frac_of_8TBps": round(gbytes / (total_ms * 1e-3) / HBM_PEAK_GBS, 4), "target_frac": 0.5}


def fp16_storage_measurement(torch, cabi, wl, dev, fi196_ms, ms_per_step, iters=20):
    """BASELINE.json configs[2] names "fp16 storage / fp32 accum": the dominant launch with the context
    tensor and its output stored as fp16 (856 B/px algorithmic).  Reported beside the fp32
    headline, never as `value`."""
    ctx16 = wl.ctx[0].to(torch.float16)
    out16 = torch.empty_like(ctx16)
    flow, filt = wl.flows[0][1], wl.filters[0]

    def run():
        assert cabi.filterinterp_forward_ori_f16(ctx16, flow, filt, out16) == 0
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / iters
    gbs = 856.0 * wl.px / ms / 1e6
    step_ms = ms_per_step - 6.0 * (fi196_ms - ms)
    return {"kernel": "fi_forward_ori_lds_f16 (C=196, image and output fp16, flow / filter / arithmetic fp32)",
            "avg_launch_ms": round(ms, 4), "algorithmic_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / HBM_PEAK_GBS, 4),
            "frames_per_s_if_the_6_context_launches_used_it": round(3.0 / (step_ms * 1e-3), 1)}


def cpu_baseline(torch, cabi, wl, dev, args):
    """Times the CPU oracle (a port of the reference's arithmetic; the reference has no CPU path)
    on a bounded sample and scales it to one step; also reports parity of the GPU result on it."""
    import numpy as np
    from oracle import cpu_oracle as oracle
    threads = args.cpu_threads or min(16, os.cpu_count() or 1)
    oracle.set_num_threads(threads)
    hst = wl.host
    frame, filt, depth, flow = (hst[k].numpy() for k in ("frame", "filt", "depth", "flow"))
    ctx_full = wl.ctx[0].cpu().numpy()
    csel = 16                                   # channels kept for the parity check below

    # one (direction, t) unit of the step at full size: 1/6 of the projection + warping work
    t0 = time.perf_counter()
    proj, _ = oracle.depthflowproj_fwd(flow, depth, 1)
    t_dfp = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref_full = oracle.filterinterp_ori_fwd(ctx_full, proj, filt, fmad=1, nthreads=threads)
    t_fi196 = time.perf_counter() - t0
    ref_ctx = ref_full[:, :csel].copy()
    del ref_full, ctx_full
    t0 = time.perf_counter()
    ref_img = oracle.filterinterp_ori_fwd(frame, proj, filt, fmad=1, nthreads=threads)
    t_fi3 = time.perf_counter() - t0
    # the 5-level correlation of one direction: 1/2 of the correlation work
    t0 = time.perf_counter()
    corr_ref = None
    for a, b in wl.corr[0]:
        corr_ref = oracle.correlation_fwd(a.cpu().numpy(), b.cpu().numpy(), 4, 1, 4, 1, 1, order=0)
    t_corr = time.perf_counter() - t0
    # repeat the dominant call until the sample holds ~10 s of CPU work (threads x wall)
    reps = 0
    while (t_dfp + t_fi196 * (1 + reps) + t_fi3 + t_corr) * threads < 10.0 and reps < 5:
        t0 = time.perf_counter()
        oracle.filterinterp_ori_fwd(wl.ctx[1].cpu().numpy(), proj, filt, fmad=1, nthreads=threads)
        t_fi196 = min(t_fi196, time.perf_counter() - t0)
        reps += 1
    step_s = 6 * (t_dfp + t_fi196 + t_fi3) + 2 * t_corr
    base = {"value": round(len(TIMES) / step_s, 5), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "oracle/vfi_oracle.c (C restatement; the reference has no CPU path) at %dx%d on %d threads: "
                      "1 DepthFlowProjection (%.3fs, sequential scatter), 1 FilterInterpolation C=196 (%.3fs, best of "
                      "%d), 1 FilterInterpolation C=3 (%.3fs), 5-level correlation of one direction (%.3fs); "
                      "step = 6 x (proj + FI196 + FI3) + 2 x corr = %.2fs"
                      % (wl.h, wl.w, threads, t_dfp, t_fi196, reps + 1, t_fi3, t_corr, step_s)}

    # parity of the GPU path on the same sample (GPU fed the oracle's projected flow -> exact compare)
    gproj = torch.tensor(proj, device=dev)
    out = torch.empty((1, csel, wl.h, wl.w), dtype=torch.float32, device=dev)
    ctx_sel = torch.empty_like(out)                 # fresh dense strides (a channel slice keeps the 196-channel batch stride)
    ctx_sel.copy_(wl.ctx[0][:, :csel])
    assert cabi.filterinterp_forward_ori(ctx_sel, gproj, wl.filters[0], out, direct=args.direct) == 0
    out3 = torch.empty_like(wl.frames[0])
    assert cabi.filterinterp_forward_ori(wl.frames[0], gproj, wl.filters[0], out3, direct=args.direct) == 0
    cnt = torch.empty_like(wl.count)
    gp = torch.empty_like(wl.proj)
    assert cabi.depthflowprojection_forward(wl.flows[0][1], wl.depth[0], cnt, gp, 1) == 0
    a, b = wl.corr[0][-1]
    gcorr = cabi.correlation_forward(a, b, 4, 1, 4, 1, 1)
    torch.cuda.synchronize(dev)
    err3 = float(np.abs(out3.cpu().numpy() - ref_img).max())
    img8 = np.clip(np.round(out3.cpu().numpy() * 255.0), 0, 255)
    ref8 = np.clip(np.round(ref_img * 255.0), 0, 255)
    mse = float(np.mean((img8 - ref8) ** 2))
    psnr = float("inf") if mse == 0 else 20.0 * np.log10(255.0 / np.sqrt(mse))    # demo_MiddleBury.py:370-378
    parity = {"vs": "CPU oracle (fmad=1) on the same inputs",
              "filterinterp_c3_max_abs_err": err3,
              "filterinterp_ctx_max_abs_err": float(np.abs(out.cpu().numpy() - ref_ctx).max()),
              "depthflowproj_max_abs_err": float(np.abs(gp.cpu().numpy() - proj).max()),
              "correlation_max_abs_err": float(np.abs(gcorr.cpu().numpy() - corr_ref).max()),
              "psnr_db_uint8_frame": 99.0 if psnr == float("inf") else round(psnr, 2)}
    return base, parity


if __name__ == "__main__":
    main()
