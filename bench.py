#!/usr/bin/env python3
"""bench.py -- throughput of the frame-synthesis hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (starts its own N ranks when N > 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workloads
  slowmo1080 (default; BASELINE.json metric "interpolated frames/sec at 1080p", configs[2]): one STEP is the
      hot path of DAIN_slowmotion x4 for ONE 1080p frame pair per GPU (padded 1152x1984,
      demo_MiddleBury.py:294-310), i.e. exactly the native calls networks/DAIN_slowmotion.py:147-183 and
      PWCNet/PWCNet.py:230-300 make for it:
          10 x correlation forward   (5 pyramid levels x 2 directions; pad=4,k=1,md=4,s1=s2=1)
           6 x DepthFlowProjection   (2 directions x t in {0.25,0.5,0.75}; fillhole=1) -- the networks' two
                                     `FlowProject(list, depth)` calls back to back (networks/DAIN_slowmotion.py:156-159,
                                     301-307) as ONE call of the library's list form (fused.FlowProject_directions)
           6 x FilterInterpolation   on the 196-channel context tensor
           6 x FilterInterpolation   on the 3-channel frame
      in the reference's order (all correlations, FlowProject of both directions, then per time offset the two context
      warps and the two frame warps), and yields 3 interpolated frames.  Weak scaling: every rank has its own pair.
  vimeo64 (BASELINE.json configs[3]): 64 Vimeo-90K triplets (256x448 padded to 320x512) through the DAIN x2 hot
      path (per pair 10 correlation + 2 FlowProjection + 2 FilterInterpolation C=3, networks/DAIN.py:198-238), the
      64 pairs sharded over the ranks with runner.shard_pairs; one STEP = every rank's shard once = 64 frames.
      Strong scaling.
The convolutional sub-networks around the path (PWC-Net convs, MegaDepth, context/rectify nets: stock MIOpen
work) are outside this repo's scope and are NOT part of a step -- `value` is hot-path frames/s.
All inputs are synthetic (vfidkr_amd/synthetic.py), float32, resident in HBM before timing.  Ranks share nothing:
the only communication is the timing join (barrier, MAX of wall time, SUM of frames) over gloo.

Every native call goes through the C ABI of libvfi_hip.so (ctypes, vfidkr_amd/cabi.py).  The JSON line also carries
  roofline          dominant kernel (FilterInterpolation, C=196): algorithmic bytes per launch (1640 B/pixel,
                    SURVEY 8d) / mean launch time from HIP events on the launch stream INSIDE the timed region
                    (each launch streams 3.6 GB: nothing of it is cache resident), vs 8 TB/s.  `traffic` = HBM
                    bytes per launch from the rocprofv3 PMC passes committed under profiles/ for exactly this
                    (frame size, flow model), else null.
  roofline_quarter  the same launch on the SURVEY-8d-literal "quarter" flow field (measured after the timed region)
  gate              north-star gate: 2 x FilterInterpolation(C=3) + 2 x FlowProjection at 1080p (530 MB
                    algorithmic), `cold` = every call on a buffer set that has left the 256 MB Infinity Cache
                    (rotation through > 512 MB of sets), `hot` = the same set every call
  best_schedule     the same work, same bits, scheduled the way the library recommends: both directions' FlowProject lists
                    in ONE call, the three context warps of a direction in shared-window launches, one HIP stream per flow
                    direction (fused.FlowProject_directions / FilterInterpolate_ctx_all / DirectionStreams)
  as_called         the step through the reference-named pybind modules (ext/*.so) with the REFERENCE wrappers' allocation
                    semantics: every output and count allocated and zero-filled per call (FilterInterpolationLayer.py:34,
                    DepthFlowProjectionLayer.py:35-36), correlation outputs allocated by the binding
  value_spread      min / median / max of five repetitions of the timed region
  cpu_baseline      the CPU oracle (oracle/, a port: kind "port") timed on the host cores on a bounded sample
                    of the same workload (one of the step's six units, every call of it run in full) at 1 thread and at
                    all cores, rank 0 at N=1 only
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
L3_BYTES = 256 * 2 ** 20        # Infinity Cache
TIMES = (0.25, 0.5, 0.75)       # x4 slow motion: numFrames = 3 (DAIN_slowmotion.py:29)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="slowmo1080", choices=["slowmo1080", "vimeo64"])
    ap.add_argument("--flow-model", default="smooth", choices=["smooth", "quarter", "uniform1"],
                    help="synthetic flow field of the timed step (vfidkr_amd/synthetic.py); 'quarter' = SURVEY 8d to the letter")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the timed region (no gate / quarter / fp16 measurements)")
    ap.add_argument("--direct", action="store_true", help="force the direct-gather FilterInterpolation kernel")
    ap.add_argument("--storage", choices=("f32", "f16"), default="f32",
                    help="slowmo1080: f16 = BASELINE.json configs[2], frames / context / correlation features and their "
                         "outputs stored as fp16, flows, filters, depth and all arithmetic fp32")
    ap.add_argument("--vimeo-batch", type=int, default=3, help="vimeo64: triplets per call (the reference's PWC-Net allows 3)")
    ap.add_argument("--streams", type=int, default=None, choices=(1, 2, 3, 4, 6, 8),
                    help="slowmo1080: 2 (or more) = one HIP stream per flow direction (the two directions are independent chains "
                         "of correlations, projections and warps); 1 = one stream, the reference's call order.  vimeo64: the "
                         "batches of a step dealt over that many streams (branches of the step's graph).  Default: 1 (the "
                         "four-branch figure of vimeo64 is reported beside the headline as `four_branches`)")
    ap.add_argument("--no-graph", action="store_true", help="vimeo64: eager calls instead of one captured HIP graph per step")
    ap.add_argument("--stub-step", type=float, default=None, metavar="SECONDS",
                    help="plumbing test: a step is a sleep of SECONDS, no GPU is touched (tests/test_abi_and_host.py)")
    args = ap.parse_args(argv)
    if args.streams is None:
        args.streams = 1
    return args


# ------------------------------------------------------------------------------------------- workloads

class SlowmoPair:
    """All tensors of one 1080p frame pair of the DAIN_slowmotion x4 hot path, resident on `dev`."""

    def __init__(self, torch, S, dev, h, w, flow_model, seed):
        gen = S.generator(seed)
        self.h, self.w, self.px = h, w, h * w
        self.sigma = 8.0 * (w / 1984.0)
        self.frames = [S.frames(1, h, w, gen).to(dev) for _ in range(2)]
        self.ctx = [S.context(1, 196, h, w, gen).to(dev) for _ in range(2)]
        self.filters = [S.filters(1, h, w, gen).to(dev) for _ in range(2)]
        self.depth = [S.depth_weight(1, h, w, gen).to(dev) for _ in range(2)]
        base = [S.flow(1, h, w, self.sigma, gen, flow_model) for _ in range(2)]
        # forward_flownets: one flow per time offset, scaled by t (DAIN_slowmotion.py:214-215)
        self.flows = [[(base[d] * (2.0 * t)).contiguous().to(dev) for t in TIMES] for d in range(2)]
        self.corr = [[(a.to(dev), b.to(dev)) for a, b in S.correlation_features(1, h, w, gen)] for _ in range(2)]
        e = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)      # noqa: E731
        self.count, self.proj = e(1, 1, h, w), e(1, 2, h, w)
        # (the list form writes every item's count plane in the same launch: one per projected flow)
        self.counts = [[e(1, 1, h, w) for _ in TIMES] for _ in range(2)]
        # FlowProject returns one projected flow per direction and time offset, all made before the first warp
        self.projs = [[e(1, 2, h, w) for _ in TIMES] for _ in range(2)]
        self.out_ctx, self.out_img = e(1, 196, h, w), e(1, 3, h, w)
        self.gen = gen

    def to_half_storage(self, torch):
        """configs[2]: what the network stores as activations becomes fp16; flows, filters and depth stay fp32"""
        self.frames = [t.half() for t in self.frames]
        self.ctx = [t.half() for t in self.ctx]
        self.corr = [[(a.half(), b.half()) for a, b in lv] for lv in self.corr]
        self.out_ctx, self.out_img = self.out_ctx.half(), self.out_img.half()
        torch.cuda.empty_cache()
        return self


class VimeoPair:
    """B 256x448 triplets' tensors for the DAIN x2 hot path (padded 320x512), one batch per call."""

    def __init__(self, torch, S, dev, seed, batch=1):
        gen = S.generator(seed)
        h, w = S.padded_size(256, 448)
        self.h, self.w, self.batch = h, w, batch
        B = batch
        self.frames = [S.frames(B, h, w, gen).to(dev) for _ in range(2)]
        self.filters = [S.filters(B, h, w, gen).to(dev) for _ in range(2)]
        self.flows = [(S.flow(B, h, w, 2.0, gen, "smooth") * 0.5).contiguous().to(dev) for _ in range(2)]   # time_offset 0.5
        self.corr = [[(a.to(dev), b.to(dev)) for a, b in S.correlation_features(B, h, w, gen)] for _ in range(2)]
        e = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)      # noqa: E731
        self.count, self.proj, self.out = e(B, 1, h, w), e(B, 2, h, w), [e(B, 3, h, w) for _ in range(2)]


def hip_timed(torch, dev, fn, iters, nsets=1, warm=3):
    """mean ms per call of fn(i % nsets) from HIP events on the current stream"""
    for i in range(max(warm, nsets)):
        fn(i % nsets)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for i in range(iters):
        fn(i % nsets)
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / iters


# ------------------------------------------------------------------------------------------- one rank

def run_rank(args):
    import torch
    import vfidkr_amd  # noqa: F401
    from vfidkr_amd import runner

    if args.stub_step is None:
        # before the process group: a rank without a device leaves with a message instead of hanging the others in a barrier
        rank, local_rank, world = runner.dist_env()
        n = torch.cuda.device_count()
        # (ranks of an external launcher may share a device, local_rank % n: the one-GPU rehearsal; our own launcher insists on one each)
        if n < 1 or (os.environ.get("VFI_BENCH_OWN_RANKS") == "1" and n < world):
            print("bench.py: rank %d: --gpus %d but %d GPU(s) visible" % (rank, max(args.gpus, world), n), file=sys.stderr)
            return 2
    rank, local_rank, world = runner.init_distributed()
    if args.stub_step is not None:
        return run_stub(args, runner, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    from vfidkr_amd import cabi, synthetic as S
    dev = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    cabi.lib()                                   # OSError here if libvfi_hip.so is missing
    if args.workload == "vimeo64":
        out = run_vimeo64(args, torch, cabi, runner, S, dev, rank, world)
    else:
        out = run_slowmo(args, torch, cabi, runner, S, dev, rank, world)
    if rank == 0:
        print(json.dumps(out), flush=True)
    runner.shutdown()
    return 0


def run_stub(args, runner, rank, world):
    elapsed = runner.timed_region(lambda i: time.sleep(args.stub_step), args.steps)
    total = runner.total_units(args.steps)
    if rank == 0:
        print(json.dumps({"metric": "stub steps/s (launcher plumbing test, no GPU work)", "value": round(total / elapsed, 3),
                          "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "stub", "config": {"workload": "stub"}}), flush=True)
    runner.shutdown()
    return 0


def run_vimeo64(args, torch, cabi, runner, S, dev, rank, world):
    n_pairs = 64
    mine = list(runner.shard_pairs(n_pairs, rank, world))
    # the rank's triplets go through in batches of at most three: PWC-Net's pre-built warp grid holds B_MAX = 3
    # (PWCNet/PWCNet.py:144, 178; SURVEY.md 8d "cfg4 ... run as 8 x B=1 or B<=3 batches"); --vimeo-batch 1 = one by one
    bmax = max(1, min(3, args.vimeo_batch))
    sizes = [min(bmax, len(mine) - k) for k in range(0, len(mine), bmax)]
    pairs = [VimeoPair(torch, S, dev, S.SEED + 1000 + mine[sum(sizes[:j])], batch=sizes[j]) for j in range(len(sizes))]

    # --streams N: the batches are independent (every batch has its own count / proj / output tensors, the library one
    # projection workspace per stream), so they can be dealt over N streams -- N parallel branches of the step's graph.
    # The launches of a 320x512 batch fill a fraction of the GPU each.
    def make_runner(nbranch):
        branches = [torch.cuda.Stream(dev) for _ in range(nbranch)] if nbranch > 1 else []

        def step(_i):
            cur = torch.cuda.current_stream(dev)
            for br in branches:
                br.wait_stream(cur)
            for j, p in enumerate(pairs):
                with torch.cuda.stream(branches[j % len(branches)] if branches else cur):
                    for d in range(2):
                        for a, b in p.corr[d]:
                            cabi.correlation_forward(a, b, 4, 1, 4, 1, 1)
                    for d in range(2):                          # DAIN.FlowProject + DAIN.FilterInterpolate (DAIN.py:218-238)
                        assert cabi.flowprojection_forward(p.flows[d], p.count, p.proj, 1) == 0
                        assert cabi.filterinterp_forward_ori(p.frames[d], p.proj, p.filters[d], p.out[d]) == 0
            for br in branches:
                cur.wait_stream(br)

        for i in range(max(1, args.warmup)):
            step(i)
        # Hundreds of short launches per step (14 per batch): the host's ctypes calls, not the GPU, would set the pace.  The step
        # is captured once into a HIP graph (every entry point is capturable: no host synchronisation, workspaces already
        # sized by the warm-up) and replayed; --no-graph times the eager calls.
        launch, run = "eager calls", step
        if not args.no_graph:
            try:
                side = torch.cuda.Stream(dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    step(0)                                     # warm-up on the capture stream: its workspaces
                    side.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=side):
                        step(0)
                torch.cuda.synchronize(dev)
                launch, run = "one HIP graph per step (captured once, replayed)", lambda _i: graph.replay()
                run(0)
            except Exception as exc:                            # capture unsupported: say so and time the eager calls
                launch = "eager calls (graph capture failed: %s)" % type(exc).__name__
                run = step
        return launch, run

    launch, run = make_runner(args.streams)
    elapsed = runner.timed_region(run, args.steps, dev)
    frames_total = runner.total_units(len(mine) * args.steps)
    h, w = pairs[0].h, pairs[0].w
    extra = {}
    if args.streams == 1 and not args.no_extras:
        # the same step with its independent batches dealt over four parallel branches of the graph (a schedule change, not
        # kernel work: reported beside the one-branch headline, which stays comparable from round to round)
        _, run4 = make_runner(4)
        el4 = runner.timed_region(run4, args.steps, dev)
        extra["four_branches"] = {"ms_per_step": round(el4 / args.steps * 1e3, 4),
                                  "frames_per_s": round(runner.total_units(len(mine) * args.steps) / el4, 1)}
    return {**extra, **{
        "metric": "interpolated frames/sec, 64 Vimeo-90K triplets (hot path only: correlation + FlowProjection + "
                  "FilterInterpolation of DAIN x2)",
        "value": round(frames_total / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "vimeo64: 64 triplets 256x448 padded to %dx%d, DAIN x2 hot path per pair (10 correlation + "
                               "2 FlowProjection(fillhole) + 2 FilterInterpolation(C=3)), B<=%d per call; 1 frame per pair"
                               % (h, w, bmax),
                   "pairs": n_pairs, "batches_per_rank_0": sizes, "pairs_per_rank": [len(runner.shard_pairs(n_pairs, r, world)) for r in range(world)],
                   "filter_size": 4, "launch": launch, "streams": args.streams,
                   "parallelism": "replicas x%d (pairs sharded, no collective)" % world},
    }}


def run_slowmo(args, torch, cabi, runner, S, dev, rank, world):
    h, w = S.padded_size(args.height, args.width)
    wl = SlowmoPair(torch, S, dev, h, w, args.flow_model, S.SEED + rank)
    px = wl.px
    fi196_events = []
    half = args.storage == "f16"
    if half:
        wl.to_half_storage(torch)
    fi_call = cabi.filterinterp_forward_ori_f16 if half else cabi.filterinterp_forward_ori

    def fi(img, flow, filt, out):
        err = fi_call(img, flow, filt, out, direct=args.direct)
        assert err == 0, err

    flat_flows, flat_counts, flat_projs = wl.flows[0] + wl.flows[1], wl.counts[0] + wl.counts[1], wl.projs[0] + wl.projs[1]
    flat_depth = [wl.depth[0]] * len(TIMES) + [wl.depth[1]] * len(TIMES)

    def step(i, record=False):
        # the reference's order (networks/DAIN_slowmotion.py:147-171): both flow networks (their correlations), FlowProject
        # of both directions for every time offset, then per time offset FilterInterpolate_ctx and FilterInterpolate
        for d in range(2):
            for a, b in wl.corr[d]:
                cabi.correlation_forward(a, b, 4, 1, 4, 1, 1)
        # [FlowProject(cur_offset_outputs[0], depth_inv[0]), FlowProject(cur_offset_outputs[1], depth_inv[1])]
        # (networks/DAIN_slowmotion.py:156-159): the two lists back to back = one library call (fused.FlowProject_directions)
        err = cabi.flowprojection_forward_batch(flat_flows, flat_counts, flat_projs, 1, flat_depth)
        assert err == 0, err
        for ti in range(len(TIMES)):
            for d in range(2):
                if record:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                fi(wl.ctx[d], wl.projs[d][ti], wl.filters[d], wl.out_ctx)
                if record:
                    e1.record()
                    fi196_events.append((e0, e1))
            for d in range(2):
                fi(wl.frames[d], wl.projs[d][ti], wl.filters[d], wl.out_img)

    # The same 46 launches on two streams, one per flow direction: direction d's chain -- its correlations, FlowProject(d, t),
    # FilterInterpolate_ctx / FilterInterpolate on ctx d / frame d -- needs nothing of the other direction's
    # (networks/DAIN_slowmotion.py:147-183: the two directions meet only in the blend and the rectify network, after the
    # hot path).  No cross-stream event inside the step; the library keeps one projection workspace per stream, and each
    # stream has its own count plane and output tensors.  The short launches of one direction (0.84 ms per step when in a
    # row) run under the 196-channel warps of the other.
    from vfidkr_amd import fused
    lanes = fused.DirectionStreams(dev)
    out_ctx2, out_img2 = torch.empty_like(wl.out_ctx), torch.empty_like(wl.out_img)

    def step2(i, record=False):
        lanes.fork()
        for d, oc, oi in ((0, wl.out_ctx, wl.out_img), (1, out_ctx2, out_img2)):
            with lanes.direction(d):
                for a, b in wl.corr[d]:
                    cabi.correlation_forward(a, b, 4, 1, 4, 1, 1)
                err = cabi.flowprojection_forward_batch(wl.flows[d], wl.counts[d], wl.projs[d], 1, wl.depth[d])
                assert err == 0, err
                for ti in range(len(TIMES)):
                    if record and d == 0:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                    fi(wl.ctx[d], wl.projs[d][ti], wl.filters[d], oc)
                    if record and d == 0:
                        e1.record()
                        fi196_events.append((e0, e1))
                    fi(wl.frames[d], wl.projs[d][ti], wl.filters[d], oi)
        lanes.join()                                        # the step ends when both streams have

    run_step = step2 if args.streams == 2 else step
    for i in range(args.warmup):
        run_step(i)
    elapsed = runner.timed_region(lambda i: run_step(i, record=True), args.steps, dev)
    frames_total = runner.total_units(len(TIMES) * args.steps)
    value = frames_total / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel, from HIP events recorded inside the timed region
    torch.cuda.synchronize(dev)
    fi196_ms = sum(a.elapsed_time(b) for a, b in fi196_events) / max(1, len(fi196_events))
    kernel = ("fi_forward_ori_lds (FilterInterpolation _ori forward, C=196, fs=4)" if not args.direct
              else "fi_forward_ori_direct<true> (C=196)")
    if half:
        kernel = kernel.replace("fi_forward_ori_lds", "fi_forward_ori_lds_f16").replace("direct<true>", "direct_f16")
    roofline = roofline_block(kernel, px, fi196_ms, len(fi196_events),
                              traffic_lookup(h, w, args.flow_model, args.direct, "fi196_f16" if half else "fi196"), 856.0 if half else 1640.0)

    out = {
        "metric": "interpolated frames/sec at 1080p (hot path only: correlation + DepthFlowProjection + "
                  "FilterInterpolation of DAIN_slowmotion x4)",
        "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 arithmetic, f16 storage" if half else "f32", "data": "synthetic",
        "config": {"workload": "DAIN_slowmotion x4 hot path, one %dx%d pair padded to %dx%d per step per GPU: "
                               "10 correlation(pad4,k1,md4) + 6 DepthFlowProjection(fillhole; both FlowProject lists as one call) + "
                               "6 FilterInterpolation(C=196) + 6 FilterInterpolation(C=3); 3 frames/step"
                               % (args.height, args.width, h, w),
                   "flow_model": args.flow_model, "filter_size": 4, "batch": 1, "storage": args.storage,
                   "streams": args.streams,
                   "schedule": ("one HIP stream per flow direction (two 196-channel launches run side by side: roofline.avg_launch_ms "
                                "is the duration of one of them while it shares the GPU)" if args.streams == 2
                                else "one stream, the reference's call order"),
                   "parallelism": "replicas x%d (one pair per GPU, no collective)" % world},
        "roofline": roofline,
    }
    # five more repetitions of the timed region (every rank takes part: the region ends in a barrier): box-to-box and
    # run-to-run spread is of the order of a kernel improvement
    reps = [elapsed] + [runner.timed_region(lambda i: run_step(i), args.steps, dev) for _ in range(4)]
    fps = sorted(world * len(TIMES) * args.steps / t for t in reps)
    out["value_spread"] = {"repetitions": len(reps), "steps_each": args.steps, "min": round(fps[0], 2), "median": round(fps[len(fps) // 2], 2),
                           "max": round(fps[-1], 2), "unit": "frames/s", "note": "`value` is the first repetition"}
    if half:
        return out              # the side measurements and the CPU baseline belong to the fp32 headline
    if rank == 0 and not args.no_extras:
        if args.flow_model != "quarter":
            out["roofline_quarter"] = quarter_measurement(torch, cabi, S, wl, dev, args, kernel)
        out["gate"] = gate_measurement(torch, cabi, S, wl, dev, args)
        out["fp16_storage"] = fp16_storage_measurement(torch, cabi, S, dev, args, h, w, rank)
        out["correlation"] = correlation_measurement(torch, cabi, wl, dev)
        out["shared_window"] = shared_window_measurement(torch, cabi, wl, dev, args)
        out["best_schedule"] = best_schedule_block(out["shared_window"], value / world, max(5, args.steps // 2))
        out["as_called"] = as_called_measurement(torch, wl, dev, args, ms_per_step)
        if args.streams == 1:
            # the same launches on two streams (`--streams 2` makes this the timed region): reported beside the headline,
            # whose step keeps one stream and the reference's call order
            n2 = max(5, args.steps // 2)
            for i in range(2):
                step2(i)
            # (rank 0 alone measures the extras: no barrier here -- the other ranks have left the timed region for good)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(n2):
                step2(i)
            torch.cuda.synchronize(dev)
            el2 = time.perf_counter() - t0
            out["two_streams"] = {"schedule": "one HIP stream per flow direction: its correlations, projections and warps (the "
                                              "directions are independent chains, networks/DAIN_slowmotion.py:147-183)",
                                  "steps_timed": n2, "ms_per_step": round(el2 / n2 * 1e3, 4),
                                  "frames_per_s": round(len(TIMES) * n2 / el2, 1)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], out["parity"] = cpu_baseline(torch, cabi, wl, dev, args)
    return out


def roofline_block(kernel, px, ms, launches, traffic, bytes_per_px=1640.0):
    nbytes = bytes_per_px * px
    achieved = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": kernel, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "algorithmic_bytes_per_launch": nbytes,
            "avg_launch_ms": round(ms, 4), "launches_timed": launches}


def traffic_lookup(h, w, flow_model, direct, op="fi196"):
    """HBM bytes per call of `op` (fi196: the C=196 launch; fi196_f16: the same with fp16 storage; fi_c3; flowproj: all three
    launches of a call) from the
    committed rocprofv3 PMC passes -- only for the exact (frame size, flow model, kernel) they were collected on
    (profiles/README.md says how); anything else has no counter evidence: null."""
    path = os.path.join(ROOT, "profiles", "traffic_by_config.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        for e in json.load(fh).get("entries", []):
            if (e.get("h"), e.get("w"), e.get("flow_model"), bool(e.get("direct")), e.get("op", "fi196")) == (h, w, flow_model, bool(direct), op):
                return e.get("hbm_bytes_per_launch")
    return None


def quarter_measurement(torch, cabi, S, wl, dev, args, kernel, iters=12):
    """The dominant launch on SURVEY 8d's literal flow field: bilinear x4 of N(0, sigma^2) drawn at quarter resolution."""
    flow = (S.flow(1, wl.h, wl.w, wl.sigma, wl.gen, "quarter") * 1.0).to(dev)       # t = 0.5: the middle offset
    cnt, proj = torch.empty_like(wl.count), torch.empty_like(wl.proj)
    assert cabi.depthflowprojection_forward(flow, wl.depth[0], cnt, proj, 1) == 0

    def run(i):
        assert cabi.filterinterp_forward_ori(wl.ctx[i % 2], proj, wl.filters[0], wl.out_ctx, direct=args.direct) == 0
    ms = hip_timed(torch, dev, run, iters, nsets=2)
    r = roofline_block(kernel, wl.px, ms, iters, traffic_lookup(wl.h, wl.w, "quarter", args.direct))
    r["flow_model"] = "quarter"
    return r


def gate_measurement(torch, cabi, S, wl, dev, args, iters=60):
    """North-star gate: 2 x FilterInterpolation(C=3) + 2 x FlowProjection at 1080p; kernel time by HIP events on the
    launch stream; algorithmic bytes 96 B/px and 20 B/px.  cold: every call works on a set of buffers
    that was last touched > 512 MB of traffic ago; hot: the same set every call."""
    px, h, w = wl.px, wl.h, wl.w
    n_fi = int(2 * L3_BYTES / (96.0 * px)) + 2
    n_fp = int(2 * L3_BYTES / (20.0 * px)) + 2
    fi_sets = [(S.frames(1, h, w, wl.gen).to(dev), wl.flows[0][1].clone(), S.filters(1, h, w, wl.gen).to(dev),
                torch.empty((1, 3, h, w), device=dev)) for _ in range(n_fi)]
    fp_sets = [(wl.flows[0][1].clone(), torch.empty((1, 1, h, w), device=dev), torch.empty((1, 2, h, w), device=dev))
               for _ in range(n_fp)]

    def fi3(i):
        img, flow, filt, out = fi_sets[i]
        assert cabi.filterinterp_forward_ori(img, flow, filt, out, direct=args.direct) == 0

    def fp(i):
        flow, cnt, out = fp_sets[i]
        assert cabi.flowprojection_forward(flow, cnt, out, 1) == 0

    res = {"what": "2 x FilterInterpolation(C=3) + 2 x FlowProjection(fillhole; all its launches), %dx%d, flow model %s"
                   % (h, w, args.flow_model), "target_frac": 0.5,
           "rotation": "%d FilterInterpolation sets (%.0f MB each), %d FlowProjection sets (%.0f MB each)"
                       % (n_fi, 96.0 * px / 1e6, n_fp, 20.0 * px / 1e6)}
    gbytes = (2 * 96.0 + 2 * 20.0) * px / 1e9
    for name, nf, np_ in (("cold", n_fi, n_fp), ("hot", 1, 1)):
        fi3_ms = hip_timed(torch, dev, fi3, iters, nf)
        fp_ms = hip_timed(torch, dev, fp, iters, np_)
        total_ms = 2 * fi3_ms + 2 * fp_ms
        res[name] = {"fi_c3_ms": round(fi3_ms, 4), "fi_c3_GBps": round(96.0 * px / fi3_ms / 1e6, 1),
                     "flowproj_ms": round(fp_ms, 4), "flowproj_GBps": round(20.0 * px / fp_ms / 1e6, 1),
                     "total_ms": round(total_ms, 4), "achieved_GBps": round(gbytes / (total_ms * 1e-3), 1),
                     "frac_of_8TBps": round(gbytes / (total_ms * 1e-3) / HBM_PEAK_GBS, 4)}
    # `pair`: what a frame pair costs as the networks call it -- FlowProject of BOTH directions as one list (DAIN.py:215-220:
    # the two FlowProject calls back to back; fused.FlowProject_directions) + the two frame warps.  Same 530 MB.
    fl2 = S.flow(1, h, w, wl.sigma, wl.gen, args.flow_model).to(dev)
    pair_sets = [([fp_sets[i][0], fl2.clone()], [fp_sets[i][1], torch.empty_like(fp_sets[i][1])],
                  [fp_sets[i][2], torch.empty_like(fp_sets[i][2])]) for i in range(n_fp)]

    def fp_pair(i):
        fl, cn, out = pair_sets[i]
        assert cabi.flowprojection_forward_batch(fl, cn, out, 1) == 0

    res["pair"] = {"what": "2 x FilterInterpolation(C=3) + ONE FlowProjection call on the list of both directions' flows"}
    for name, nf, np_ in (("cold", n_fi, n_fp), ("hot", 1, 1)):
        fp2_ms = hip_timed(torch, dev, fp_pair, iters, np_)
        total_ms = 2 * res[name]["fi_c3_ms"] + fp2_ms
        res["pair"][name] = {"flowproj_both_directions_ms": round(fp2_ms, 4), "total_ms": round(total_ms, 4),
                             "achieved_GBps": round(gbytes / (total_ms * 1e-3), 1),
                             "frac_of_8TBps": round(gbytes / (total_ms * 1e-3) / HBM_PEAK_GBS, 4)}
    res["algorithmic_GB"] = round(gbytes, 4)
    res["frac_of_8TBps"] = res["cold"]["frac_of_8TBps"]             # the figure that counts: nothing cache resident
    # HBM-side bytes per call from the PMC passes under profiles/ (null for a configuration they were not collected on)
    res["traffic"] = {"fi_c3_bytes_per_call": traffic_lookup(h, w, args.flow_model, args.direct, "fi_c3"),
                      "flowproj_bytes_per_call": traffic_lookup(h, w, args.flow_model, False, "flowproj"),
                      "fi_c3_algorithmic_bytes": 96.0 * px, "flowproj_algorithmic_bytes": 20.0 * px}
    return res


def fp16_storage_measurement(torch, cabi, S, dev, args, h, w, rank):
    """BASELINE.json configs[2] ("fp16 storage / fp32 accum"): the SAME step with frames, context, correlation features
    and their outputs stored as fp16 -- `bench.py --storage f16` runs it as the timed region; here a short measured
    run of it is reported beside the fp32 headline, never as `value`.  Step time and launch time are measured."""
    wl = SlowmoPair(torch, S, dev, h, w, args.flow_model, S.SEED + rank).to_half_storage(torch)
    events = []

    def step(i, record=False):
        for d in range(2):
            for a, b in wl.corr[d]:
                cabi.correlation_forward(a, b, 4, 1, 4, 1, 1)
        for d in range(2):
            assert cabi.flowprojection_forward_batch(wl.flows[d], wl.counts[d], wl.projs[d], 1, wl.depth[d]) == 0
        for ti in range(len(TIMES)):
            for d in range(2):
                if record:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                assert cabi.filterinterp_forward_ori_f16(wl.ctx[d], wl.projs[d][ti], wl.filters[d], wl.out_ctx, direct=args.direct) == 0
                if record:
                    e1.record()
                    events.append((e0, e1))
            for d in range(2):
                assert cabi.filterinterp_forward_ori_f16(wl.frames[d], wl.projs[d][ti], wl.filters[d], wl.out_img, direct=args.direct) == 0

    for i in range(3):
        step(i)
    torch.cuda.synchronize(dev)
    steps = max(5, args.steps // 2)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i, record=True)
    torch.cuda.synchronize(dev)
    step_ms = (time.perf_counter() - t0) / steps * 1e3
    ms = sum(a.elapsed_time(b) for a, b in events) / len(events)
    gbs = 856.0 * wl.px / ms / 1e6
    return {"kernel": "fi_forward_ori_lds_f16 (C=196, image and output fp16, flow / filter / arithmetic fp32)",
            "avg_launch_ms": round(ms, 4), "algorithmic_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": traffic_lookup(h, w, args.flow_model, False, "fi196_f16"), "algorithmic_bytes_per_launch": 856.0 * wl.px,
            "steps_timed": steps, "ms_per_step": round(step_ms, 4), "frames_per_s": round(len(TIMES) / (step_ms * 1e-3), 1),
            "step": "measured: 10 half correlations + 6 DepthFlowProjection (fp32) + 6 + 6 half-storage FilterInterpolation"}


def correlation_measurement(torch, cabi, wl, dev, iters=100):
    """The step's ten correlation calls (5 pyramid levels x 2 flow networks) one by one, and as five calls that take the two
    networks' tensors of a level together (vfi_correlation_forward_pair / fused.corr_pair: at the coarse levels a launch is
    latency); same bits."""
    def singles(i):
        for d in range(2):
            for a, b in wl.corr[d]:
                cabi.correlation_forward(a, b, 4, 1, 4, 1, 1)

    def pairs(i):
        for (a0, b0), (a1, b1) in zip(wl.corr[0], wl.corr[1]):
            cabi.correlation_forward_pair(a0, b0, a1, b1, 4, 1, 4, 1, 1)
    nbytes = 2 * sum((2 * a.size(1) + 81) * 4.0 * a.size(2) * a.size(3) for a, _ in wl.corr[0])
    # (launches of 12-25 us each: the host has to keep ahead of the GPU, and a busy host core shows up as GPU time -- one run
    #  of five gave 336 us for the pair calls on a box that gave 127 otherwise -- so: the best of three short runs each)
    one = min(hip_timed(torch, dev, singles, max(10, iters // 3)) for _ in range(3))
    two = min(hip_timed(torch, dev, pairs, max(10, iters // 3)) for _ in range(3))
    return {"ten_calls_ms": round(one, 4), "five_pair_calls_ms": round(two, 4), "best_of": 3, "algorithmic_bytes": nbytes,
            "pair_algorithmic_GBps": round(nbytes / (two * 1e-3) / 1e9, 1)}


def shared_window_measurement(torch, cabi, wl, dev, args):
    """The same step with the three time offsets of a direction warped by ONE call (vfi_filterinterp_forward_ori_multi /
    fused.FilterInterpolate_ctx_all: a shared-window launch for two of the flows + a single-flow launch) instead of three
    FilterInterpolation calls: same outputs bit for bit, 4072 instead of 3 x 1640 algorithmic bytes per pixel.  Reported
    beside the headline, whose step keeps the reference's call sequence."""
    nt = len(TIMES)
    projs = wl.projs
    outs = [torch.empty_like(wl.out_ctx) for _ in range(nt)]
    imgs = [torch.empty_like(wl.out_img) for _ in range(nt)]         # the frame warps go through the same entry point (3 x 134 -> 110 us)
    events = []

    def step(i, record=False):
        for d in range(2):
            for a, b in wl.corr[d]:
                cabi.correlation_forward(a, b, 4, 1, 4, 1, 1)
        for d in range(2):
            assert cabi.flowprojection_forward_batch(wl.flows[d], wl.counts[d], projs[d], 1, wl.depth[d]) == 0
        for d in range(2):
            if record:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            assert cabi.filterinterp_forward_ori_multi(wl.ctx[d], projs[d], wl.filters[d], outs) == 0
            if record:
                e1.record()
                events.append((e0, e1))
            assert cabi.filterinterp_forward_ori_multi(wl.frames[d], projs[d], wl.filters[d], imgs) == 0

    for i in range(3):
        step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    steps = max(5, args.steps // 2)
    for i in range(steps):
        step(i, record=True)
    torch.cuda.synchronize(dev)
    step_ms = (time.perf_counter() - t0) / steps * 1e3
    ms = sum(a.elapsed_time(b) for a, b in events) / len(events)
    # the library forms launches of two flows (one staged window, two outputs) + a single-flow launch for an odd one
    npair, nsingle = nt // 2, nt % 2
    nbytes = (npair * (2 * 2 + 16 + 196 + 2 * 196) + nsingle * (2 + 16 + 196 + 196)) * 4.0 * wl.px
    gbs = nbytes / (ms * 1e-3) / 1e9
    res = {"kernel": "vfi_filterinterp_forward_ori_multi on %d flows (C=196) = %d x fi_forward_ori_multi<2> (one staged window, two outputs)%s"
                     % (nt, npair, " + 1 x fi_forward_ori_lds" if nsingle else ""),
           "avg_call_ms": round(ms, 4), "ms_per_output": round(ms / nt, 4), "algorithmic_bytes_per_call": nbytes,
           "achieved": round(gbs, 1), "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
           "ms_per_step": round(step_ms, 4), "frames_per_s": round(nt / (step_ms * 1e-3), 1)}

    # ... and with one HIP stream per flow direction as well (fused.DirectionStreams; see two_streams)
    from vfidkr_amd import fused
    lanes = fused.DirectionStreams(dev)
    imgs2 = [torch.empty_like(wl.out_img) for _ in range(nt)]
    outs2 = [torch.empty_like(wl.out_ctx) for _ in range(nt)]

    def step_lanes(i):
        lanes.fork()
        for d, oc, oi in ((0, outs, imgs), (1, outs2, imgs2)):
            with lanes.direction(d):
                for a, b in wl.corr[d]:
                    cabi.correlation_forward(a, b, 4, 1, 4, 1, 1)
                assert cabi.flowprojection_forward_batch(wl.flows[d], wl.counts[d], projs[d], 1, wl.depth[d]) == 0
                assert cabi.filterinterp_forward_ori_multi(wl.ctx[d], projs[d], wl.filters[d], oc) == 0
                assert cabi.filterinterp_forward_ori_multi(wl.frames[d], projs[d], wl.filters[d], oi) == 0
        lanes.join()

    for i in range(2):
        step_lanes(i)
    reps = []
    for _ in range(5):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            step_lanes(i)
        torch.cuda.synchronize(dev)
        reps.append((time.perf_counter() - t0) / steps * 1e3)
    lanes_ms = reps[0]
    res["two_streams"] = {"ms_per_step": round(lanes_ms, 4), "frames_per_s": round(nt / (lanes_ms * 1e-3), 1)}
    res["_two_streams_reps_ms"] = reps
    return res


def best_schedule_block(shared, value, steps):
    """The BASELINE metric on the schedule the library recommends (same work, bit-identical outputs:
    tests/test_gpu_parity.py::test_two_streams_same_bits_as_one, test_filterinterp_multi_flow*): one HIP stream per flow
    direction (fused.DirectionStreams), per direction its correlations, ONE FlowProject call on its list of flows
    (fused.FlowProject), the three context warps and the three frame warps as shared-window launches (fused.FilterInterpolate_ctx_all)."""
    reps = sorted(shared.pop("_two_streams_reps_ms"))
    nt = len(TIMES)
    fps = [nt / (ms * 1e-3) for ms in reps]
    return {"schedule": "one HIP stream per flow direction; per direction: 5 correlations, FlowProject(list of 3 flows) as one call, "
                        "FilterInterpolate_ctx and the frame warps for the three time offsets as shared-window launches",
            "frames_per_s": round(fps[len(fps) // 2], 1), "ms_per_step": round(reps[len(reps) // 2], 4), "unit": "frames/s",
            "spread": {"repetitions": len(reps), "steps_each": steps, "min": round(fps[-1], 1), "median": round(fps[len(fps) // 2], 1),
                       "max": round(fps[0], 1)},
            "vs_value": round(fps[len(fps) // 2] / value, 3)}


def as_called_measurement(torch, wl, dev, args, cabi_ms):
    """The step as an UNCHANGED reference caller pays for it: through the reference-named pybind modules (ext/*.so: same
    function names and positional signatures as my_package/*_cuda.cc and correlation_cuda.cc) with the reference wrappers'
    allocation semantics -- every call allocates its output (and count) and zero-fills it, as FilterInterpolationLayer.py:34,
    DepthFlowProjectionLayer.py:35-36 do (this library's kernels write every element, so the fill is pure cost: 1.79 GB per
    196-channel call); the correlation binding sizes and zero-fills rbot1 / rbot2 / output itself (correlation.py:21-26).
    SURVEY 8d 'Timing': the as-called figure beside the kernel-only one."""
    import correlation_cuda
    import depthflowprojection_cuda
    import filterinterpolation_cuda
    h, w = wl.h, wl.w

    def fi(img, flow, filt):
        out = torch.zeros_like(img)
        assert filterinterpolation_cuda.FilterInterpolationLayer_gpu_forward_ori(img, flow, filt, out) == 0
        return out

    def dfp(flow, depth):
        count = torch.zeros((1, 1, h, w), dtype=torch.float32, device=dev)
        out = torch.zeros_like(flow)
        assert depthflowprojection_cuda.DepthFlowProjectionLayer_gpu_forward(flow, depth, count, out, 1) == 0
        return out

    def step(i):
        for d in range(2):
            for a, b in wl.corr[d]:
                rb1, rb2, out = a.new(), b.new(), a.new()
                correlation_cuda.forward(a, b, rb1, rb2, out, 4, 1, 4, 1, 1, 1)
        projs = [[dfp(wl.flows[d][ti], wl.depth[d]) for ti in range(len(TIMES))] for d in range(2)]
        for ti in range(len(TIMES)):
            for d in range(2):
                fi(wl.ctx[d], projs[d][ti], wl.filters[d])
            for d in range(2):
                fi(wl.frames[d], projs[d][ti], wl.filters[d])

    for i in range(2):
        step(i)
    torch.cuda.synchronize(dev)
    n = max(5, args.steps // 2)
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / n * 1e3
    return {"what": "the 28 reference-named binding calls of the step (10 correlation_cuda.forward, 6 DepthFlowProjectionLayer_gpu_forward, "
                    "12 FilterInterpolationLayer_gpu_forward_ori), each with freshly allocated, zero-filled outputs as the reference's "
                    "Layer wrappers make them", "steps_timed": n, "ms_per_step": round(ms, 4),
            "frames_per_s": round(len(TIMES) / (ms * 1e-3), 1), "c_abi_ms_per_step": round(cabi_ms, 4),
            "zero_fill_and_allocation_ms": round(ms - cabi_ms, 4)}


def cpu_baseline(torch, cabi, wl, dev, args):
    """Times the CPU oracle (a port of the reference's arithmetic; the reference has no CPU path) on one (direction, t)
    unit of the step -- DepthFlowProjection, FilterInterpolation on the frame and on the WHOLE 196-channel context tensor,
    the 5-level correlation of one direction -- and scales it to a step (6 units + 2 correlation pyramids).  At every
    host core: median of 5 runs per call (the 196-channel call included: it is 85 % of the step).  At one thread: median
    of 5 for the short calls, ONE run of the 196-channel call (~11 s).  Also reports the parity of the GPU results on the
    same inputs, the whole context tensor included."""
    import numpy as np
    from oracle import cpu_oracle as oracle
    ncpu = os.cpu_count() or 1
    frame, filt = wl.frames[0].cpu().numpy(), wl.filters[0].cpu().numpy()
    depth, flow = wl.depth[0].cpu().numpy(), wl.flows[0][1].cpu().numpy()
    ctx = wl.ctx[0].cpu().numpy()
    corr_np = [(a.cpu().numpy(), b.cpu().numpy()) for a, b in wl.corr[0]]
    proj, _ = oracle.depthflowproj_fwd(flow, depth, 1)

    def med(fn, n=5):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), r

    def one_config(threads, runs196):
        oracle.set_num_threads(threads)
        t_dfp, _ = med(lambda: oracle.depthflowproj_fwd(flow, depth, 1))            # sequential scatter: one thread whatever `threads`
        t_fi3, ref_img = med(lambda: oracle.filterinterp_ori_fwd(frame, proj, filt, fmad=1, nthreads=threads))
        t_fic, ref_ctx = med(lambda: oracle.filterinterp_ori_fwd(ctx, proj, filt, fmad=1, nthreads=threads), runs196)
        t_corr, corr_ref = med(lambda: [oracle.correlation_fwd(a, b, 4, 1, 4, 1, 1, order=0) for a, b in corr_np][-1])
        step_s = 6 * (t_dfp + t_fic + t_fi3) + 2 * t_corr
        return dict(threads=threads, dfp=t_dfp, fi3=t_fi3, fi196=t_fic, corr=t_corr, step=step_s), ref_img, ref_ctx, corr_ref

    c1, _, _, _ = one_config(1, 1)
    cn, ref_img, ref_ctx, corr_ref = one_config(ncpu, 5)
    base = {"value": round(len(TIMES) / cn["step"], 5), "unit": "frames/s", "cores": ncpu, "kind": "port",
            "value_1_thread": round(len(TIMES) / c1["step"], 5),
            "sample": "oracle/vfi_oracle.c (C restatement; the reference has no CPU path) at %dx%d, one (direction, t) unit + one "
                      "correlation pyramid, every call run in full, median of 5 runs (the 1-thread C=196 call: one run), at 1 thread / "
                      "%d threads: DepthFlowProjection %.3f / %.3f s (sequential scatter), FilterInterpolation C=3 %.3f / %.3f s, "
                      "C=196 %.2f / %.2f s, 5-level correlation of one direction %.3f / %.3f s; step = 6 x (proj + FI196 + FI3) + "
                      "2 x corr = %.1f / %.2f s"
                      % (wl.h, wl.w, ncpu, c1["dfp"], cn["dfp"], c1["fi3"], cn["fi3"], c1["fi196"], cn["fi196"], c1["corr"],
                         cn["corr"], c1["step"], cn["step"])}

    # parity of the GPU path on the same inputs (GPU fed the oracle's projected flow -> exact compare)
    gproj = torch.tensor(proj, device=dev)
    assert cabi.filterinterp_forward_ori(wl.ctx[0], gproj, wl.filters[0], wl.out_ctx, direct=args.direct) == 0
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
    chain_parity = chained_parity(torch, cabi, wl, dev, args, ncpu)
    parity = {"vs": "CPU oracle (fmad=1) on the same inputs",
              "chain": chain_parity,
              "filterinterp_c3_max_abs_err": err3,
              "filterinterp_ctx_max_abs_err": float(np.abs(wl.out_ctx.cpu().numpy() - ref_ctx).max()),
              "depthflowproj_max_abs_err": float(np.abs(gp.cpu().numpy() - proj).max()),
              "correlation_max_abs_err": float(np.abs(gcorr.cpu().numpy() - corr_ref).max()),
              "psnr_db_uint8_frame": 99.0 if psnr == float("inf") else round(psnr, 2)}
    return base, parity


def chained_parity(torch, cabi, wl, dev, args, ncpu):
    """SURVEY 8d harness-level parity: ONE interpolated frame (t = 0.5) made by the GPU chain only -- FlowProject of both
    directions, FilterInterpolate + blend, crop / x255 / round to uint8 -- against the same chain on the oracle only
    (oracle/chain.py), same inputs; PSNR of the uint8 frames as demo_MiddleBury.py:370-378, thresholds 60 dB (fp32) and 45 dB
    (fp16 storage).  `window_origins_moved` = pixels where the two projected flows put int(x + fx) on different sides of an
    integer: the only place a last-bit difference of the projection becomes a whole-pixel difference of the warp."""
    import numpy as np
    from oracle import chain
    from vfidkr_amd import fused
    if (wl.h, wl.w) != (args.height + sum(fused.padding_for(args.height, args.width)[2:]), args.width + sum(fused.padding_for(args.height, args.width)[:2])):
        return None
    left, right, top, bottom = fused.padding_for(args.height, args.width)
    ti = 1
    t = TIMES[ti]
    flows = [wl.flows[0][ti], wl.flows[1][ti]]
    ref = chain.unit([f.cpu().numpy() for f in wl.frames], [f.cpu().numpy() for f in flows], [d.cpu().numpy() for d in wl.depth],
                     [k.cpu().numpy() for k in wl.filters], t, args.height, args.width, left, top, nthreads=ncpu)
    proj = fused.FlowProject_directions([[flows[0]], [flows[1]]], wl.depth)
    p0, p2 = proj[0][0], proj[1][0]
    blend, _, _ = fused.FilterInterpolate(wl.frames[0], wl.frames[1], [p0, p2], wl.filters, 16, t)
    u8 = fused.padded_to_frames(blend, args.height, args.width, (left, right, top, bottom))
    o0 = torch.empty(wl.frames[0].shape, device=dev, dtype=torch.float16)
    o2 = torch.empty_like(o0)
    assert cabi.filterinterp_forward_ori_f16(wl.frames[0].half(), p0, wl.filters[0], o0) == 0
    assert cabi.filterinterp_forward_ori_f16(wl.frames[1].half(), p2, wl.filters[1], o2) == 0
    blend16 = (o0.float() * (1.0 - t) + o2.float() * t).half().float().contiguous()
    u16 = fused.padded_to_frames(blend16, args.height, args.width, (left, right, top, bottom))
    torch.cuda.synchronize(dev)
    g8, g16 = u8.cpu().numpy(), u16.cpu().numpy()
    moved = sum(chain.int_flips(p.cpu().numpy(), ref["proj"][d]) for d, p in enumerate((p0, p2)))
    return {"what": "GPU-only chain vs oracle-only chain, one frame at t = %.2f: DepthFlowProjection x 2 -> FilterInterpolation x 2 -> blend -> uint8" % t,
            "psnr_db_fp32": round(chain.psnr_u8(g8, ref["u8"]), 2), "psnr_db_fp16_storage": round(chain.psnr_u8(g16, ref["u8"]), 2),
            "thresholds_db": {"fp32": 60.0, "fp16_storage": 45.0},
            "uint8_values_differing": int(np.count_nonzero(g8 != ref["u8"])), "uint8_values": int(ref["u8"].size),
            "window_origins_moved": int(moved),
            "projection_max_abs_diff": float(max(np.abs(p.cpu().numpy() - ref["proj"][d]).max() for d, p in enumerate((p0, p2))))}


# ------------------------------------------------------------------------------------------- launcher

def main(argv=None):
    args = parse(argv)
    from vfidkr_amd import runner
    if args.gpus > 1 and not runner.launched_externally():
        # start the ranks ourselves; the parent never touches the GPU runtime (not even to count devices: on ROCm that can
        # open HIP) -- every rank validates its own device and exits 2 with a message, which spawn_ranks passes on
        child = [os.path.abspath(__file__)] + (list(argv) if argv is not None else sys.argv[1:])
        return runner.spawn_ranks(child, args.gpus, extra_env={"VFI_BENCH_OWN_RANKS": "1"})
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
