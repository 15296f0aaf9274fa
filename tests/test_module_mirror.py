"""Counterpart of the reference's only test artefact, `my_package/test_module.py` (a script that ran every module once
on CPU tensors and once on CUDA tensors, compared outputs and gradients against 1e-6 and printed both times; it no
longer runs there: its imports are commented out and the CPU entry points are gone -- SURVEY.md section 4).

Same structure here, with the CPU oracle in the role of the CPU run: for every module, the canonical shape of that
script (`B, C, H, W = 1, 2, 512, 704` for the flow field, `test_module.py:1007`), its input distributions (flow
~ U(-1, 1) `:1018`, depth weight ~ U(0.1, 1) `:1019`, images and filters ~ U(0, 1) `:917-919`), forward, then
`output.backward(output.data)` (`:28`), outputs and gradients compared -- bit for bit where the op is deterministic,
1e-6 relative where only the summation order differs (the script's threshold), 1e-4 for the atomics-summed image
gradients -- and the two times printed.
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
B, H, W = 1, 512, 704


def _gpu(torch, a, grad=True):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0").requires_grad_(grad)


def _run(torch, name, module, inputs, oracle_fwd, oracle_bwd, exact_out=True, grad_tol=()):
    t0 = time.time()
    ref = oracle_fwd(*inputs)
    t1 = time.time()
    ref_grads = oracle_bwd(*inputs, ref)
    t2 = time.time()
    print("%s: CPU Forward and backward time is : %.4fs\t%.4fs" % (name, t1 - t0, t2 - t1))
    gin = [_gpu(torch, a) for a in inputs]
    torch.cuda.synchronize()
    t0 = time.time()
    out = module(*gin)
    torch.cuda.synchronize()
    t1 = time.time()
    out.backward(out.data)
    torch.cuda.synchronize()
    t2 = time.time()
    print("%s: GPU Forward and backward time is : %.4fs\t%.4fs" % (name, t1 - t0, t2 - t1))
    got = out.detach().cpu().numpy()
    if exact_out:
        assert np.array_equal(got, ref), name
    else:
        assert np.abs(got - ref).max() <= 1e-4, name
    for k, (g, r) in enumerate(zip(gin, ref_grads)):
        if r is None:
            continue
        gg = g.grad.cpu().numpy()
        tol = grad_tol[k] if k < len(grad_tol) else 0.0
        if tol == 0.0:
            assert np.array_equal(gg, r), (name, "gradient", k)
        else:
            assert np.abs(gg - r).max() <= tol * max(1.0, float(np.abs(r).max())), (name, "gradient", k, float(np.abs(gg - r).max()))


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("`-m gpu` tests need a GPU: torch.cuda.is_available() is False")
    import vfidkr_amd  # noqa: F401  (puts the reference-named extension modules on sys.path)
    return torch


@pytest.fixture(scope="module")
def data():
    rng = np.random.default_rng(1007)
    return {
        "img": rng.random((B, 3, H, W), dtype=f32),
        "flow": rng.uniform(-1, 1, (B, 2, H, W)).astype(f32),
        "filt": rng.random((B, 16, H, W), dtype=f32),
        "depth": rng.uniform(0.1, 1.0, (B, 1, H, W)).astype(f32),
    }


def test_FilterInterpolation(torch_mod, oracle, data):
    from vfidkr_amd.my_package.FilterInterpolation import FilterInterpolationModule
    _run(torch_mod, "FilterInterpolation", FilterInterpolationModule(), (data["img"], data["flow"], data["filt"]),
         lambda i, f, k: oracle.filterinterp_ori_fwd(i, f, k, fmad=1),
         lambda i, f, k, g: oracle.filterinterp_ori_bwd(i, f, k, g, fmad=1), grad_tol=(1e-4, 0.0, 0.0))


def test_InterpolationModule_and_Ch(torch_mod, oracle, data):
    from vfidkr_amd.my_package.Interpolation import InterpolationModule
    from vfidkr_amd.my_package.InterpolationCh import InterpolationChModule
    for name, mod, img in (("Interpolation", InterpolationModule(), data["img"]),
                           ("InterpolationCh", InterpolationChModule(), np.concatenate([data["img"], data["img"][:, :2]], 1))):
        _run(torch_mod, name, mod, (img, data["flow"]), lambda i, f: oracle.interp_fwd(i, f, fmad=1),
             lambda i, f, g: oracle.interp_bwd(i, f, g, fmad=1), grad_tol=(1e-4, 0.0))


def test_FlowProjectionModule(torch_mod, oracle, data):
    from vfidkr_amd.my_package.FlowProjection import FlowProjectionModule
    # requires_grad=True: no hole filling, as in training (FlowProjectionLayer.py:23)
    _run(torch_mod, "FlowProjection", FlowProjectionModule(True), (data["flow"],),
         lambda f: oracle.flowproj_fwd(f, 0)[0],
         lambda f, g: (oracle.flowproj_bwd(f, oracle.flowproj_fwd(f, 0)[1], g),), exact_out=False, grad_tol=(1e-6,))


def test_DepthFlowProjectionModule(torch_mod, oracle, data):
    from vfidkr_amd.my_package.DepthFlowProjection import DepthFlowProjectionModule

    def bwd(f, d, g):
        out, count = oracle.depthflowproj_fwd(f, d, 0)
        return oracle.depthflowproj_bwd(f, d, count, out, g)
    _run(torch_mod, "DepthFlowProjection", DepthFlowProjectionModule(True), (data["flow"], data["depth"]),
         lambda f, d: oracle.depthflowproj_fwd(f, d, 0)[0], bwd, exact_out=False, grad_tol=(1e-4, 1e-4))


@pytest.mark.parametrize("fs", [5])
def test_SeparableConv_and_Flow(torch_mod, oracle, data, fs):
    from vfidkr_amd.my_package.SeparableConv import SeparableConvModule
    from vfidkr_amd.my_package.SeparableConvFlow import SeparableConvFlowModule
    rng = np.random.default_rng(fs)
    oh, ow = H - fs + 1, W - fs + 1
    v, h = rng.random((B, fs, oh, ow), dtype=f32), rng.random((B, fs, oh, ow), dtype=f32)
    _run(torch_mod, "SeparableConv", SeparableConvModule(fs), (data["img"], v, h),
         lambda i, a, b: oracle.sepconv_fwd(i, a, b, fmad=1), lambda i, a, b, g: oracle.sepconv_bwd(i, a, b, g),
         grad_tol=(1e-4, 0.0, 0.0))
    _run(torch_mod, "SeparableConvFlow", SeparableConvFlowModule(fs), (data["img"], v, h),
         lambda i, a, b: oracle.sepconvflow_fwd(a, b, H, W, fmad=1),
         lambda i, a, b, g: (None,) + tuple(oracle.sepconvflow_bwd(a, b, g, H, W, fmad=1)))
