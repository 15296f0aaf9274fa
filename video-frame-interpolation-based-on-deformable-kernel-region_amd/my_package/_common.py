"""Shared helpers of the wrapper mirrors."""
import torch


def require_gpu(*tensors):
    """The reference wrappers branch to `my_lib.*_cpu_forward`, which no extension
    exports (FilterInterpolationLayer.py:42); here the absence is explicit."""
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("vfidkr_amd: this op has no CPU path; move the tensors to the GPU")
        if t.dtype != torch.float32:
            raise RuntimeError("vfidkr_amd: float32 tensors expected (the reference allocates torch.cuda.FloatTensor)")
