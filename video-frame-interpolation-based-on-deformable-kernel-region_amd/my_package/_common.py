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


def check(err, what):
    """The reference wrappers only print a non-zero return of the binding (FlowProjectionLayer.py:41-42) and go
    on with their zero-filled outputs.  The forwards here allocate with torch.empty (the kernels write every
    element), so after a refused call the output would be uninitialised memory: raise instead."""
    if err != 0:
        raise RuntimeError("vfidkr_amd: %s returned %d (shape / stride mismatch: the reference binding's silent `return 1`)"
                           % (what, err))
