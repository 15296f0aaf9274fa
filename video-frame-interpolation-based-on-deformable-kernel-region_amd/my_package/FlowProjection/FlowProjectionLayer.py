"""Mirror of my_package/FlowProjection/FlowProjectionLayer.py:10-92 (reference)."""
import torch
from torch.autograd import Function

import flowprojection_cuda as my_lib

from .._common import check, require_gpu


class FlowProjectionLayer(Function):
    @staticmethod
    def forward(ctx, input1, requires_grad):
        assert input1.is_contiguous()
        require_gpu(input1)
        fillhole = 1 if requires_grad == False else 0     # noqa: E712  (reference :23)
        # the reference zero-fills both (:35-36) because its kernels accumulate into them; this
        # library writes every element, so plain allocations do
        count = torch.empty((input1.size(0), 1, input1.size(2), input1.size(3)), dtype=torch.float32,
                            device=input1.device)
        output = torch.empty_like(input1)
        err = my_lib.FlowProjectionLayer_gpu_forward(input1, count, output, fillhole)
        check(err, "FlowProjectionLayer_gpu_forward")
        ctx.save_for_backward(input1, count)
        ctx.fillhole = fillhole
        return output

    @staticmethod
    def backward(ctx, gradoutput):
        input1, count = ctx.saved_tensors
        gradoutput = gradoutput.contiguous()
        gradinput1 = torch.zeros_like(input1)
        err = my_lib.FlowProjectionLayer_gpu_backward(input1, count, gradoutput, gradinput1)
        if err != 0:
            print(err)
        return gradinput1, None
