"""Mirror of my_package/DepthFlowProjection/DepthFlowProjectionLayer.py:7-98 (reference)."""
import torch
from torch.autograd import Function

import depthflowprojection_cuda as my_lib

from .._common import check, require_gpu


class DepthFlowProjectionLayer(Function):
    @staticmethod
    def forward(ctx, input1, input2, requires_grad):
        assert input1.is_contiguous()
        assert input2.is_contiguous()
        require_gpu(input1, input2)
        fillhole = 1 if requires_grad == False else 0     # noqa: E712
        # (the reference zero-fills both; this library writes every element)
        count = torch.empty((input1.size(0), 1, input1.size(2), input1.size(3)), dtype=torch.float32,
                            device=input1.device)
        output = torch.empty_like(input1)
        err = my_lib.DepthFlowProjectionLayer_gpu_forward(input1, input2, count, output, fillhole)
        check(err, "DepthFlowProjectionLayer_gpu_forward")
        ctx.save_for_backward(input1, input2, count, output)
        ctx.fillhole = fillhole
        return output

    @staticmethod
    def backward(ctx, gradoutput):
        input1, input2, count, output = ctx.saved_tensors
        gradoutput = gradoutput.contiguous()
        gradinput1 = torch.zeros_like(input1)
        gradinput2 = torch.zeros_like(input2)
        err = my_lib.DepthFlowProjectionLayer_gpu_backward(input1, input2, count, output, gradoutput, gradinput1,
                                                           gradinput2)
        if err != 0:
            print(err)
        return gradinput1, gradinput2, None
