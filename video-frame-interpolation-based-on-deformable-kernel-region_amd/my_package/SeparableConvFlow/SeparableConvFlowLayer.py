"""Mirror of my_package/SeparableConvFlow/SeparableConvFlowLayer.py:10-95 (reference;
legacy instance-style Function there, static-method Function here)."""
import torch
from torch.autograd import Function

import separableconvflow_cuda as my_lib

from .._common import require_gpu


class SeparableConvFlowLayer(Function):
    @staticmethod
    def forward(ctx, input1, input2, input3, filtersize):
        require_gpu(input1, input2, input3)
        filter_size = min(input2.size(1), input3.size(1))
        out_h = min(input2.size(2), input3.size(2))
        out_w = min(input2.size(3), input3.size(3))
        assert input1.size(2) - filtersize == out_h - 1
        assert input1.size(3) - filtersize == out_w - 1
        assert filter_size == filtersize
        assert input1.is_contiguous() and input2.is_contiguous() and input3.is_contiguous()
        flow_output = torch.zeros((input1.size(0), 2, out_h, out_w), dtype=torch.float32, device=input1.device)
        err = my_lib.SeparableConvFlowLayer_gpu_forward(input1, input2, input3, flow_output)
        if err != 0:
            print(err)
        ctx.save_for_backward(input1, input2, input3)
        return flow_output

    @staticmethod
    def backward(ctx, gradoutput):
        input1, input2, input3 = ctx.saved_tensors
        gradoutput = gradoutput.contiguous()
        gradinput1 = torch.zeros_like(input1)       # the image does not enter the flow: stays zero
        gradinput2 = torch.zeros_like(input2)
        gradinput3 = torch.zeros_like(input3)
        err = my_lib.SeparableConvFlowLayer_gpu_backward(input1, input2, input3, gradoutput, gradinput1, gradinput2,
                                                        gradinput3)
        if err != 0:
            print(err)
        return gradinput1, gradinput2, gradinput3, None
