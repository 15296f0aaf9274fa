"""Working counterpart of my_package/SeparableConv/SeparableConvLayer.py (reference; its
own copy is stale: it imports `_ext.my_lib`, :4, and cannot be imported).  Native side:
separableconv_cuda.cc:10-174."""
import torch
from torch.autograd import Function

import separableconv_cuda as my_lib

from .._common import require_gpu


class SeparableConvLayer(Function):
    @staticmethod
    def forward(ctx, input1, input2, input3, filtersize):
        assert input1.is_contiguous() and input2.is_contiguous() and input3.is_contiguous()
        require_gpu(input1, input2, input3)
        assert min(input2.size(1), input3.size(1)) == filtersize
        out_h = min(input2.size(2), input3.size(2))
        out_w = min(input2.size(3), input3.size(3))
        assert input1.size(2) - filtersize == out_h - 1
        assert input1.size(3) - filtersize == out_w - 1
        output = torch.zeros((input1.size(0), input1.size(1), out_h, out_w), dtype=torch.float32,
                             device=input1.device)
        err = my_lib.SeparableConvLayer_gpu_forward(input1, input2, input3, output)
        if err != 0:
            print(err)
        ctx.save_for_backward(input1, input2, input3)
        return output

    @staticmethod
    def backward(ctx, gradoutput):
        input1, input2, input3 = ctx.saved_tensors
        gradoutput = gradoutput.contiguous()
        gradinput1 = torch.zeros_like(input1)
        gradinput2 = torch.zeros_like(input2)
        gradinput3 = torch.zeros_like(input3)
        err = my_lib.SeparableConvLayer_gpu_backward(input1, input2, input3, gradoutput, gradinput1, gradinput2,
                                                    gradinput3)
        if err != 0:
            print(err)
        return gradinput1, gradinput2, gradinput3, None
