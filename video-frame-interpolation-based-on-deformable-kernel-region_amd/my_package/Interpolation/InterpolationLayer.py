"""Mirror of my_package/Interpolation/InterpolationLayer.py:10-77 (reference)."""
import torch
from torch.autograd import Function

import interpolation_cuda as my_lib

from .._common import require_gpu


class InterpolationLayer(Function):
    @staticmethod
    def forward(ctx, input1, input2):
        assert input1.is_contiguous()
        assert input2.is_contiguous()
        require_gpu(input1, input2)
        # zero fill kept: a rejected shape (return 1) must leave zeros, as in the reference
        output = torch.zeros_like(input1)
        err = my_lib.InterpolationLayer_gpu_forward(input1, input2, output)
        if err != 0:
            print(err)
        ctx.save_for_backward(input1, input2)
        return output

    @staticmethod
    def backward(ctx, gradoutput):
        input1, input2 = ctx.saved_tensors
        gradoutput = gradoutput.contiguous()
        gradinput1 = torch.zeros_like(input1)
        gradinput2 = torch.zeros_like(input2)
        err = my_lib.InterpolationLayer_gpu_backward(input1, input2, gradoutput, gradinput1, gradinput2)
        if err != 0:
            print(err)
        return gradinput1, gradinput2
