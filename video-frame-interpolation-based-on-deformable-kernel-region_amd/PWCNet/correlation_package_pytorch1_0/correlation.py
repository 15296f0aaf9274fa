"""Mirror of PWCNet/correlation_package_pytorch1_0/correlation.py:6-63 (reference).

The reference's CorrelationFunction is a legacy instance-style autograd Function
(removed from torch); this one keeps the constructor arguments of `Correlation`
and the call sequence into `correlation_cuda.forward/backward`."""
import torch
from torch.autograd import Function
from torch.nn.modules.module import Module

import correlation_cuda


class CorrelationFunction(Function):
    @staticmethod
    def forward(ctx, input1, input2, pad_size=3, kernel_size=3, max_displacement=20, stride1=1, stride2=2,
                corr_multiply=1):
        if not input1.is_cuda:
            raise RuntimeError("vfidkr_amd: correlation has no CPU path")
        ctx.save_for_backward(input1, input2)
        ctx.params = (pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)
        with torch.cuda.device_of(input1):
            rbot1 = input1.new()
            rbot2 = input2.new()
            output = input1.new()
            correlation_cuda.forward(input1, input2, rbot1, rbot2, output, *ctx.params)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        with torch.cuda.device_of(input1):
            rbot1 = input1.new()
            rbot2 = input2.new()
            grad_input1 = input1.new()
            grad_input2 = input2.new()
            correlation_cuda.backward(input1, input2, rbot1, rbot2, grad_output.contiguous(), grad_input1,
                                      grad_input2, *ctx.params)
        return grad_input1, grad_input2, None, None, None, None, None, None


class Correlation(Module):
    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1):
        super(Correlation, self).__init__()
        self.pad_size = pad_size
        self.kernel_size = kernel_size
        self.max_displacement = max_displacement
        self.stride1 = stride1
        self.stride2 = stride2
        self.corr_multiply = corr_multiply

    def forward(self, input1, input2):
        return CorrelationFunction.apply(input1, input2, self.pad_size, self.kernel_size, self.max_displacement,
                                         self.stride1, self.stride2, self.corr_multiply)
