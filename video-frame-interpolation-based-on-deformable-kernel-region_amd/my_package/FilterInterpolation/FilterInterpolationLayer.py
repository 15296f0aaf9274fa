"""Mirror of my_package/FilterInterpolation/FilterInterpolationLayer.py:10-92 (reference).

The shipped reference calls the `_ori` kernels (FilterInterpolationLayer.py:35, 73);
the three deformable-kernel variants it keeps commented out (:36-38, :74-76) are
exposed here as separate Functions so they can be used and tested."""
import torch
from torch.autograd import Function

import filterinterpolation_cuda as my_lib

from .._common import check, require_gpu


class FilterInterpolationLayer(Function):
    @staticmethod
    def forward(ctx, input1, input2, input3):
        assert input1.is_contiguous()
        assert input2.is_contiguous()
        assert input3.is_contiguous()
        require_gpu(input1, input2, input3)
        # the kernel writes every element: no zero fill needed (reference: .zero_(), :34)
        output = torch.empty_like(input1)
        err = my_lib.FilterInterpolationLayer_gpu_forward_ori(input1, input2, input3, output)
        check(err, "FilterInterpolationLayer_gpu_forward_ori")
        ctx.save_for_backward(input1, input2, input3)
        return output

    @staticmethod
    def backward(ctx, gradoutput):
        input1, input2, input3 = ctx.saved_tensors
        gradoutput = gradoutput.contiguous()
        gradinput1 = torch.zeros_like(input1)
        gradinput2 = torch.zeros_like(input2)
        gradinput3 = torch.zeros_like(input3)
        err = my_lib.FilterInterpolationLayer_gpu_backward_ori(input1, input2, input3, gradoutput, gradinput1,
                                                              gradinput2, gradinput3)
        if err != 0:
            print(err)
        return gradinput1, gradinput2, gradinput3


class _Deformable(Function):
    """The deformable-kernel variants (forward and backward)."""
    variant = None

    @classmethod
    def _fwd(cls, ctx, input1, input2, input3, input4):
        require_gpu(input1, input2, input3)
        # the 4-input forward leaves the caller's zeros for filter sizes other than 4 and 6
        output = torch.zeros_like(input1)
        if cls.variant == "offset":
            err = my_lib.FilterInterpolationLayer_gpu_forward(input1, input2, input3, input4, output)
        elif cls.variant == "deforconv":
            err = my_lib.FilterInterpolationLayer_gpu_forward_deforconv(input1, input2, input3, input4, output)
        else:
            err = my_lib.FilterInterpolationLayer_gpu_forward_nofilterwithdeforconv(input1, input2, input3, output)
        if err != 0:
            print(err)
        if input4 is None:
            ctx.save_for_backward(input1, input2, input3)
        else:
            ctx.save_for_backward(input1, input2, input3, input4)
        return output

    @classmethod
    def _bwd(cls, ctx, gradoutput):
        gradoutput = gradoutput.contiguous()
        saved = ctx.saved_tensors
        grads = [torch.zeros_like(t) for t in saved]
        if cls.variant == "offset":
            err = my_lib.FilterInterpolationLayer_gpu_backward(*saved, gradoutput, *grads)
        elif cls.variant == "deforconv":
            err = my_lib.FilterInterpolationLayer_gpu_backward_deforconv(*saved, gradoutput, *grads)
        else:
            err = my_lib.FilterInterpolationLayer_gpu_backward_nofilterwithdeforconv(*saved, gradoutput, *grads)
        if err != 0:
            print(err)
        return tuple(grads)


class FilterInterpolationOffsetLayer(_Deformable):
    """FilterInterpolationLayer_gpu_forward / _backward (reference FilterInterpolationLayer.py:36, :74)."""
    variant = "offset"

    @staticmethod
    def forward(ctx, input1, input2, input3, input4):
        return FilterInterpolationOffsetLayer._fwd(ctx, input1.contiguous(), input2.contiguous(), input3.contiguous(),
                                                   input4.contiguous())

    @staticmethod
    def backward(ctx, gradoutput):
        return FilterInterpolationOffsetLayer._bwd(ctx, gradoutput)


class FilterInterpolationDeforConvLayer(_Deformable):
    """FilterInterpolationLayer_gpu_forward_deforconv / _backward_deforconv (reference :37, :75)."""
    variant = "deforconv"

    @staticmethod
    def forward(ctx, input1, input2, input3, input4):
        return FilterInterpolationDeforConvLayer._fwd(ctx, input1.contiguous(), input2.contiguous(),
                                                      input3.contiguous(), input4.contiguous())

    @staticmethod
    def backward(ctx, gradoutput):
        return FilterInterpolationDeforConvLayer._bwd(ctx, gradoutput)


class FilterInterpolationNoFilterLayer(_Deformable):
    """FilterInterpolationLayer_gpu_forward_nofilterwithdeforconv / _backward_... (reference :38, :76)."""
    variant = "nofilter"

    @staticmethod
    def forward(ctx, input1, input2, input3):
        return FilterInterpolationNoFilterLayer._fwd(ctx, input1.contiguous(), input2.contiguous(),
                                                     input3.contiguous(), None)

    @staticmethod
    def backward(ctx, gradoutput):
        return FilterInterpolationNoFilterLayer._bwd(ctx, gradoutput)
