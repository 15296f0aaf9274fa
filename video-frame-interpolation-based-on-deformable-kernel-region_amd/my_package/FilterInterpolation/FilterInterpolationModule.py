"""Mirror of my_package/FilterInterpolation/FilterInterpolationModule.py:8-17 (reference)."""
from torch.nn import Module

from .FilterInterpolationLayer import (FilterInterpolationLayer, FilterInterpolationOffsetLayer,
                                       FilterInterpolationDeforConvLayer, FilterInterpolationNoFilterLayer)

__all__ = ["FilterInterpolationModule", "FilterInterpolationDeformableModule"]


class FilterInterpolationModule(Module):
    def __init__(self):
        super(FilterInterpolationModule, self).__init__()

    def forward(self, input1, input2, input3):
        # input1: reference image, input2: flow offset, input3: kernel filter
        return FilterInterpolationLayer.apply(input1, input2, input3)


class FilterInterpolationDeformableModule(Module):
    """The deformable-kernel variants the reference compiles but leaves unwired
    (FilterInterpolationLayer.py:36-38): mode in {"offset", "deforconv", "nofilter"}."""

    def __init__(self, mode="deforconv"):
        super(FilterInterpolationDeformableModule, self).__init__()
        assert mode in ("offset", "deforconv", "nofilter")
        self.mode = mode

    def forward(self, input1, input2, input3, input4=None):
        if self.mode == "offset":
            return FilterInterpolationOffsetLayer.apply(input1, input2, input3, input4)
        if self.mode == "deforconv":
            return FilterInterpolationDeforConvLayer.apply(input1, input2, input3, input4)
        return FilterInterpolationNoFilterLayer.apply(input1, input2, input3)
