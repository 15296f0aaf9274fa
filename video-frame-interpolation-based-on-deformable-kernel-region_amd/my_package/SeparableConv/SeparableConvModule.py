"""Counterpart of my_package/SeparableConv/SeparableConvModule.py (reference)."""
from torch.nn import Module

from .SeparableConvLayer import SeparableConvLayer

__all__ = ["SeparableConvModule"]


class SeparableConvModule(Module):
    def __init__(self, filtersize):
        super(SeparableConvModule, self).__init__()
        self.filtersize = filtersize

    def forward(self, input1, input2, input3):
        return SeparableConvLayer.apply(input1, input2, input3, self.filtersize)
