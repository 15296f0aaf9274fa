"""Mirror of my_package/SeparableConvFlow/SeparableConvFlowModule.py (reference)."""
from torch.nn import Module

from .SeparableConvFlowLayer import SeparableConvFlowLayer

__all__ = ["SeparableConvFlowModule"]


class SeparableConvFlowModule(Module):
    def __init__(self, filtersize):
        super(SeparableConvFlowModule, self).__init__()
        self.filtersize = filtersize

    def forward(self, input1, input2, input3):
        return SeparableConvFlowLayer.apply(input1, input2, input3, self.filtersize)
