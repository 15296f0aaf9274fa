"""Mirror of my_package/DepthFlowProjection/DepthFlowProjectionModule.py:7-16 (reference)."""
from torch.nn.modules.module import Module

from .DepthFlowProjectionLayer import DepthFlowProjectionLayer

__all__ = ["DepthFlowProjectionModule"]


class DepthFlowProjectionModule(Module):
    def __init__(self, requires_grad=True):
        super(DepthFlowProjectionModule, self).__init__()
        self.requires_grad = requires_grad

    def forward(self, input1, input2):
        return DepthFlowProjectionLayer.apply(input1, input2, self.requires_grad)
