"""Mirror of my_package/Interpolation/InterpolationModule.py (reference)."""
from torch.nn import Module

from .InterpolationLayer import InterpolationLayer

__all__ = ["InterpolationModule"]


class InterpolationModule(Module):
    def __init__(self):
        super(InterpolationModule, self).__init__()

    def forward(self, input1, input2):
        return InterpolationLayer.apply(input1, input2)
