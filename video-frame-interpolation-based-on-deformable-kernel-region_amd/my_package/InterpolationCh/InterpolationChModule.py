"""Mirror of my_package/InterpolationCh/InterpolationChModule.py (reference)."""
from torch.nn import Module

from .InterpolationChLayer import InterpolationChLayer

__all__ = ["InterpolationChModule"]


class InterpolationChModule(Module):
    def __init__(self):
        super(InterpolationChModule, self).__init__()

    def forward(self, input1, input2):
        return InterpolationChLayer.apply(input1, input2)
