from .InterpolationChModule import *
