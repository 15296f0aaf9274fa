from .InterpolationModule import *
