from .FilterInterpolationModule import *
