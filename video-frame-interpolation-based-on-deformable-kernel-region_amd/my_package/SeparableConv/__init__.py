from .SeparableConvModule import *
