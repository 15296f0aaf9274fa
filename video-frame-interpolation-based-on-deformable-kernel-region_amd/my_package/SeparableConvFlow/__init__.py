from .SeparableConvFlowModule import *
