from .DepthFlowProjectionModule import *
