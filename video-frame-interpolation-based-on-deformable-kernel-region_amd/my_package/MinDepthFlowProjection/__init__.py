from .minDepthFlowProjectionModule import *
