from .FlowProjectionModule import *
