"""``from elektronn2_amd import neuromancer as nm`` -- the node API of
elektronn2/neuromancer (the subset on the 3-D conv/pool/upconv hot path)."""
from .graphutils import TaggedShape, make_func, floatX, as_floatX
from .variables import VariableParam, VariableWeight, ConstantParam, initweights
from .node_basic import (Node, Input, Input_like, Concat, Add, model_manager,
                         choose_name)
from .neural import Conv, UpConv, Pool, Crop, AutoMerge, UpConvMerge, FragmentsToDense, Perceptron
from .loss import (Softmax, MultinoulliNLL, MalisNLL, AggregateLoss, Classification,
                   Errors)
from .optimiser import Optimiser, SGD, Adam
from .model import Model, modelload, params_from_model_file
from .options import set_plan_options, plan_options
