"""Mirror of the `sparseconvnet` package surface used by the reference's 3-D detection path
(SparseConvNet/sparseconvnet/__init__.py), backed by libd3d_hip.so."""
from . import SCN
from .modules import (AddTable, BatchNormalization, BatchNormLeakyReLU, BatchNormReLU, ConcatTable,
                      Convolution, Deconvolution, Identity, InputLayer, Metadata, OutputLayer,
                      Sequential, SparseConvNetTensor, SparseToDense, SubmanifoldConvolution,
                      add_feature_planes, toLongTensor)
from .fpn_net import FPN_Net

forward_pass_multiplyAdd_count = 0
forward_pass_hidden_states = 0
