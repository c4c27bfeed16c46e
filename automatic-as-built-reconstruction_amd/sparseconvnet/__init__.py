"""`sparseconvnet` operator API on MI355X -- the subset of the reference package
(SparseConvNet/sparseconvnet/__init__.py) that the detector's FPN_Net backbone instantiates,
with the same names, constructor signatures, parameter names and tensor contracts."""
forward_pass_multiplyAdd_count = 0
forward_pass_hidden_states = 0

from . import SCN  # noqa: E402
from .batchNormalization import BatchNormalization, BatchNormReLU, BatchNormLeakyReLU  # noqa: E402
from .convolution import Convolution  # noqa: E402
from .deconvolution import Deconvolution  # noqa: E402
from .identity import Identity  # noqa: E402
from .ioLayers import InputLayer, OutputLayer  # noqa: E402
from .metadata import Metadata  # noqa: E402
from .networkInNetwork import NetworkInNetwork  # noqa: E402
from .sequential import Sequential  # noqa: E402
from .sparseConvNetTensor import SparseConvNetTensor  # noqa: E402
from . import sparseToDense  # noqa: E402
from .sparseToDense import SparseToDense  # noqa: E402
from .submanifoldConvolution import SubmanifoldConvolution, ValidConvolution  # noqa: E402
from .tables import JoinTable, AddTable, ConcatTable  # noqa: E402
from .utils import add_feature_planes, concatenate_feature_planes, toLongTensor, optionalTensor, \
    optionalTensorReturn  # noqa: E402
from .fpn_net import FPN_Net, GeometryPrefetcher  # noqa: E402
