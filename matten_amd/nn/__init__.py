from .edge_encoding import (  # noqa: F401
    BesselBasis,
    PolynomialCutoff,
    RadialBasisEdgeEncoding,
    ShiftedSoftPlus,
    SphericalHarmonicEdgeAttrs,
    with_batch,
    with_edge_vectors,
)
from .point_conv import PointConv, PointConvMessage, PointConvMessagePassing, PointConvUpdate  # noqa: F401
from .utils import ActivationLayer, UVUTensorProduct, scatter_add, tp_path_exists  # noqa: F401
from .embedding import SpeciesEmbedding  # noqa: F401
from .nodewise import NodewiseLinear, NodewiseReduce, NodewiseSelect  # noqa: F401
from .readout import CartesianTensor, IrrepsToCartesianTensor  # noqa: F401
from .sequential import Sequential, create_sequential_module  # noqa: F401
