"""matten_amd: MI355X-native implementation of eigenn's tensor-field convolution
message-passing layer (mjwen/matten `eigenn/nn/point_conv.py`) and its per-edge encoders.

Host language is Python (as the reference's); the kernels are hand-written HIP for gfx950
behind the C ABI of include/tfc.h (libtfc.so, built in-tree by __graft_entry__.build()).
"""
from . import keys as DataKey  # noqa: F401
from .irreps import Irrep, Irreps  # noqa: F401
from .module_irreps import ModuleIrreps  # noqa: F401

__version__ = "0.1.0"
