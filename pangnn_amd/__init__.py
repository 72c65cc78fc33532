"""pangnn_amd — MI355X-native implementation of panGNN's edge-weighted message-passing hot path.

Importing the package loads libpangnn_hip.so (the HIP kernels behind the C ABI in
include/pangnn_hip.h) and libpangnn_torch.so (the TORCH_LIBRARY(pangnn, ...) registration of the operators over
that ABI) and raises if either has not been built: there is no CPU / eager fallback."""
from . import _lib

_lib.load()

from . import torch_ops                                       # noqa: E402  (registers torch.ops.pangnn.*; raises if not built)
from .convolution import EdgeConv, GCNConv, MessagePassing   # noqa: E402
from .data import Batch, Data, DataLoader                    # noqa: E402
from .deferred import BCEWithLogitsLoss, DeferredLogits        # noqa: E402
from .gnn import AlternateGCN                                 # noqa: E402
from .graph import EdgeStructure, structure_of                # noqa: E402

__all__ = ["AlternateGCN", "GCNConv", "MessagePassing", "EdgeConv", "Data", "Batch", "DataLoader",
           "EdgeStructure", "structure_of", "BCEWithLogitsLoss", "DeferredLogits"]
