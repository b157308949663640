"""MI355X-native GSS-GCN embedding trainer: the graph-convolution hot path of bowang-lab/gcn-drug-repurposing
(modules/model.py GSS_GNNLayer / ResidualGraphConvolutionalNetwork / GSS_loss + the train.py loop) as
hand-written HIP kernels behind a C ABI (include/gssgcn.h), with the reference's Python interface on top."""
from . import _lib
from ._lib import GssError, build, load

__all__ = ["GssError", "build", "load", "ResidualGraphConvolutionalNetwork", "GSS_GNNLayer", "GSS_loss", "GssGraph",
           "GssEngine"]


def __getattr__(name):  # torch-dependent pieces load lazily so `build()` works before anything else
    if name in ("ResidualGraphConvolutionalNetwork", "GSS_GNNLayer", "GSS_loss"):
        from . import model
        return getattr(model, name)
    if name in ("GssGraph", "DeviceCSR", "knn_descriptor_adj", "edgelist_adj"):
        from . import graph
        return getattr(graph, name)
    if name == "GssEngine":
        from .engine import GssEngine
        return GssEngine
    raise AttributeError(name)
