"""Host side of libmlgnn.so: loader, graph container, autograd bindings, data-parallel helper.

There is no CPU fallback.  Every op in this package launches a HIP kernel through the C ABI
declared in ``include/mlgnn.h``; importing :mod:`mlgnn._lib` raises if the shared library has not
been built (``python __graft_entry__.py`` or ``python multilevel-gnn_amd/build_native.py``).
"""
from .graph import CSRGraph, as_graph  # noqa: F401
from .ops import (LowRankEdge, RankOneEdge, TableEdge, edge_type_embedding, gen_aggregate, share_edge_gradient,  # noqa: F401
                  weighted_mean_aggregate)
