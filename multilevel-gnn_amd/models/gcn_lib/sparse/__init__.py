from .torch_nn import MLP, act_layer, norm_layer  # noqa: F401
from .torch_message import GenMessagePassing, MsgNorm  # noqa: F401
from .torch_vertex import GENConv, SAGEConv, RSAGEConv, GraphConv  # noqa: F401
