import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multilevel-gnn_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from types import SimpleNamespace
from _util import make_args
from models import get_model
import mlgnn.dense as D, mlgnn.norm as Nn
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(11)
N, E, H = 12000, 90000, 256
ei = torch.randint(0, N, (2, E), generator=gen)
mk = lambda dt: SimpleNamespace(x=torch.randn(N, 3, generator=torch.Generator().manual_seed(1)).to(dev).to(dt),
                                edge_index=ei.to(dev),
                                edge_attr=torch.rand(E, 1, generator=torch.Generator().manual_seed(2)).to(dev),
                                batch=(torch.arange(N) // (N // 4)).clamp(max=3).to(dev), age=torch.zeros(4, device=dev, dtype=dt),
                                pathway_node_attr=None, node_size=torch.full((4,), N // 4, device=dev))
args = make_args(num_layers=3, hidden_channels=H, dropout=0.0, conv_encode_edge=True, use_edge_attr=True,
                 use_column="w", global_edge="none", gcn_aggr="softmax", block="res+", norm="layer",
                 graph_pooling="mean", pathway_readout=None)
torch.manual_seed(0)
model = get_model("deepergcn")(args).to(dev)
print("fp32", model(mk(torch.float32)).detach().cpu())
model.to(torch.bfloat16)
print("bf16 native", model(mk(torch.bfloat16)).detach().float().cpu())
D.WGRAD_MIN_ROWS = 10 ** 9
import models.gcn_lib.sparse.torch_nn as TN, models.gcn_lib.sparse.torch_vertex as TV, models.deepergcn as DG
print("bf16 library linear", model(mk(torch.bfloat16)).detach().float().cpu())
orig = Nn.fused_supported
Nn.fused_supported = lambda x: False
print("bf16 library linear + aten LN", model(mk(torch.bfloat16)).detach().float().cpu())
D.WGRAD_MIN_ROWS = 8192
print("bf16 native linear + aten LN", model(mk(torch.bfloat16)).detach().float().cpu())
