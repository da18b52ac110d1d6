"""Development tool: three forward + backward GEN softmax aggregations at the bench shape, for counter collection
(tools/pmc_aggregate.sh)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multilevel-gnn_amd"))
from mlgnn import CSRGraph, RankOneEdge, gen_aggregate
dev = torch.device("cuda:0")
B, n, e, d = 64, 10000, 160000, int(os.environ.get("D", "128"))
if os.environ.get("SHAPE") == "configs4":            # BASELINE configs[4]: one graph, 200 000 nodes, 3 M edges, d = 256, bf16 storage
    B, n, e, d = 1, 200000, 3000000, 256
DT = torch.bfloat16 if os.environ.get("SHAPE") == "configs4" else torch.float32
gen = torch.Generator().manual_seed(1)
src = torch.randint(0, n, (B, e), generator=gen); dst = torch.randint(0, n, (B, e), generator=gen)
off = (torch.arange(B) * n)[:, None]
ei = torch.stack([(src + off).reshape(-1), (dst + off).reshape(-1)]).to(dev)
N = B * n
g = CSRGraph(ei, N)
x = torch.randn(N, d, device=dev).to(DT).requires_grad_(True)
w = torch.rand(ei.shape[1], device=dev); u, v = torch.randn(d, device=dev), torch.randn(d, device=dev)
for _ in range(3):
    o2 = gen_aggregate(x, g, RankOneEdge(w, u, v), aggr="softmax")
    torch.autograd.grad(o2, [x], torch.ones_like(o2))
torch.cuda.synchronize()
