#!/usr/bin/env python3
"""Development tool: which library-GEMM call form is fastest for the MLP / DiffPool shapes."""
import torch
import torch.nn.functional as F


def timed(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


dev = "cuda:0"
N = 640000
for (K, O) in [(128, 256), (256, 128), (3, 128)]:
    x = torch.randn(N, K, device=dev)
    W = torch.randn(O, K, device=dev) * 0.05
    b = torch.randn(O, device=dev)
    Wt = W.t().contiguous()
    go = torch.randn(N, O, device=dev)
    fl = 2.0 * N * K * O / 1e9
    r = {}
    r["F.linear(x,W,b)"] = timed(lambda: F.linear(x, W, b))
    r["F.linear(x,W)"] = timed(lambda: F.linear(x, W))
    r["x@Wt"] = timed(lambda: x @ Wt)
    r["addmm(b,x,Wt)"] = timed(lambda: torch.addmm(b, x, Wt))
    r["x@W.t() (view)"] = timed(lambda: x @ W.t())
    r["dX=go@W"] = timed(lambda: go @ W)
    r["dX=go@Wt.t()"] = timed(lambda: go @ Wt.t())
    r["dW=go.t()@x"] = timed(lambda: go.t() @ x)
    r["dWt=x.t()@go"] = timed(lambda: x.t() @ go)
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multilevel-gnn_amd"))
    from mlgnn import _lib
    n = int(_lib.lib.mlgnn_linear_wgrad_workspace_floats(N, O, K, 0))
    ws = torch.empty(n, device=dev)
    outb = torch.empty(O * K + O, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    r["mlgnn_linear_wgrad"] = timed(lambda: _lib.lib.mlgnn_linear_wgrad(go.data_ptr(), x.data_ptr(), None, None, None, None, outb.data_ptr(),
                                                                         ws.data_ptr(), n, N, O, K, 0, st))
    from mlgnn.dense import tall_matmul_nt, tall_matmul_supported
    if tall_matmul_supported(N, K, O):
        r["tallgemm fwd (x,W,b)"] = timed(lambda: tall_matmul_nt(x, W, b))
        r["tallgemm dX (go,Wt)"] = timed(lambda: tall_matmul_nt(go, Wt))
    print("K=%d O=%d  (%.1f GFLOP)" % (K, O, fl))
    for k, v in r.items():
        print("   %-18s %7.3f ms  %6.1f TF/s" % (k, v, fl / v))

Bp, n, C = 384, 146, 128
x = torch.randn(Bp, n, C, device=dev)
adj = torch.rand(n, n, device=dev)
print("DiffPool adj@x  [146,146] x [384,146,128]")
print("   matmul(adj,x)        %.3f ms" % timed(lambda: torch.matmul(adj, x)))
print("   bmm(expand)          %.3f ms" % timed(lambda: torch.bmm(adj.expand(Bp, n, n), x)))
print("   adj@x.permute->2D    %.3f ms" % timed(lambda: (adj @ x.permute(1, 0, 2).reshape(n, Bp * C)).reshape(n, Bp, C).permute(1, 0, 2)))
print("   einsum               %.3f ms" % timed(lambda: torch.einsum('ij,bjc->bic', adj, x)))
xt = x.permute(1, 0, 2).reshape(n, Bp * C).contiguous()
print("   adj@X2d (pre-laid)   %.3f ms" % timed(lambda: adj @ xt))
print("   (x^T adj^T) form     %.3f ms" % timed(lambda: torch.matmul(x.transpose(1, 2), adj.t()).transpose(1, 2)))
