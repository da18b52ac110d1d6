#!/usr/bin/env python3
"""Development tool: random shapes through the fp32 three-term paths -- ``mlgnn_linear_f32x3_*`` (ragged row counts,
widths that are multiples of 128) and ``mlgnn_diffpool_large_f32_*`` (batches, own / shared adjacency, the symmetric
shortcut, adjacency gradient) -- against fp64 on the device.  `python tools/fuzz_dense_f32.py [cases] [seed]`."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multilevel-gnn_amd"))
sys.path.insert(0, ROOT)


def main():
    import torch
    from mlgnn import dense
    from oracle import primitives as OP
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = "cuda:0"
    for i in range(n_cases):
        g = torch.Generator().manual_seed(1000 + i)
        if i % 2 == 0:
            N = rng.choice([8192, 8193, 9000, 12345, 20000, 33333])
            R, J = 128 * rng.randint(1, 5), 128 * rng.randint(1, 5)
            bias = rng.random() < 0.7
            cfg = ("linear", N, R, J, bias)
            x, w = torch.randn(N, R, generator=g).to(dev), (torch.randn(J, R, generator=g) * R ** -0.5).to(dev)
            b = torch.randn(J, generator=g).to(dev) if bias else None
            cot = torch.randn(N, J, generator=g).to(dev)
            xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
            bd = b.double().requires_grad_(True) if bias else None
            ref = torch.nn.functional.linear(xd, wd, bd)
            (ref * cot.double()).sum().backward()
            xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            bc = b.clone().requires_grad_(True) if bias else None
            y = dense._WideLinearF32.apply(xc, wc, bc)
            (y * cot).sum().backward()
            checks = [("y", y, ref.detach(), x.double().abs() @ w.double().abs().t() + (b.double().abs() if bias else 0.0)),
                      ("dx", xc.grad, xd.grad, cot.double().abs() @ w.double().abs()),
                      ("dw", wc.grad, wd.grad, cot.double().abs().t() @ x.double().abs())]
            if bias:
                checks.append(("db", bc.grad, bd.grad, cot.double().abs().sum(0)))
        else:
            N, K, C = 128 * rng.randint(1, 6), 128 * rng.randint(1, 6), 128 * rng.randint(1, 3)
            B, shared, sym, gadj = rng.randint(1, 3), rng.random() < 0.5, rng.random() < 0.4, rng.random() < 0.6
            cfg = ("diffpool", N, K, C, B, shared, sym, gadj)
            z = torch.randn(B, N, C, generator=g)
            a = torch.rand(1 if shared else B, N, N, generator=g) + torch.eye(N)
            if sym:
                a = (a + a.transpose(1, 2)) * 0.5
            s = torch.randn(B, N, K, generator=g) * 2.0
            wx, wa = torch.randn(B, K, C, generator=g).double().to(dev), (torch.randn(B, K, K, generator=g) / K).double().to(dev)
            zd, ad, sd = (t.double().to(dev).requires_grad_(True) for t in (z, a, s))
            rx, ra, rl, re = OP.dense_diff_pool(zd, ad, sd)
            ((rx * wx).sum() + (ra * wa).sum() + rl * 3e4 + re * 2.0).backward()
            zc, sc = z.to(dev).requires_grad_(True), s.to(dev).requires_grad_(True)
            ac = a.to(dev).requires_grad_(gadj)
            x_, ao, link, ent = dense.dense_diff_pool(zc, ac, sc, adj_symmetric=sym)
            ((x_ * wx.float()).sum() + (ao * wa.float()).sum() + link * 3e4 + ent * 2.0).backward()
            one = lambda t: torch.ones_like(t) * max(1.0, float(t.abs().max()))
            checks = [("x'", x_, rx.detach(), one(rx)), ("A'", ao, ra.detach(), one(ra)),
                      ("dz", zc.grad, zd.grad, one(zd.grad)), ("ds", sc.grad, sd.grad, one(sd.grad)),
                      ("link", link.reshape(1), rl.detach().reshape(1), rl.detach().abs().reshape(1)),
                      ("ent", ent.reshape(1), re.detach().reshape(1), re.detach().abs().reshape(1))]
            if gadj:
                checks.append(("dA", ac.grad, ad.grad, one(ad.grad)))
        for name, got, ref_, bound in checks:
            err = ((got.double() - ref_).abs() / (bound + 1e-300)).max()
            if not bool(err <= 1e-4):
                print("FAILED case %d %r: %s off by %.3e of its bound" % (i, cfg, name, float(err)))
                raise SystemExit(1)
        print("case %d ok %r" % (i, cfg), flush=True)
    print("all %d cases ok" % n_cases)


if __name__ == "__main__":
    main()
