"""One-pass Linear backward (csrc/linear_bwd.hip: dX, dW, db from one stage of go and x, with the LayerNorm-backward /
shifted-cotangent epilogues of the GENConv MLP, models/gcn_lib/sparse/torch_nn.py:54-75) against fp64 on the device."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _inputs(N, M, K, seed, go_scale=1.0, row_spread=False):
    g = torch.Generator(device=DEV).manual_seed(seed)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)           # noqa: E731
    go = r(N, M) * go_scale
    if row_spread:                                                     # rows over six decades
        go = go * torch.pow(10.0, -6.0 * torch.rand(N, 1, device=DEV, generator=g))
    x = r(N, K)
    w = r(M, K) * 0.1
    return go, x, w, g


def _act(gamma, xhat, beta):
    """relu(fma(xhat, gamma, beta)) as the kernels evaluate it (one rounding: the sign of a value that cancels to
    within an ulp decides the ReLU mask)."""
    return torch.relu((gamma.double() * xhat.double() + beta.double()).float())


def _bound(a_abs, b_abs):
    """3 * 2^-22 per product on the summed magnitudes, with head room for the fp32 accumulation."""
    return 2e-6 * (a_abs @ b_abs)


@pytest.mark.parametrize("N", [1, 31, 32, 33, 64, 4099, 200 * 1024 + 5])
def test_ln_epilogue_matches_fp64(N):
    from mlgnn import dense as D
    M, K = 128, 256
    go, xhat, w, g = _inputs(N, M, K, 3 + N)
    rstd = torch.rand(N, device=DEV, generator=g) + 0.5
    gamma = torch.rand(K, device=DEV, generator=g) + 0.5
    beta = torch.randn(K, device=DEV, generator=g) * 0.3
    act = _act(gamma, xhat, beta)
    out = D.linear_backward(go, w, xhat, go.abs().amax(1), act.abs().amax(1), D.LB_LN, rstd=rstd, gamma=gamma, beta=beta)
    torch.cuda.synchronize()
    go64, w64, x64, act64 = go.double(), w.double(), xhat.double(), act.double()
    dA = go64 @ w64
    gy = dA * (act64 > 0)
    gg = gy * gamma.double()
    dh = rstd.double()[:, None] * (gg - gg.mean(1, keepdim=True) - x64 * (gg * x64).mean(1, keepdim=True))
    # dh: the error of dA (bounded per element by the magnitudes) passes through an affine map of norm <= 3 rstd gamma
    tol_dA = _bound(go64.abs(), w64.abs())
    tol = 4.0 * rstd.double()[:, None] * gamma.double().abs().max() * (tol_dA + tol_dA.mean(1, keepdim=True) * (1 + x64.abs())) + 1e-30
    err = (out["dx"].double() - dh).abs()
    assert bool((err <= tol).all()), float((err / tol).max())
    gw = go64.t() @ act64
    assert bool(((out["gw"].double() - gw).abs() <= _bound(go64.abs().t(), act64.abs()) + 1e-30).all())
    assert torch.allclose(out["gb"].double(), go64.sum(0), rtol=1e-5, atol=1e-5 * float(go64.abs().sum(0).max()))
    ref_gg, ref_gb = (gy * x64).sum(0), gy.sum(0)
    scale = float((gy.abs() * (1 + x64.abs())).sum(0).max()) + 1e-30
    assert float((out["ggamma"].double() - ref_gg).abs().max()) <= 1e-5 * scale
    assert float((out["gbeta"].double() - ref_gb).abs().max()) <= 1e-5 * scale
    assert abs(float(out["parts"].max()) - float(out["dx"].abs().max())) <= 1e-6 * float(out["dx"].abs().max()) + 1e-30


@pytest.mark.parametrize("N", [1, 32, 33, 4099, 200 * 1024 + 5])
@pytest.mark.parametrize("shift", [False, True])
def test_plain_and_shift_epilogues_match_fp64(N, shift):
    from mlgnn import dense as D
    M, K = 256, 128
    go, x, w, g = _inputs(N, M, K, 11 + N, row_spread=True)
    lse = (torch.randn(N, K, device=DEV, generator=g) * 5.0) if shift else None
    out = D.linear_backward(go, w, x, go.abs().amax(1), x.abs().amax(1), D.LB_SHIFT if shift else D.LB_PLAIN, lse=lse)
    torch.cuda.synchronize()
    go64, w64, x64 = go.double(), w.double(), x.double()
    dx = go64 @ w64
    # global scaling: an absolute floor of 2^-38 of the largest products next to the per-element bound
    tol = _bound(go64.abs(), w64.abs()) + 4e-12 * float(go64.abs().max()) * float(w64.abs().sum(0).max())
    err = (out["dx"].double() - dx).abs()
    assert bool((err <= tol).all()), float((err / tol).max())
    gw = go64.t() @ x64
    tolw = _bound(go64.abs().t(), x64.abs()) + 4e-12 * N * float(go64.abs().max()) * float(x64.abs().max())
    assert bool(((out["gw"].double() - gw).abs() <= tolw).all())
    assert torch.allclose(out["gb"].double(), go64.sum(0), rtol=1e-5, atol=1e-5 * float(go64.abs().sum(0).max()))
    if shift:
        want = out["dx"].double() * torch.exp2(-lse.double())
        assert torch.allclose(out["gt"].double(), want, rtol=2e-6, atol=0.0)
        assert int(out["flag"][0]) == 0


def test_shift_flag_and_determinism():
    from mlgnn import dense as D
    N, M, K = 5000, 256, 128
    go, x, w, g = _inputs(N, M, K, 5)
    lse = torch.zeros(N, K, device=DEV)
    lse[1234, 7] = 61.0
    a = D.linear_backward(go, w, x, go.abs().amax(1), x.abs().amax(1), D.LB_SHIFT, lse=lse)
    b = D.linear_backward(go, w, x, go.abs().amax(1), x.abs().amax(1), D.LB_SHIFT, lse=lse)
    assert int(a["flag"][0]) == 1
    for k in ("dx", "gw", "gb", "gt"):
        assert torch.equal(a[k], b[k]), k


def test_chained_partial_maxima():
    """The dx_max_parts of one call scale the go operand of the next (the MLP's two Linears back to back)."""
    from mlgnn import dense as D
    N = 9000
    go, xhat, w2, g = _inputs(N, 128, 256, 21)
    rstd = torch.rand(N, device=DEV, generator=g) + 0.5
    gamma = torch.rand(256, device=DEV, generator=g) + 0.5
    beta = torch.randn(256, device=DEV, generator=g) * 0.3
    act = _act(gamma, xhat, beta)
    a = D.linear_backward(go, w2, xhat, go.abs().amax(1), act.abs().amax(1), D.LB_LN, rstd=rstd, gamma=gamma, beta=beta)
    x = torch.randn(N, 128, device=DEV, generator=g)
    w1 = torch.randn(256, 128, device=DEV, generator=g) * 0.1
    b = D.linear_backward(a["dx"], w1, x, a["parts"], x.abs().amax(1), D.LB_PLAIN, go_max_is_parts=True)
    c = D.linear_backward(a["dx"], w1, x, a["dx"].abs().amax(1), x.abs().amax(1), D.LB_PLAIN)
    assert torch.equal(b["dx"], c["dx"]) and torch.equal(b["gw"], c["gw"])


def test_repeatable_under_a_competing_stream():
    """The kernel's software pipeline waits by hand (counted s_waitcnt on LDS-DMAs and asm loads, bare barriers): a
    mis-counted wait would read an LDS stage before it has landed -- timing dependent, so it shows as outputs that
    differ between identical calls.  40 calls of each epilogue with a second stream hammering memory next to them."""
    from mlgnn import dense as D
    N = 200 * 1024 + 5
    go, xhat, w2, g = _inputs(N, 128, 256, 77)
    rstd = torch.rand(N, device=DEV, generator=g) + 0.5
    gamma = torch.rand(256, device=DEV, generator=g) + 0.5
    beta = torch.randn(256, device=DEV, generator=g) * 0.3
    act = _act(gamma, xhat, beta)
    gomax, amax = go.abs().amax(1), act.abs().amax(1)
    gh, x, w1, _ = _inputs(N, 256, 128, 78)
    lse = torch.randn(N, 128, device=DEV, generator=g) * 4.0
    ghmax, xmax = gh.abs().amax(1), x.abs().amax(1)
    side = torch.cuda.Stream()
    junk = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    ref = None
    for it in range(40):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                junk.fill_(it & 255)
        a = D.linear_backward(go, w2, xhat, gomax, amax, D.LB_LN, rstd=rstd, gamma=gamma, beta=beta)
        b = D.linear_backward(gh, w1, x, ghmax, xmax, D.LB_SHIFT, lse=lse)
        if ref is None:
            ref = (a, b)
            continue
        for k in ("dx", "gw", "gb", "ggamma", "gbeta", "parts"):
            assert torch.equal(a[k], ref[0][k]), (it, "LN", k)
        for k in ("dx", "gw", "gb", "gt", "parts"):
            assert torch.equal(b[k], ref[1][k]), (it, "SHIFT", k)
    torch.cuda.synchronize()
