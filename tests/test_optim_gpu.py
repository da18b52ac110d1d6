"""Fused flat-buffer optimizer step (csrc/optim.hip, mlgnn/optim.py) against torch.optim.Adam + clip_grad_norm_, the
pair the reference's loop runs (train.py:63-66,112-114), step by step."""
import copy

import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(24, 40)
        self.dead = nn.Linear(40, 40)            # never called: grad stays None (like SAGEConv.lin_l in the reference)
        self.b = nn.Linear(40, 7, bias=False)
        self.scale = nn.Parameter(torch.tensor([0.7]))

    def forward(self, x):
        return self.b(torch.relu(self.a(x))) * self.scale


@pytest.mark.parametrize("wd,clip", [(0.0, None), (1e-2, None), (0.0, 0.05), (5e-3, 20.0)])
def test_flat_adam_matches_torch_adam_over_20_steps(wd, clip):
    from mlgnn.optim import FlatAdam
    torch.manual_seed(0)
    ref = _Net().to(DEV)
    mine = copy.deepcopy(ref)
    topt = torch.optim.Adam(ref.parameters(), lr=3e-3, betas=(0.9, 0.999), weight_decay=wd)
    fopt = FlatAdam(mine, lr=3e-3, betas=(0.9, 0.999), weight_decay=wd, clip_grad_norm=clip)
    dead0 = mine.dead.weight.detach().clone()
    gen = torch.Generator().manual_seed(1)
    for step in range(20):
        x = torch.randn(32, 24, generator=gen).to(DEV)
        y = torch.randn(32, 7, generator=gen).to(DEV)
        topt.zero_grad(set_to_none=True)
        ((ref(x) - y) ** 2).mean().mul(50.0).backward()
        norm_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), clip) if clip else None
        topt.step()
        fopt.zero_grad()
        ((mine(x) - y) ** 2).mean().mul(50.0).backward()
        fopt.bucket.collect()
        fopt.step()
        if clip:
            assert abs(float(fopt.grad_norm) - float(norm_ref)) <= 1e-5 * float(norm_ref)
    for (n, p), q in zip(ref.named_parameters(), mine.parameters()):
        err = float((p - q).abs().max())
        assert err <= 1e-6 * max(1.0, float(p.abs().max())), (n, err)
    # the parameter no backward reaches is untouched, weight decay or not (torch skips grad None)
    assert torch.equal(mine.dead.weight, dead0) and torch.equal(ref.dead.weight, dead0)
    assert fopt.bucket.check_views()
    lo, hi = fopt.flat_p.data_ptr(), fopt.flat_p.data_ptr() + 4 * fopt.flat_p.numel()
    assert all(lo <= v.data_ptr() < hi for v in mine.state_dict().values())     # parameters are views of the flat buffer


def test_clip_writes_back_scaled_gradients():
    from mlgnn.optim import FlatAdam
    torch.manual_seed(2)
    net = _Net().to(DEV)
    opt = FlatAdam(net, lr=1e-3, clip_grad_norm=0.01)
    x = torch.randn(8, 24, device=DEV)
    opt.zero_grad()
    net(x).sum().backward()
    opt.bucket.collect()
    raw = opt.bucket.flat.clone()
    opt.step()
    norm = float(raw.norm())
    assert abs(float(opt.grad_norm) - norm) <= 1e-5 * norm
    want = raw * min(1.0, 0.01 / (norm + 1e-6))
    assert float((opt.bucket.flat - want).abs().max()) <= 1e-6 * float(want.abs().max())


class _Interleaved(nn.Module):
    """100 small layers, every other one never called: 100 separate live parameter ranges (a table of 64 ranges was the
    limit of the first version of the kernel)."""

    def __init__(self):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(8, 8) for _ in range(200)])

    def forward(self, x):
        # (a sum of shallow branches: a 100-deep chain makes the clipped gradients sit at Adam's eps, where two fp32
        #  implementations of the same step legitimately part ways)
        return sum(torch.tanh(l(x)) for l in self.layers[::2]) / 10.0


def test_many_interleaved_unreached_parameters():
    from mlgnn.optim import FlatAdam
    torch.manual_seed(4)
    ref = _Interleaved().to(DEV)
    mine = copy.deepcopy(ref)
    topt = torch.optim.Adam(ref.parameters(), lr=1e-2, weight_decay=1e-2)
    fopt = FlatAdam(mine, lr=1e-2, weight_decay=1e-2, clip_grad_norm=0.05)
    x = torch.randn(16, 8, device=DEV)
    for _ in range(5):
        topt.zero_grad(set_to_none=True)
        ref(x).pow(2).mean().backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.05)
        topt.step()
        fopt.zero_grad()
        mine(x).pow(2).mean().backward()
        fopt.bucket.collect()
        fopt.step()
    assert fopt.bucket.reached == [i % 4 < 2 for i in range(400)]
    for (n, p), q in zip(ref.named_parameters(), mine.parameters()):
        assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(p.abs().max())), n


def test_nan_gradient_poisons_the_step_like_clip_grad_norm():
    """``clip_grad_norm_`` with a NaN total norm multiplies every gradient by NaN: the failure is visible in every
    reached parameter after the step (``fminf(NaN, 1) = 1`` would have hidden it)."""
    from mlgnn.optim import FlatAdam
    torch.manual_seed(5)
    ref = _Net().to(DEV)
    mine = copy.deepcopy(ref)
    topt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    fopt = FlatAdam(mine, lr=1e-3, clip_grad_norm=20.0)
    x = torch.randn(8, 24, device=DEV)
    x[3, 5] = float("nan")
    ref(x).sum().backward()
    torch.nn.utils.clip_grad_norm_(ref.parameters(), 20.0)
    topt.step()
    fopt.zero_grad()
    mine(x).sum().backward()
    fopt.bucket.collect()
    fopt.step()
    for (n, p), q in zip(ref.named_parameters(), mine.parameters()):
        assert torch.equal(torch.isnan(p), torch.isnan(q)), n
    assert bool(torch.isnan(mine.a.weight).all()) and not bool(torch.isnan(mine.dead.weight).any())
