"""Pseudo-label update on the device against fixture G5 and the oracle restatement."""
import os

import numpy as np
import pytest
import torch

from oracle import pseudo as opseudo

pytestmark = pytest.mark.gpu


def test_refine_matches_reference_golden(golden_dir):
    from mopa_amd import pseudo
    g = dict(np.load(os.path.join(golden_dir, "g5_misc.npz")))
    logit = torch.from_numpy(g["logit"]).cuda()
    maxp, lab = pseudo.fuse(logit)
    ref_p, ref_l = torch.softmax(torch.from_numpy(g["logit"]), 1).max(1)
    assert torch.equal(lab.cpu(), ref_l)
    np.testing.assert_allclose(maxp.cpu().numpy(), ref_p.numpy(), rtol=2e-6)
    # bit-exact integer result on the reference's own probabilities
    out = pseudo.refine_pseudo_labels(ref_p.cuda(), ref_l.cuda(), num_classes=5)
    assert np.array_equal(out.cpu().numpy(), g["refined"])


@pytest.mark.parametrize("n,c", [(1, 5), (7, 5), (4097, 10), (279040, 5), (50000, 32)])
def test_refine_bit_exact_vs_oracle(n, c):
    from mopa_amd import pseudo
    gen = torch.Generator().manual_seed(n + c)
    prob = torch.rand(n, generator=gen) ** 0.3            # skewed towards 1: exercises the 0.9 cap
    prob[::7] = prob[0]                                    # many exact ties around the medians
    lab = torch.randint(0, c, (n,), generator=gen)
    if n > 100:
        lab[lab == 1] = 0                                  # an absent class
        lab[:3] = -100                                     # already-ignored labels pass through
    ref = opseudo.refine_pseudo_labels(prob, lab)
    out = pseudo.refine_pseudo_labels(prob.cuda(), lab.cuda(), num_classes=c)
    assert torch.equal(out.cpu(), ref)


def test_fusion_and_pseudo_labels_vs_oracle():
    from mopa_amd import pseudo
    gen = torch.Generator().manual_seed(3)
    n, c = 20000, 5
    l2, l3 = torch.randn(n, c, generator=gen) * 2, torch.randn(n, c, generator=gen) * 2
    ref = opseudo.fuse_probs(l2, l3)
    maxp, lab = pseudo.fuse(l2.cuda(), l3.cuda())
    top2 = ref.topk(2, dim=1)[0]
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5             # the argmax is only defined up to fp32 rounding at near-ties
    assert torch.equal(lab.cpu()[clear], ref.argmax(1)[clear]) and float(clear.float().mean()) > 0.99
    np.testing.assert_allclose(maxp.cpu().numpy(), ref.max(1)[0].numpy(), rtol=1e-5)
    for xm in (True, False):
        r2, r3 = opseudo.pseudo_labels(l2, l3, xm)
        o2, o3 = pseudo.pseudo_labels(l2.cuda(), l3.cuda(), xm)
        # labels agree except where a near-tie / a probability within rounding of its class threshold decides
        assert float((o2.cpu() != r2).float().mean()) < 2e-3 and float((o3.cpu() != r3).float().mean()) < 2e-3


def test_flat_ema_matches_torch_ema_rule():
    from mopa_amd.optim import FlatAdam
    from mopa_amd.pseudo import FlatEMA
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)).cuda()
    opt = FlatAdam(net.parameters())
    ema = FlatEMA(opt, decay=0.99)
    ref = opseudo.EMA([p.detach().cpu() for p in net.parameters()], 0.99)
    for it in range(5):
        with torch.no_grad():
            for p in net.parameters():
                p.add_(torch.randn_like(p) * 0.1)
        ema.update()
        ref.update([p.detach().cpu() for p in net.parameters()])
    before = [p.detach().clone() for p in net.parameters()]
    with ema.average_parameters():
        for p, s in zip(net.parameters(), ref.shadow):
            np.testing.assert_allclose(p.detach().cpu().numpy(), s.numpy(), rtol=1e-6, atol=1e-7)
    assert all(torch.equal(p.detach(), b) for p, b in zip(net.parameters(), before))
