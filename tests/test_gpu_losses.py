"""HIP loss / optimiser kernels vs the oracle and the reference-generated golden vectors (G2, G3)."""
import os

import numpy as np
import pytest
import torch

from oracle import losses as olosses

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def test_kl_and_ce_golden(golden_dir):
    from mopa_amd.common.utils.loss import seg_ce, xm_kl
    g = _load(golden_dir, "g3_kl_ce.npz")
    a = torch.from_numpy(g["a"]).cuda().requires_grad_(True)
    b, lab, w = torch.from_numpy(g["b"]).cuda(), torch.from_numpy(g["label"]).cuda(), torch.from_numpy(g["weight"]).cuda()
    kl = xm_kl(a, b)
    np.testing.assert_allclose(kl.item(), float(g["kl"]), rtol=2e-6)
    ga, = torch.autograd.grad(kl * 3.0, a)
    np.testing.assert_allclose(ga.cpu().numpy() / 3.0, g["grad_kl"], rtol=1e-4, atol=1e-8)
    ce = seg_ce(a, lab, w)
    np.testing.assert_allclose(ce.item(), float(g["ce"]), rtol=2e-6)
    ga, = torch.autograd.grad(ce, a)
    np.testing.assert_allclose(ga.cpu().numpy(), g["grad_ce"], rtol=1e-4, atol=1e-8)
    ce2 = seg_ce(a, lab, None)
    np.testing.assert_allclose(ce2.item(), float(g["ce_noweight"]), rtol=2e-6)
    ga, = torch.autograd.grad(ce2, a)
    np.testing.assert_allclose(ga.cpu().numpy(), g["grad_ce_noweight"], rtol=1e-4, atol=1e-8)
    # int32 pseudo labels (train_xmuda_mopa.py:458 .long()) and an all-ignored batch -> nan like torch
    ce3 = seg_ce(a, lab.int(), w)
    assert ce3.item() == ce.item()
    assert torch.isnan(seg_ce(a, torch.full_like(lab, -100), w))


def test_kl_ce_large_random_vs_oracle():
    from mopa_amd.common.utils.loss import seg_ce, xm_kl
    rng = np.random.Generator(np.random.PCG64(1))
    N, C = 279_040, 10
    a = torch.from_numpy(rng.standard_normal((N, C), dtype=np.float32) * 4)
    b = torch.from_numpy(rng.standard_normal((N, C), dtype=np.float32) * 4)
    lab = torch.from_numpy(rng.integers(0, C, N))
    lab[rng.random(N) < 0.3] = -100
    w = torch.from_numpy(rng.uniform(1, 3, C).astype(np.float32))
    ad = a.cuda().requires_grad_(True)
    ar = a.double().requires_grad_(True)
    kl, klr = xm_kl(ad, b.cuda()), olosses.xm_kl(ar, b.double())
    ce, cer = seg_ce(ad, lab.cuda(), w.cuda()), olosses.seg_ce(ar, lab, w.double())
    np.testing.assert_allclose(kl.item(), klr.item(), rtol=1e-5)
    np.testing.assert_allclose(ce.item(), cer.item(), rtol=1e-5)
    (kl + ce).backward()
    (klr + cer).backward()
    np.testing.assert_allclose(ad.grad.cpu().numpy(), ar.grad.float().numpy(), rtol=1e-4, atol=1e-10)


def test_mask_cons_golden_and_softmax(golden_dir):
    from mopa_amd.common.utils.loss import mask_cons_loss, softmax_lastdim
    g = _load(golden_dir, "g2_mask_cons.npz")
    logits = torch.from_numpy(g["logits"]).cuda().requires_grad_(True)
    masks = [torch.from_numpy(m) for m in g["masks"]]  # CPU int32 tensors, as collate delivers them
    loss = mask_cons_loss(softmax_lastdim(logits), [m.cuda() for m in masks], True)
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=2e-5)
    loss.backward()
    np.testing.assert_allclose(logits.grad.cpu().numpy(), g["grad_logits"], rtol=2e-3, atol=2e-8)
    l2 = mask_cons_loss(softmax_lastdim(logits.detach()), masks, False)
    np.testing.assert_allclose(l2.item(), float(g["loss_noent"]), rtol=2e-5)
    assert mask_cons_loss(softmax_lastdim(logits.detach()), [], True) == 0


def test_mask_cons_full_size_vs_oracle():
    from mopa_amd import synth
    from mopa_amd.common.utils.loss import mask_cons_loss, softmax_lastdim
    rng = np.random.Generator(np.random.PCG64(3))
    B, H, W, C = 2, 302, 480, 5
    logits = torch.from_numpy(rng.standard_normal((B, H, W, C), dtype=np.float32))
    masks = [torch.from_numpy(synth.sam_mask(rng, H, W)) for _ in range(B)]
    ld = logits.cuda().requires_grad_(True)
    lr = logits.double().requires_grad_(True)
    loss = mask_cons_loss(softmax_lastdim(ld), masks, True)
    ref = olosses.mask_cons_loss(torch.softmax(lr, 3), masks, True)
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-5)
    loss.backward()
    ref.backward()
    np.testing.assert_allclose(ld.grad.cpu().numpy(), lr.grad.float().numpy(), rtol=1e-3, atol=1e-10)


def test_mask_cons_loss_is_bit_reproducible():
    """The per-mask sums are ordered (wave shuffles + wave-private LDS accumulators, csrc/losses.hip::k_mc_pass): forward and
    backward of mask_cons_loss (loss.py:241-283) at 8 x 302 x 480 give the same bits run to run -- and with random, spatially
    INCOHERENT ids (the leader loop's worst case: up to 64 distinct ids per wave) and 11 classes as well."""
    from mopa_amd import synth
    from mopa_amd.common.utils.loss import mask_cons_loss, softmax_lastdim
    rng = np.random.Generator(np.random.PCG64(11))
    for B, C, coherent in ((8, 5, True), (2, 11, False)):
        H, W = 302, 480
        logits = torch.from_numpy(rng.standard_normal((B, H, W, C), dtype=np.float32)).cuda()
        if coherent:
            masks = [torch.from_numpy(synth.sam_mask(rng, H, W)).cuda() for _ in range(B)]
        else:
            masks = [torch.from_numpy(rng.integers(-1, 256, (H, W)).astype(np.int32)).cuda() for _ in range(B)]
        runs = []
        for _ in range(3):
            ld = logits.clone().requires_grad_(True)
            loss = mask_cons_loss(softmax_lastdim(ld), masks, True)
            loss.backward()
            runs.append((loss.detach().clone(), ld.grad.clone()))
        for l, g in runs[1:]:
            assert torch.equal(l, runs[0][0]) and torch.equal(g, runs[0][1])
        # and the value is the oracle's (the incoherent case has no golden vector)
        ref = olosses.mask_cons_loss(torch.softmax(logits.cpu().double(), 3), [m.cpu() for m in masks], True)
        np.testing.assert_allclose(runs[0][0].item(), ref.item(), rtol=1e-5)


def test_flat_adam_matches_torch_adam():
    from mopa_amd.optim import FlatAdam
    torch.manual_seed(0)
    ps = [torch.randn(s, device="cuda").requires_grad_(True) for s in ((7, 3), (130,), (5, 5, 5), (1,))]
    qs = [p.detach().clone().requires_grad_(True) for p in ps]
    opt, ref = FlatAdam(ps, lr=1e-2), torch.optim.Adam(qs, lr=1e-2)
    for it in range(5):
        opt.zero_grad()
        ref.zero_grad()
        for p, q in zip(ps, qs):
            (p.sin() * (it + 1)).sum().backward()
            (q.sin() * (it + 1)).sum().backward()
        opt.step()
        ref.step()
    for p, q in zip(ps, qs):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
