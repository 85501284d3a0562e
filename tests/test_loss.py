"""Training loss, forward (first slice; reference app/ml/training/trainer.py:48-56,130-142).

CPU: the oracle's float64 restatement against the golden scalars produced by the reference's own DiceLoss + nn.BCELoss
(tests/golden/make_golden_loss.py).  GPU (-m gpu): the HIP pass (include/vtd.h: vtd_dbloss_forward, through vtd_amd.training) against the
same goldens at 1e-6 relative, saturated probabilities included, bitwise repeatable, and -- at the full B = 32 x 640 x 640 size -- additive
over a split of the batch (the five sums of the halves add up to the sums of the whole)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import loss as oloss

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ("prob_loss", "thresh_loss", "dice_loss", "loss")


def _cases():
    spec = importlib.util.spec_from_file_location("make_golden_loss", os.path.join(HERE, "golden", "make_golden_loss.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)      # (only its seeded input recipe is used here: nothing touches /root/reference)
    return mod.cases()


@pytest.fixture(scope="module")
def golden():
    g = np.load(os.path.join(HERE, "golden", "dbloss.npz"))
    cases = _cases()
    for name, (prob, thresh, prob_t, thresh_t) in cases.items():   # the stored inputs ARE what the recipe regenerates
        if name + "/prob" in g:
            assert np.array_equal(g[name + "/prob"], prob.numpy()) and np.array_equal(g[name + "/thresh_t"], thresh_t.numpy()), name
        assert list(g[name + "/shape"]) == list(prob.shape)
    return g, cases


def test_oracle_loss_matches_the_reference_golden(golden):
    g, cases = golden
    for name, (prob, thresh, prob_t, thresh_t) in cases.items():
        got = oloss.detection_loss({"probability": prob.numpy(), "threshold": thresh.numpy()},
                                   {"probability_map": prob_t.numpy(), "threshold_map": thresh_t.numpy()})
        want = g[name + "/expected"]
        for k, w in zip(KEYS, want):
            assert abs(got[k] - w) <= 2e-6 * abs(w), (name, k, got[k], w)   # the golden is float32 arithmetic, the oracle float64
    assert oloss.bce(np.array([0.0, 1.0]), np.array([1.0, 0.0])) == 100.0    # torch's clamp


@pytest.mark.gpu
def test_hip_loss_matches_the_reference_golden(hip, golden):
    from vtd_amd import training
    g, cases = golden
    for name, (prob, thresh, prob_t, thresh_t) in cases.items():
        outputs = {"probability": prob.cuda(), "threshold": thresh.cuda()}
        targets = {"probability_map": prob_t.cuda(), "threshold_map": thresh_t.cuda()}
        got = training.detection_loss(outputs, targets)
        want = g[name + "/expected"]
        for k, w in zip(KEYS, want):
            v = float(got[k])
            assert abs(v - w) <= 1e-6 * abs(w), (name, k, v, w)
        again = training.detection_loss(outputs, targets, want_sums=True)
        assert all(float(again[k]) == float(got[k]) for k in KEYS)                      # bitwise repeatable
        d = training.DiceLoss()(outputs["probability"], targets["probability_map"])     # the reference's module surface
        assert float(d) == float(got["dice_loss"])
        ora = oloss.detection_loss({k: v.cpu().numpy() for k, v in outputs.items()}, {k: v.cpu().numpy() for k, v in targets.items()})
        assert abs(float(got["loss"]) - ora["loss"]) <= 1e-6 * ora["loss"]
    with pytest.raises(Exception):
        training.DiceLoss()(prob, prob_t)          # CPU tensors: no fallback path


@pytest.mark.gpu
def test_hip_loss_is_additive_over_a_batch_split_at_full_size(hip):
    """B = 32 maps of 640 x 640 (BASELINE's batch): the float64 sums behind the scalars are additive over any split of the batch, and the
    scalars follow from them by the reference's formulas -- a size-independent check that needs no CPU pass over 13 M elements per map."""
    from vtd_amd import training
    g = torch.Generator(device="cuda").manual_seed(7)
    shape = (32, 1, 640, 640)
    prob = torch.sigmoid(torch.randn(shape, generator=g, device="cuda") * 2)
    thresh = torch.sigmoid(torch.randn(shape, generator=g, device="cuda"))
    prob_t = (torch.rand(shape, generator=g, device="cuda") < 0.15).float()
    thresh_t = 0.3 + 0.4 * torch.rand(shape, generator=g, device="cuda")

    def run(sl):
        r = training.detection_loss({"probability": prob[sl], "threshold": thresh[sl]}, {"probability_map": prob_t[sl], "threshold_map": thresh_t[sl]},
                                    want_sums=True)
        return r, r["sums"].cpu().numpy()

    whole, s_all = run(slice(0, 32))
    _, s_a = run(slice(0, 13))
    _, s_b = run(slice(13, 32))
    assert np.allclose(s_a + s_b, s_all, rtol=1e-12, atol=0.0)
    n = prob.numel()
    sp = np.float32
    dice = sp(1) - (sp(2) * sp(s_all[2]) + sp(1e-5)) / (sp(s_all[3]) + sp(s_all[4]) + sp(1e-5))
    assert float(whole["prob_loss"]) == float(sp(s_all[0] / n)) and float(whole["thresh_loss"]) == float(sp(s_all[1] / n))
    assert float(whole["dice_loss"]) == float(dice)
    ref = torch.nn.functional.binary_cross_entropy(prob[:2].cpu(), prob_t[:2].cpu())     # torch's own kernel on a slice, as a spot check
    part, _ = run(slice(0, 2))
    assert abs(float(part["prob_loss"]) - float(ref)) <= 1e-6 * float(ref)
