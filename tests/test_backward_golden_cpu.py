"""CPU: tests/golden/golden_backward.npz is self-consistent -- the loss the reference's training_step returned is the sum of its own
logged terms under the stored weights (so the composition the GPU test rebuilds, tests/test_gpu_backward_golden.trainer_loss, is the
reference's), every parameter group has a finite gradient, and the table summaries agree with their own 1-in-32 subset."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ALIAS = {"train/loss_rgb": "lambda_rgb_l1", "train/loss_rgb_phys": "lambda_rgb_phys_l1"}


VARIANTS = ("default", "allterms", "lipshitz", "uniform_default")      # uniform_default: render_mode uniform_light, spp 512 (the shipped training estimator)


def _B():
    return np.load(os.path.join(HERE, "golden", "golden_backward.npz"))


def test_loss_is_the_weighted_sum_of_the_logged_terms():
    B = _B()
    for name in VARIANTS:
        lam = {k: float(v) for k, v in (str(s).split("=", 1) for s in B[f"{name}_lambdas"])}
        terms = {k: float(v) for k, v in (str(s).split("=", 1) for s in B[f"{name}_loss_terms"]) if k.startswith("train/loss_")}
        total = 0.0
        for k, v in terms.items():
            lk = ALIAS.get(k, "lambda_" + k[len("train/loss_"):])
            total += lam.get(lk, 0.0) * v
        assert abs(total - float(B[f"{name}_loss"])) <= 2e-6 * abs(total), (name, total, float(B[f"{name}_loss"]))


def test_every_group_has_a_finite_gradient_and_the_summaries_are_consistent():
    B = _B()
    for name in VARIANTS:
        names = [str(s) for s in B[f"{name}_grad_names"]]
        assert len(names) == 25
        for p in names:
            if p.endswith("encoding.encoding.params"):
                pre = f"{name}_grad_{p}:"
                l1, l2, nnz = B[pre + "level_l1"], B[pre + "level_l2"], B[pre + "level_nnz"]
                assert np.all(l2 <= l1 + 1e-12) and np.all(l1 <= np.sqrt(np.maximum(nnz, 1)) * l2 + 1e-9)
                assert np.all(np.abs(B[pre + "level_sum"]) <= l1 + 1e-12)
                sub = B[pre + "sub_value"]
                assert sub.size > 10000 and np.isfinite(sub).all() and np.all(sub != 0)
                # the subset is ~ 1/32 of the touched entries
                assert abs(sub.size * 32 / nnz.sum() - 1.0) < 0.05
            else:
                assert np.isfinite(B[f"{name}_grad_{p}"]).all(), p
    bars = json.load(open(os.path.join(HERE, "golden", "grad_parity_bars.json")))
    assert len(bars["groups"]) >= 3 * 19 and len(bars["tables"]) >= 6


def test_the_uniform_light_variant_is_the_shipped_training_configuration():
    B = _B()
    assert str(B["uniform_default_run"]) == "uniform_light_512_gi_train" and str(B["default_run"]) == "light_16_gi_train"
    lam_u = sorted(str(s) for s in B["uniform_default_lambdas"])
    assert lam_u == sorted(str(s) for s in B["default_lambdas"])          # the default loss composition (configs/config.yaml:87-109)
    assert sorted(str(s) for s in B["uniform_default_grad_names"]) == sorted(str(s) for s in B["default_grad_names"])
