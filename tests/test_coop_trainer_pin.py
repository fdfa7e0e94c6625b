"""BASELINE configs[0] is `--trainer CoOp`: the reference's trainers/coop.py with its OWN PromptLearner / TextEncoder /
CustomCLIP (trainers/coop.py:45-80, 83-212, 215-260).  Every parity test of the ViT-B/32 CoOp case reads
`full_vitb32_coop_end.npz`, which oracle/make_golden.py produced through trainers/mvlpt.py with VPT.N_CTX = 0.  These tests
pin that the two reference trainers compute the same thing on the same inputs, so the fixture stands for configs[0]:

* `full_vitb32_coop_trainer.npz` (made by `python oracle/make_golden.py coop` = trainers/coop.py's classes on the same
  frozen weights, context vectors, images and labels) equals the MVLPT-made fixture to fp32 round-off;
* where /root/reference is present (the build container) the coop fixture is regenerated and must reproduce to 1e-5.
"""
import numpy as np
import pytest

from oracle import ref_shim
from tests.golden_util import load_npz


def test_coop_trainer_equals_mvlpt_fixture():
    a, b = load_npz("full_vitb32_coop_trainer"), load_npz("full_vitb32_coop_end")
    assert np.array_equal(a["param_ctx"], b["param_ctx"]) and np.array_equal(a["label"], b["label"])
    np.testing.assert_allclose(a["out_logits"], b["out_logits"], rtol=0, atol=1e-6 * float(np.abs(b["out_logits"]).max()))
    np.testing.assert_allclose(a["out_loss"], b["out_loss"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(a["grad_ctx"], b["grad_ctx"], rtol=0, atol=1e-6 * float(np.abs(b["grad_ctx"]).max()))


@pytest.mark.skipif(not ref_shim.reference_available(), reason="needs /root/reference (build container only)")
def test_coop_trainer_fixture_regenerates():
    from oracle import make_golden as MG
    _, cm = ref_shim.load_reference()
    d = MG.run_coop_trainer_case(cm)
    z = load_npz("full_vitb32_coop_trainer")
    for k in ("out_logits", "out_loss", "grad_ctx"):         # fp32 reduction order varies with the thread count: 1e-5 of the scale (measured 2e-6 through 12 layers)
        np.testing.assert_allclose(d[k], z[k], rtol=0, atol=1e-5 * max(float(np.abs(z[k]).max()), 1e-30),
                                   err_msg=f"{k} of trainers/coop.py no longer reproduces the committed fixture")
