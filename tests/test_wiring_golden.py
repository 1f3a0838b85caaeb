"""oracle/model.py and oracle/training.py against the reference's EXECUTED graph wiring (tests/golden/wiring_golden.*).

The fixture was produced by importing the reference's Architecture / UNet / Tiramisu / SourceEncoder / FeatureEngineering / KernelPrediction /
MultiScalePrediction / LossDifference / Training modules with tests/golden/tf_stub.py in place of `tensorflow` (tests/golden/make_wiring_golden.py)
and running Architecture.predict + Training.model_fn on seeded float64 inputs.  The stub's ops ARE oracle/tf_ops.py, so this pins no TensorFlow
arithmetic; it pins everything between the ops: which variables exist, under which names, created in which order and shared by which passes;
slices, concats, scale order, the source a predicted kernel is applied to, where standardization is inverted, how the loss terms are weighted.
A misreading shared by oracle/model.py and deepdenoiser_amd/program.py (same author) can no longer hide: the oracle is now tied to executed
reference code, and the product is tied to the oracle by the -m gpu parity tests."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import training as OT
from oracle.model import OracleArchitecture

HERE = os.path.dirname(os.path.abspath(__file__))
META = json.load(open(os.path.join(HERE, "golden", "wiring_golden.json")))
_NPZ = None


def _arrays(case):
    global _NPZ
    if _NPZ is None:
        _NPZ = np.load(os.path.join(HERE, "golden", "wiring_golden.npz"))
    pre = case + "|"
    return {k[len(pre):]: _NPZ[k] for k in _NPZ.files if k.startswith(pre)}


def _rel(a, b):
    a, b = torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)
    den = float(b.abs().max())
    return float((a - b).abs().max()) / den if den > 0 else float((a - b).abs().max())


@pytest.mark.parametrize("case", sorted(META.keys()))
def test_oracle_reproduces_the_executed_reference_graph(case):
    m, arr = META[case], _arrays(case)
    aj, tj = json.loads(m["architecture_json"]), json.loads(m["training_json"])
    feats = {k[len("feature:"):]: torch.from_numpy(v) for k, v in arr.items() if k.startswith("feature:")}
    labels = {k[len("label:"):]: torch.from_numpy(v) for k, v in arr.items() if k.startswith("label:")}
    oracle = OracleArchitecture(aj, dtype=torch.float64, seed=0)
    # the reference's variables, in the reference's creation order; the oracle must ask for exactly these names and shapes -- and create none
    for name in m["variables"]:
        oracle.vs.vars[name] = torch.from_numpy(arr["var:" + name]).clone().requires_grad_(True)
    preds = oracle.predict(feats)
    assert list(oracle.vs.vars.keys()) == m["variables"], "the oracle created a variable the reference's graph does not have"
    assert len(preds) == m["n_scales"]
    for s, d in enumerate(preds):
        assert sorted(d.keys()) == m["prediction_keys"][s]
        for k, v in d.items():
            want = arr["prediction:%d:%s" % (s, k)]
            assert tuple(v.shape) == tuple(want.shape), (case, s, k, tuple(v.shape), want.shape)
            assert _rel(v.detach(), want) < 1e-12, (case, s, k, _rel(v.detach(), want))
    loss = OT.model_loss(oracle, aj, tj, preds, labels)
    want_loss = float(arr["loss"])
    assert abs(float(loss.detach()) - want_loss) <= 1e-12 * abs(want_loss), (case, float(loss.detach()), want_loss)
    params = [oracle.vs.vars[n] for n in m["variables"]]
    grads = torch.autograd.grad(loss, params, allow_unused=True)
    used = 0
    for n, g in zip(m["variables"], grads):
        want = arr["grad:" + n]
        if g is None:
            assert not np.any(want), (case, n, "the reference's loss depends on a variable the oracle's does not")
            continue
        used += 1
        assert _rel(g, want) < 1e-9, (case, n, _rel(g, want))
    assert used >= len(params) - 2


def test_fixture_covers_the_reference_paths_it_claims():
    """What the generator ran (read from the fixture's own records): both backbones, SINGLE and COMBINED tuples, the three flag modes, kernel
    prediction on the standardized and on the raw source, 3x3 and 5x5 kernels, one to three scales, inversion before and after the scale
    composition, every LossDifference kind in use, all nine loss-term families, and the channels_first build of the same graph."""
    seen = {"core": set(), "tuple": set(), "flags": set(), "kp": set(), "ksize": set(), "scales": set(), "invert_after": set(), "loss": set(),
            "layout": set()}
    for m in META.values():
        aj, tj = json.loads(m["architecture_json"]), json.loads(m["training_json"])
        a = aj["architecture"]
        seen["core"].add(a["core_architecture"]["name"])
        seen["tuple"].add(a["source_encoder"]["feature_prediction_tuple_type"])
        seen["flags"].add(a["source_encoder"]["feature_flag_mode"])
        seen["kp"].add((a["kernel_prediction"]["use_kernel_prediction"], a["kernel_prediction"]["use_standardized_source_for_kernel_prediction"]))
        seen["ksize"].add(a["kernel_prediction"]["kernel_size"])
        seen["scales"].add(m["n_scales"])
        seen["invert_after"].add(a["multiscale_prediction"]["invert_standardization_after_multiscale_predictions"])
        seen["loss"].add(tj["loss_difference"])
        seen["layout"].add(m["data_format"])
    assert seen["core"] == {"U-Net", "Tiramisu"} and seen["tuple"] == {"SINGLE", "COMBINED"}
    assert seen["flags"] == {"NONE", "ONE_HOT_ENCODING", "EMBEDDING"}
    assert {(True, True), (True, False), (False, True)} <= seen["kp"] and {3, 5} <= seen["ksize"] and seen["scales"] == {1, 2, 3}
    assert seen["invert_after"] == {True, False} and seen["layout"] == {"channels_last", "channels_first"}
    assert {"SMAPE", "ABSOLUTE", "SQUARED", "SMOOTH_ABSOLUTE"} <= seen["loss"]
    # variable sharing as the reference's scopes produced it: 17 tuple passes bind to one set of backbone variables
    log = META["example_single_embedding"]["variable_log"]
    created = [l for l in log if l.endswith(" create")]
    reused = [l for l in log if l.endswith(" reuse")]
    assert len(created) == len(META["example_single_embedding"]["variables"]) and len(reused) > 10 * len(created)


def test_stub_variable_scopes_follow_the_tf1_rules_the_wiring_relies_on():
    """tests/golden/tf_stub.py alone (no reference needed): default layer names count per variable scope and restart when the scope is entered
    again; reuse=False on an existing variable and reuse=True on a missing one raise as TensorFlow does; AUTO_REUSE creates once."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("dd_tf_stub_selftest", os.path.join(HERE, "golden", "tf_stub.py"))
    tf = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = tf
    spec.loader.exec_module(tf)
    tf.STORE.reset(3)
    x = torch.ones(1, 4, 4, 3, dtype=torch.float64)
    with tf.variable_scope("core", reuse=False):
        y = tf.layers.conv2d(x, 5, (3, 3), padding="same", activation=tf.nn.relu)
        y = tf.layers.conv2d(y, 5, (1, 1), padding="same")
        y = tf.layers.conv2d_transpose(y, 2, (2, 2), strides=(2, 2), padding="same")
    assert list(tf.STORE.vars) == ["core/conv2d/kernel", "core/conv2d/bias", "core/conv2d_1/kernel", "core/conv2d_1/bias",
                                   "core/conv2d_transpose/kernel", "core/conv2d_transpose/bias"]
    assert tuple(tf.STORE.vars["core/conv2d_transpose/kernel"].shape) == (2, 2, 2, 5) and tuple(y.shape) == (1, 8, 8, 2)
    with tf.variable_scope("core", reuse=True):                      # second pass: the counters restart, the same variables bind in order
        tf.layers.conv2d(x, 5, (3, 3), padding="same")
        tf.layers.conv2d(torch.ones(1, 4, 4, 5, dtype=torch.float64), 5, (1, 1), padding="same")
        with pytest.raises(ValueError):
            tf.layers.conv2d(x, 5, (3, 3), padding="same")          # would be core/conv2d_2: does not exist
    assert len(tf.STORE.vars) == 6
    with tf.variable_scope("core", reuse=False):
        with pytest.raises(ValueError):
            tf.layers.conv2d(x, 5, (3, 3), padding="same")          # core/conv2d exists: TensorFlow refuses without reuse
    for _ in range(2):
        with tf.variable_scope("embedding", reuse=tf.AUTO_REUSE):
            m = tf.get_variable("feature_flags_embedding_matrix", [4, 2], trainable=True)
    assert "embedding/feature_flags_embedding_matrix" in tf.STORE.vars and tuple(m.shape) == (4, 2)
    # symmetric padding mirrors INCLUDING the edge sample
    p = tf.pad(torch.arange(9, dtype=torch.float64).reshape(1, 3, 3, 1), [[0, 0], [1, 1], [1, 1], [0, 0]], "symmetric")
    assert p[0, :, :, 0].tolist() == [[0, 0, 1, 2, 2], [0, 0, 1, 2, 2], [3, 3, 4, 5, 5], [6, 6, 7, 8, 8], [6, 6, 7, 8, 8]]
