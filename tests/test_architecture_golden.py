"""The feature-tuple contract and the recombination against outputs of the REFERENCE's own lines, executed by
tests/golden/make_architecture_golden.py in the build container (Architecture.py:367-473 `__prepare_feature_predictions`, :510-522,
FeatureFlags.py:12-48, Prediction.py:443-481).  No restatement is involved on the golden side; inputs and outputs only are committed."""
import copy
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from deepdenoiser_amd.architecture import Architecture
from oracle import tiling_ref

GOLD_DIR = os.path.join(os.path.dirname(__file__), "golden")
ARCH = json.load(open(os.path.join(GOLD_DIR, "architecture_golden.json")))
REC = json.load(open(os.path.join(GOLD_DIR, "recombine_golden.json")))


def _variant(example, changes):
    j = copy.deepcopy(example)
    for path, value in changes.items():
        node, keys = j, path.split("__")
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = value
    return j


def _record(fp):
    st, fv = fp.feature_standardization, fp.feature_variance
    return {"name": fp.name, "type": fp.feature_prediction_type, "load_data": fp.load_data, "is_target": fp.is_target,
            "number_of_channels": fp.number_of_channels, "invert_standardization": fp.invert_standardization,
            "standardization": {"use_log1p": st.use_log1p, "mean": st.mean, "variance": st.variance},
            "feature_variance": {"use_variance": fv.use_variance, "variance_mode": fv.variance_mode, "relative_variance": fv.relative_variance,
                                 "compute_before_standardization": fv.compute_before_standardization,
                                 "compress_to_one_channel": fv.compress_to_one_channel}}


def test_golden_covers_the_literal_example_and_combined_mode():
    labels = [c["label"] for c in ARCH["cases"]]
    assert labels[0] == "example (SINGLE)" and "COMBINED" in labels
    r0, r1 = ARCH["cases"][0]["result"], ARCH["cases"][1]["result"]
    assert len(r0["feature_prediction_tuples"]) == 17 and r0["number_of_output_channels"] == 25
    assert len(r1["feature_prediction_tuples"]) == 8 and r1["number_of_output_channels"] == 75
    assert sum(1 for f in r1["feature_predictions"] if not f["load_data"]) == 7                           # generated passes exist only in COMBINED mode


@pytest.mark.parametrize("case", ARCH["cases"], ids=[c["label"] for c in ARCH["cases"]])
def test_feature_tuples_match_the_reference_lines(case):
    pj = _variant(ARCH["example_json"], case["changes"])
    arch = Architecture(pj, device="cpu")
    want = case["result"]
    keys = ("name", "type", "load_data", "is_target", "number_of_channels", "invert_standardization", "standardization", "feature_variance")
    for got_list, want_list in ((arch.auxiliary_features, want["auxiliary_features"]), (arch.feature_predictions, want["feature_predictions"])):
        assert len(got_list) == len(want_list)
        for fp, w in zip(got_list, want_list):
            assert _record(fp) == {k: w[k] for k in keys}, (fp.name, w["name"])
    got_tuples = [{"name": t.name, "type": t.feature_prediction_tuple_type, "members": [None if f is None else f.name for f in t.feature_predictions]}
                  for t in arch.feature_prediction_tuples]
    assert got_tuples == want["feature_prediction_tuples"]
    assert arch.tuple_size == want["feature_prediction_tuple_size"]
    assert arch.number_of_output_channels == want["number_of_output_channels"]
    assert list(arch.feature_flag_names) == want["feature_flag_names"]
    assert len(arch.feature_flag_names) == want["vocabulary_size"] and len(arch.feature_flag_names) // 2 == want["embedding_dimension"]
    # the source is preserved exactly when the kernel prediction does not filter the standardized source (Architecture.py:361)
    assert all(w["preserve_source"] == (not arch.use_standardized_source_for_kernel_prediction) for w in want["feature_predictions"])
    assert all(w["number_of_sources"] == arch.number_of_sources_per_target for w in want["feature_predictions"])


def recombine_inputs(case):
    """The seeded inputs of a recombination case, rebuilt as the generator drew them; verified against the committed hash."""
    rng = np.random.default_rng(case["seed"])
    shape, scale = tuple(case["shape"]), case["scale"]
    passes = {}
    for n in case["order"]:
        v = rng.standard_normal(shape) * np.exp(scale * rng.standard_normal(shape))
        v[rng.random(shape) < 0.05] = 0.0
        passes[n] = v.astype(np.float32)
    digest = hashlib.sha256(b"".join(np.ascontiguousarray(v, dtype="<f4").tobytes() for v in passes.values())).hexdigest()
    assert digest == case["inputs_sha256"], "numpy's generator no longer reproduces the committed inputs"
    if "inputs_hex" in case:
        for n, h in case["inputs_hex"].items():
            assert np.array_equal(np.frombuffer(bytes.fromhex(h), dtype="<f4").reshape(shape), passes[n])
    return passes


@pytest.mark.parametrize("case", REC["cases"], ids=["seed%d" % c["seed"] for c in REC["cases"]])
def test_oracle_recombination_is_bit_exact_against_the_reference_lines(case):
    image = tiling_ref.recombine(recombine_inputs(case))
    assert image.dtype == np.float32
    assert hashlib.sha256(np.ascontiguousarray(image, dtype="<f4").tobytes()).hexdigest() == case["image_sha256"]
    if "image_hex" in case:
        assert np.array_equal(np.frombuffer(bytes.fromhex(case["image_hex"]), dtype="<f4").reshape(case["shape"]), image)


@pytest.mark.gpu
@pytest.mark.parametrize("case", REC["cases"], ids=["seed%d" % c["seed"] for c in REC["cases"]])
def test_dd_recombine_is_bit_exact_against_the_reference_lines(lib, case):
    """dd_recombine (through the C-ABI) reproduces the bytes Prediction.py:443-481 produced."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctypes as C
    from deepdenoiser_amd import _lib as L
    passes = recombine_inputs(case)
    dev = {n: torch.tensor(v).cuda() for n, v in passes.items()}
    h, w, _ = case["shape"]
    out = torch.zeros(h, w, 3).cuda()
    d = L.RecombineDesc()
    d.n_triples = 4
    for k, c in enumerate(("Diffuse", "Glossy", "Subsurface", "Transmission")):
        d.color[k], d.direct[k], d.indirect[k] = dev[c + " Color"].data_ptr(), dev[c + " Direct"].data_ptr(), dev[c + " Indirect"].data_ptr()
    d.n_singles = 4
    for j, n in enumerate(("Volume Direct", "Volume Indirect", "Environment", "Emission")):
        d.single[j] = dev[n].data_ptr()
    d.image = out.data_ptr()
    L.check(lib.dd_recombine(C.byref(d), h * w, None))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert hashlib.sha256(np.ascontiguousarray(got, dtype="<f4").tobytes()).hexdigest() == case["image_sha256"]
