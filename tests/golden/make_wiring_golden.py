"""Generates tests/golden/wiring_golden.npz by EXECUTING the reference's own graph-building code ("Next 3").

The integer and string contracts were already pinned by exec-ing reference lines (make_tiling_golden.py, make_architecture_golden.py); the
floating-point GRAPH -- Architecture.predict (Architecture.py:537-617), UNet.predict / Tiramisu.predict, SourceEncoder, FeatureEngineering,
KernelPrediction, MultiScalePrediction, LossDifference and the loss assembly of Training.model_fn (Training.py:607-660) with the FeatureTraining
objects Training.main builds (Training.py:979-991, 1008-1197) -- was a restatement (oracle/model.py, oracle/training.py) by the same author as
the product.  Here the reference's modules are IMPORTED from /root/reference at generation time with tests/golden/tf_stub.py standing in for
the `tensorflow` module (its ~60 ops are one-line calls into oracle/tf_ops.py / torch, float64, eager) and run on seeded inputs; the
fixture holds inputs, the variables the reference code created (names in creation order, values), its predictions per scale, its loss and the
gradient of that loss with respect to every variable (torch autograd through the reference's graph).

The stub is a stand-in for a library the image lacks: it pins NO TensorFlow arithmetic (parity of the primitive ops stays "unpinned",
SURVEY Appendix A).  What it pins is the WIRING: slice indices, concat order, variable-scope reuse and naming, scale order, which source a
kernel is applied to, standardization placement, loss weights and scale factors.  tests/test_wiring_golden.py checks oracle/model.py and
oracle/training.py against it to 1e-12.

Nothing of the reference travels: the .npz holds arrays and the JSON documents the cases were run on.  Run in the build container only:
    python tests/golden/make_wiring_golden.py
"""
import copy
import json
import os
import sys
import textwrap
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference/TensorFlow"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import tf_stub  # noqa: E402

tf = tf_stub.install()
sys.path.insert(0, REF)
import Architecture as RefArchitecture  # noqa: E402  (the reference's module, on the stub)
import Training as RefTraining  # noqa: E402
from Naming import Naming  # noqa: E402
from RenderPasses import RenderPasses  # noqa: E402
from FeatureFlags import FeatureFlags, FeatureFlagMode  # noqa: E402

from deepdenoiser_amd import configs  # noqa: E402  (JSON builders only; the same documents the tests use)


def ref_block(fname, first_marker, last_marker):
    """The lines of Training.main from the one containing first_marker through the one containing last_marker, dedented and compiled."""
    with open(os.path.join(REF, fname)) as f:
        lines = f.read().split("\n")
    i0 = next(i for i, l in enumerate(lines) if first_marker in l)
    i1 = next(i for i, l in enumerate(lines) if last_marker in l and i > i0)
    return compile(textwrap.dedent("\n".join(lines[i0:i1 + 1])), "%s:%d-%d" % (fname, i0 + 1, i1 + 1), "exec")


SETTINGS = ref_block("Training.py", "loss_difference = parsed_json['loss_difference']", "features_training_settings = parsed_json['features_training_settings']")
TRAININGS = ref_block("Training.py", "loss_weights = features_training_settings['loss_weights']", "0., 0.)")


def build_trainings(architecture, parsed_architecture_json, parsed_json):
    """Training.main's construction of feature_trainings / combined_feature_trainings / combined_image_feature_training, executed."""
    ns = dict(vars(RefTraining))
    ns.update({"architecture": architecture, "parsed_architecture_json": parsed_architecture_json, "parsed_json": parsed_json})
    exec(SETTINGS, ns)
    exec(TRAININGS, ns)
    return {k: ns[k] for k in ("use_multiscale_loss", "use_multiscale_metrics", "feature_trainings", "combined_feature_trainings",
                               "combined_image_feature_training")}


def make_inputs(arch, B, H, W, seed, flag_mode):
    g = torch.Generator().manual_seed(seed)
    feats, labels = {}, {}
    for f in arch.feature_predictions + arch.auxiliary_features:
        if not f.load_data:
            continue
        # positive, heavy-tailed render-pass-like values; some exact zeros so that sign(0) and the non-zero masks are exercised
        x = torch.randn(B, H, W, f.number_of_channels, generator=g, dtype=torch.float64).abs() * torch.exp(torch.randn(B, H, W, 1, generator=g, dtype=torch.float64))
        x = x * (torch.rand(B, H, W, 1, generator=g, dtype=torch.float64) > 0.1)
        if f.name == "Normal":
            x = x - 0.5      # signed inputs for signed_log1p
        feats[Naming.source_feature_name(f.name, index=0)] = x
    for f in arch.feature_predictions:
        if f.is_target and f.load_data:
            t = torch.randn(B, H, W, f.number_of_channels, generator=g, dtype=torch.float64).abs()
            labels[Naming.target_feature_name(f.name)] = t * (torch.rand(B, H, W, 1, generator=g, dtype=torch.float64) > 0.2)
    # members of a COMBINED tuple that no pass exists for ("Environment Direct"): the reference's input_fn makes them up -- ones for a colour,
    # 0.5 for direct / indirect, sources and targets alike (FeatureTrainingLoader.add_to_sources_dictionary / add_to_targets_dictionary, executed)
    for f in arch.feature_predictions:
        if not f.load_data:
            loader, one_s, one_t = RefTraining.FeatureTrainingLoader(f), {}, {}
            loader.add_to_sources_dictionary(one_s, None, (0,) * f.number_of_sources, H, W)
            loader.add_to_targets_dictionary(one_t, H, W)
            for k, v in one_s.items():
                feats[k] = v[None].repeat(B, 1, 1, 1)
            for k, v in one_t.items():
                labels[k] = v[None].repeat(B, 1, 1, 1)
    if flag_mode == "ONE_HOT_ENCODING":      # Training.py:757-758 / Prediction.py:97-98: the input_fn adds the constant one-hot planes
        ff = FeatureFlags([t.name for t in arch.feature_prediction_tuples], FeatureFlagMode.ONE_HOT_ENCODING, "channels_last")
        for t in arch.feature_prediction_tuples:
            plane = ff.feature_flag_name_to_feature_flags[t.name]      # [1, 1, V]
            feats[Naming.feature_flags_name(t.name)] = plane.reshape(1, 1, 1, -1).repeat(B, H, W, 1)
    return feats, labels


def run_case(name, aj, tj, B, H, W, seed, data_format="channels_last"):
    tf_stub.STORE.reset(seed)
    arch = RefArchitecture.Architecture(copy.deepcopy(aj), source_data_format="channels_last", data_format=data_format)
    flag_mode = aj["architecture"]["source_encoder"]["feature_flag_mode"]
    feats, labels = make_inputs(arch, B, H, W, seed + 1, flag_mode)
    out = {"architecture_json": json.dumps(aj), "training_json": json.dumps(tj), "data_format": data_format}
    params = {"architecture": arch, "learning_rate": tj["learning_rate"], "batch_size": tj["batch_size"]}
    params.update(build_trainings(arch, aj, copy.deepcopy(tj)))
    captured = {}
    real_predict = arch.predict

    def predict(features, mode):      # (model_fn calls architecture.predict itself: keep what it returned)
        captured["predictions"] = real_predict(features, mode)
        return captured["predictions"]
    arch.predict = predict
    spec = RefTraining.model_fn(dict(feats), dict(labels), tf.estimator.ModeKeys.TRAIN, params)
    preds, loss = captured["predictions"], spec.loss
    names = list(tf_stub.STORE.vars.keys())
    vars_ = [tf_stub.STORE.vars[n] for n in names]
    grads = torch.autograd.grad(loss, vars_, allow_unused=True)
    arrays = {}
    for k, v in feats.items():
        arrays["feature:" + k] = v.numpy()
    for k, v in labels.items():
        arrays["label:" + k] = v.numpy()
    for n, v, gr in zip(names, vars_, grads):
        arrays["var:" + n] = v.detach().numpy()
        arrays["grad:" + n] = (gr if gr is not None else torch.zeros_like(v)).numpy()
    for s, d in enumerate(preds):
        for k, v in d.items():
            arrays["prediction:%d:%s" % (s, k)] = v.detach().numpy()
    arrays["loss"] = np.array(float(loss.detach()))
    out["variables"] = names
    out["n_scales"] = len(preds)
    out["prediction_keys"] = [sorted(d.keys()) for d in preds]
    out["variable_log"] = ["%s %s" % kv for kv in tf_stub.STORE.log]
    print("%-28s %s  B=%d %dx%d  %d variables (%d scalars), %d scales, loss %.12g" % (
        name, data_format, B, H, W, len(names), sum(v.numel() for v in vars_), len(preds), float(loss.detach())))
    return out, arrays


def cases():
    small_combined = {k: configs._FULL_COMBINED[k] for k in ("Diffuse", "Volume", "Emission", "Alpha")}
    no_alpha = {k: v for k, v in configs._FULL_COMBINED.items() if k != "Alpha"}      # (masked terms on the Alpha pass raise: Training.py:103-113)
    small_no_alpha = {k: configs._FULL_COMBINED[k] for k in ("Glossy", "Volume", "Environment")}
    full_training = configs.training(feature_variation=0.5, masked_mean=0.25, combined_variation=0.25, image_variation=0.125, combined_masked_mean=0.5)
    c = []
    # the literal example configuration of the reference at reduced width: 17 SINGLE tuples, EMBEDDING flags, U-Net, 5x5 kernels, 3 scales,
    # every loss term switched on (feature / combined / image x mean / variation / masked)
    c.append(("example_single_embedding", configs.architecture(filters=(4, 6, 8), convs=2, combined=no_alpha), full_training, 1, 16, 16))
    c.append(("example_smape_defaults", configs.architecture(filters=(4, 6), convs=1), configs.training(), 2, 8, 12))
    # COMBINED tuples (three members per pass through the backbone), ONE_HOT flags, raw source for kernel prediction, 3x3 kernels,
    # standardization inverted BEFORE the scales are composed
    c.append(("combined_onehot_rawsource", configs.architecture(filters=(4, 6), convs=1, tuple_type="COMBINED", flag_mode="ONE_HOT_ENCODING",
                                                                kernel_size=3, standardized_kp_source=False, invert_after_multiscale=False,
                                                                combined=small_no_alpha, use_log1p=False),
              configs.training(loss_difference="ABSOLUTE", image_mean=0.0, combined_mean=2.0, combined_masked_mean=1.0, masked_mean=0.5), 2, 8, 8))
    # no kernel prediction (direct 3-channel outputs), no flags, single scale; Alpha (1 channel) among the passes
    c.append(("direct_noflags_singlescale", configs.architecture(filters=(4, 6), convs=2, flag_mode="NONE", kernel_prediction=False, multiscale=False,
                                                                 combined=small_combined, variance=False, use_log1p=False),
              configs.training(loss_difference="SQUARED", image_mean=0.0, combined_mean=1.0, multiscale_loss=False), 1, 8, 8))
    # BASELINE cfg-1 / cfg-2 / cfg-3 shapes at reduced width
    # (cfg-1 itself has NO auxiliary feature, which the reference cannot construct: Architecture.py:394-404 names the handling classes after the
    #  auxiliary loop's leftover `feature_name` -> UnboundLocalError; the product accepts it.  Same network with one plain auxiliary pass.)
    cfg1 = configs.cfg1_small_unet()
    cfg1["architecture"]["core_architecture"]["number_of_filters_for_convolution_blocks"] = [6, 8]
    cfg1["auxiliary_features"] = {"Normal": configs._aux(variance=False)}
    c.append(("cfg1_small_unet_plus_normal", cfg1, configs.bench_training(), 2, 16, 16))
    c.append(("cfg2_unet_kpcn", configs.cfg2_unet_kpcn(filters=(4, 6, 8), convs=2), configs.bench_training(), 1, 16, 16))
    c.append(("cfg3_tiramisu", configs.cfg3_tiramisu(filters=(4, 6, 8), convs=2), configs.bench_training(), 1, 16, 16))
    tir = configs.architecture(core="Tiramisu", filters=(4, 6), convs=2, combined=small_combined, flag_mode="EMBEDDING")
    c.append(("tiramisu_single_embedding", tir, configs.training(loss_difference="SMOOTH_ABSOLUTE", image_mean=0.0, combined_mean=3.0, feature_variation=1.0),
              1, 8, 8))
    return c


def main():
    meta, arrays = {}, {}
    for i, (name, aj, tj, B, H, W) in enumerate(cases()):
        m, a = run_case(name, aj, tj, B, H, W, seed=100 + i)
        meta[name] = m
        arrays.update({name + "|" + k: v for k, v in a.items()})
    # the same graph built channels_first (the reference's GPU layout, Training.py:964-967): its transposes are wiring too
    name, aj, tj, B, H, W = cases()[1]
    m, a = run_case(name + "_nchw", aj, tj, B, H, W, seed=101, data_format="channels_first")
    meta[name + "_nchw"] = m
    arrays.update({name + "_nchw|" + k: v for k, v in a.items()})
    np.savez_compressed(os.path.join(HERE, "wiring_golden.npz"), **arrays)
    with open(os.path.join(HERE, "wiring_golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote wiring_golden.npz (%.1f KiB), wiring_golden.json" % (os.path.getsize(os.path.join(HERE, "wiring_golden.npz")) / 1024))


if __name__ == "__main__":
    main()
