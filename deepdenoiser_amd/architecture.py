"""Architecture.json -> MI355X launch programs.  Mirrors the reference's model entry
(`Architecture(parsed_json, source_data_format, data_format)`, `predict(features, mode) -> list[dict]`,
reference TensorFlow/Architecture.py:341-617) and lowers it onto the static executor (engine.py).

MI355X-first differences from the reference's graph (results are the same, SURVEY.md section 7 "hard parts"):
  * the weight-shared backbone passes of all feature-prediction tuples (17 for the example JSON,
    Architecture.py:561-571) are FOLDED INTO THE BATCH (B_eff = tuples x B): one launch per layer, GEMM-M large,
    and the shared-weight gradient accumulates inside one wgrad kernel;
  * NHWC everywhere: the reference's NCHW transposes (SourceEncoder.py:76-77, Architecture.py:334-338) and concat
    copies (UNet.py:91-92, Tiramisu.py:40) disappear -- convs write straight into channel ranges of concat buffers;
  * standardize + variance + concat (+ embedding broadcast) are one prepare kernel per pass and one gather kernel.
"""

import copy
import ctypes as C
import json
import math

import torch

from . import _lib as L
from .engine import DT, Graph, ParamStore, round_up
from .naming import Naming
from .render_passes import RenderPasses


class ModeKeys:
    TRAIN, EVAL, PREDICT = "train", "eval", "infer"     # values of tf.estimator.ModeKeys


# ----------------------------------------------------------------------------------------------- JSON structure
class FeatureStandardization:
    """Architecture.py:25-55."""

    def __init__(self, use_log1p, mean, variance, name):
        self.use_log1p, self.mean, self.variance, self.name = bool(use_log1p), float(mean), float(variance), name

    def key(self):
        return (self.use_log1p, self.mean, self.variance)


class FeatureVariance:
    """Architecture.py:58-73."""

    def __init__(self, j, name):
        self.use_variance = bool(j["use_variance"])
        self.variance_mode = j["variance_mode"]
        self.relative_variance = bool(j["relative_variance"])
        self.compute_before_standardization = bool(j["compute_before_standardization"])
        self.compress_to_one_channel = bool(j["compress_to_one_channel"])
        self.name = name
        assert self.variance_mode in ("uniform", "neighbor")

    def channels(self, source_channels):
        if not self.use_variance:
            return 0
        return 1 if self.compress_to_one_channel else source_channels


class FeaturePrediction:
    """Architecture.py:82-165 (state that is per-graph in the reference lives in the Program here)."""

    def __init__(self, feature_prediction_type, load_data, is_target, standardization, invert_standardization,
                 feature_variance, number_of_channels, name):
        self.feature_prediction_type, self.load_data, self.is_target = feature_prediction_type, load_data, is_target
        self.feature_standardization, self.invert_standardization = standardization, invert_standardization
        self.feature_variance, self.number_of_channels, self.name = feature_variance, number_of_channels, name


class FeaturePredictionTuple:
    def __init__(self, feature_predictions, feature_prediction_tuple_type, name):
        self.feature_predictions, self.feature_prediction_tuple_type, self.name = feature_predictions, feature_prediction_tuple_type, name


class Architecture:
    def __init__(self, parsed_json, source_data_format="channels_last", data_format="channels_last",
                 device="cuda", dtype="f32", seed=2, loss_scale=None):
        # `data_format` is accepted for drop-in compatibility; the MI355X path is NHWC-native, both values give the same results.
        if source_data_format != "channels_last":
            raise ValueError("features are exchanged channels_last (NHWC), as in the reference's callers")
        self.source_data_format, self.data_format = source_data_format, data_format
        assert dtype in ("f32", "bf16", "f16"), dtype
        self.device, self.dtype, self.seed = torch.device(device), dtype, seed
        self.loss_scale = loss_scale          # None: the storage type's default (program.Program)
        self.model_directory = parsed_json["model_directory"]
        self.number_of_sources_per_target = parsed_json["number_of_sources_per_target"]
        if self.number_of_sources_per_target != 1:
            raise ValueError("number_of_sources_per_target: the only valid value is 1 (ArchitectureExample.json:5-6)")
        arch = parsed_json["architecture"]
        self.feature_prediction_tuple_type = arch["source_encoder"]["feature_prediction_tuple_type"]
        assert self.feature_prediction_tuple_type in ("SINGLE", "COMBINED")
        self.feature_flag_mode = arch["source_encoder"]["feature_flag_mode"]
        assert self.feature_flag_mode in ("NONE", "ONE_HOT_ENCODING", "EMBEDDING")
        core = arch["core_architecture"]
        self.core_name = core["name"]
        assert self.core_name in ("U-Net", "Tiramisu")
        self.filters = list(core["number_of_filters_for_convolution_blocks"])
        self.convs_per_block = int(core["number_of_convolutions_per_block"])
        kp = arch["kernel_prediction"]
        self.use_kernel_prediction = bool(kp["use_kernel_prediction"])
        self.kernel_size = int(kp["kernel_size"])
        self.use_standardized_source_for_kernel_prediction = bool(kp["use_standardized_source_for_kernel_prediction"])
        ms = arch["multiscale_prediction"]
        self.use_multiscale_predictions = bool(ms["use_multiscale_predictions"])
        self.invert_standardization_after_multiscale_predictions = bool(ms["invert_standardization_after_multiscale_predictions"])
        self._prepare_feature_predictions(parsed_json["combined_features"], parsed_json["combined_features_handling"],
                                          parsed_json["auxiliary_features"])
        tuple_size = 1 if self.feature_prediction_tuple_type == "SINGLE" else 3
        self.tuple_size = tuple_size
        if self.use_kernel_prediction:     # Architecture.py:515-522
            self.member_channels = self.kernel_size ** 2
        else:
            self.member_channels = 3
        self.number_of_output_channels = self.number_of_sources_per_target * tuple_size * self.member_channels
        self.feature_flag_names = sorted(t.name for t in self.feature_prediction_tuples)   # FeatureFlags.py:22
        self.params = ParamStore()
        self._programs = {}

    def _prepare_feature_predictions(self, combined_json, handling_json, auxiliary_json):
        """Architecture.py:367-473 (sorted names => deterministic order)."""
        self.auxiliary_features = []
        for name in sorted(auxiliary_json.keys()):
            j = auxiliary_json[name]
            s = j["standardization"]
            self.auxiliary_features.append(FeaturePrediction(
                "AUXILIARY", True, False, FeatureStandardization(s["use_log1p"], s["mean"], s["variance"], name), False,
                FeatureVariance(j["feature_variance"], name), j["number_of_channels"], name))
        self.feature_predictions, self.feature_prediction_tuples = [], []
        self.combined_feature_names = []     # (combined name, [color, direct, indirect pass names]) incl. generated names
        for cname in sorted(combined_json.keys()):
            members = []
            self.combined_feature_names.append((cname, [combined_json[cname][t] or (cname + " " + t) for t in ("Color", "Direct", "Indirect")]))
            for ftype in ("Color", "Direct", "Indirect"):
                fname = combined_json[cname][ftype]
                h = handling_json[ftype]
                s = h["standardization"]
                channels = RenderPasses.number_of_channels(fname)
                load_data = True
                if fname is None or fname == "":
                    fname, load_data = cname + " " + ftype, False
                fp = None
                if load_data or self.feature_prediction_tuple_type == "COMBINED":
                    fp = FeaturePrediction(ftype.upper(), load_data, True,
                                           FeatureStandardization(s["use_log1p"], s["mean"], s["variance"], fname),
                                           bool(h["invert_standardization"]), FeatureVariance(h["feature_variance"], fname),
                                           channels, fname)
                    self.feature_predictions.append(fp)
                members.append(fp)
            if self.feature_prediction_tuple_type == "COMBINED":
                self.feature_prediction_tuples.append(FeaturePredictionTuple(members, "COMBINED", cname))
        if self.feature_prediction_tuple_type == "SINGLE":
            for fp in self.feature_predictions:
                self.feature_prediction_tuples.append(FeaturePredictionTuple([fp], "SINGLE", fp.name))

    # ------------------------------------------------------------------ derived sizes
    def input_channels(self):
        """Channel accounting of SourceEncoder.prepare_neural_network_input."""
        t0 = self.feature_prediction_tuples[0]
        c = 0
        for f in list(t0.feature_predictions) + self.auxiliary_features:
            c += 3 + f.feature_variance.channels(f.number_of_channels)
        if self.feature_flag_mode == "ONE_HOT_ENCODING":
            c += len(self.feature_flag_names)
        elif self.feature_flag_mode == "EMBEDDING":
            c += len(self.feature_flag_names) // 2
        return c

    def number_of_scales(self):
        return len(self.filters) if self.use_multiscale_predictions else 1

    def required_source_names(self):
        return [Naming.source_feature_name(f.name, index=0) for f in self.feature_predictions + self.auxiliary_features]

    # ------------------------------------------------------------------ programs
    def program(self, B, H, W, training_json=None, architecture_json=None):
        """Static launch program for a (batch, tile size[, training settings]) configuration (cached)."""
        # keyed on the CONTENT of the training settings: a mutated or re-created dict must not alias a stale program / loss descriptor
        # (the learning rate is read at every optimizer step, not baked into the launch program: a schedule must not build a program per value)
        tkey = None
        if training_json is not None:
            tkey = json.dumps({k: v for k, v in training_json.items() if k != "learning_rate"}, sort_keys=True, default=str)
        key = (B, H, W, tkey)
        if key not in self._programs:
            from .program import Program
            self._programs[key] = Program(self, B, H, W, None if training_json is None else copy.deepcopy(training_json))
        elif training_json is not None:
            self._programs[key].training_json["learning_rate"] = training_json["learning_rate"]
        return self._programs[key]

    def predict(self, features, mode=ModeKeys.PREDICT):
        """Drop-in for Architecture.predict (Architecture.py:537-617): dict of NHWC float32 tensors
        `source_image/<i>/<Pass>` -> list (one per scale, largest first) of dicts `prediction/<Pass>`."""
        first = features[Naming.source_feature_name(self.feature_predictions[0].name, index=0)]
        B, H, W = int(first.shape[0]), int(first.shape[1]), int(first.shape[2])
        prog = self.program(B, H, W)
        prog.set_inputs(features)
        prog.forward()
        return prog.prediction_dictionaries()
