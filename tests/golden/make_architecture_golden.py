"""Generates tests/golden/architecture_golden.json and tests/golden/recombine_golden.json by EXECUTING the reference's own lines.

Both ranges are plain Python / NumPy living inside modules that import TensorFlow at the top, so the modules cannot be imported here.
As tests/golden/make_tiling_golden.py does for the tiling code, this script reads the cited line ranges out of /root/reference AT
GENERATION TIME, dedents and exec()s them, and records only what they computed:

  * TensorFlow/Architecture.py:25-73, :77-191  the plain data classes (FeatureStandardization, FeatureVariance, FeaturePrediction[Type],
                                               FeaturePredictionTuple[Type]); only their constructors run
  * TensorFlow/Architecture.py:367-473         Architecture.__prepare_feature_predictions on the literal ArchitectureExample.json (SINGLE)
                                               and on its COMBINED variant: auxiliary features, feature predictions, tuples, in order
  * TensorFlow/Architecture.py:510-522         tuple size and number of output channels of the backbone post-processing
  * TensorFlow/FeatureFlags.py:12-48           FeatureFlags.__init__ in EMBEDDING mode: sorted flag names, vocabulary, embedding width
  * TensorFlow/Prediction.py:443-481           recombination of the predicted passes into the image (np.multiply / np.add sequence)

Nothing of the reference's text is embedded here or in the JSON files: the fixtures hold inputs (the example JSON's own values, seeds) and
the outputs the reference code produced.  Run in the build container only (/root/reference does not exist on the GPU box):
    python tests/golden/make_architecture_golden.py
"""
import copy
import hashlib
import json
import os
import sys
import textwrap
import types
from enum import Enum

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/TensorFlow"
sys.path.insert(0, REF)
from Naming import Naming  # noqa: E402  (pure Python in the reference)
from RenderPasses import RenderPasses  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def ref_lines(fname, first, last, must_contain):
    with open(os.path.join(REF, fname)) as f:
        lines = f.read().split("\n")
    block = lines[first - 1:last]
    assert must_contain[0] in block[0] and must_contain[1] in block[-1], (fname, first, last, block[0], block[-1])
    return compile(textwrap.dedent("\n".join(block)), "%s:%d-%d" % (fname, first, last), "exec")


CLASSES_A = ref_lines("Architecture.py", 25, 73, ("FeatureStandardization", "return result"))
CLASSES_B = ref_lines("Architecture.py", 76, 191, ("FeaturePredictionType", "self.name"))
PREPARE = ref_lines("Architecture.py", 367, 473, ("__prepare_feature_predictions", "append"))
OUTPUT_CHANNELS = ref_lines("Architecture.py", 510, 522, ("feature_prediction_tuple_type", "number_of_output_channels"))
FLAGS = ref_lines("FeatureFlags.py", 12, 48, ("FeatureFlagMode", "embedding_dimension"))
RECOMBINE = ref_lines("Prediction.py", 443, 481, ("diffuse_direct", "emission"))


def run_architecture(parsed_json):
    ns = {"Enum": Enum, "RenderPasses": RenderPasses, "Naming": Naming}
    exec(CLASSES_A, ns)
    exec(CLASSES_B, ns)
    exec(PREPARE, ns)            # defines __prepare_feature_predictions as a module-level function: no name mangling of self.__x
    arch_json = parsed_json["architecture"]
    self = types.SimpleNamespace()
    self.number_of_sources_per_target = parsed_json["number_of_sources_per_target"]
    self.feature_prediction_tuple_type = ns["FeaturePredictionTupleType"][arch_json["source_encoder"]["feature_prediction_tuple_type"]]
    setattr(self, "__preserve_source", not arch_json["kernel_prediction"]["use_standardized_source_for_kernel_prediction"])
    ns["__prepare_feature_predictions"](self, parsed_json["combined_features"], parsed_json["combined_features_handling"],
                                        parsed_json["auxiliary_features"])

    def fp_record(fp):
        if fp is None:
            return None
        st, fv = fp.feature_standardization, fp.feature_variance
        return {"name": fp.name, "type": fp.feature_prediction_type.name, "load_data": bool(fp.load_data), "is_target": bool(fp.is_target),
                "number_of_sources": int(fp.number_of_sources), "preserve_source": bool(fp.preserve_source),
                "number_of_channels": int(fp.number_of_channels), "invert_standardization": bool(fp.invert_standardization),
                "standardization": {"use_log1p": bool(st.use_log1p), "mean": float(st.mean), "variance": float(st.variance)},
                "feature_variance": {"use_variance": bool(fv.use_variance), "variance_mode": fv.variance_mode,
                                     "relative_variance": bool(fv.relative_variance),
                                     "compute_before_standardization": bool(fv.compute_before_standardization),
                                     "compress_to_one_channel": bool(fv.compress_to_one_channel)}}
    out = {"auxiliary_features": [fp_record(f) for f in self.auxiliary_features],
           "feature_predictions": [fp_record(f) for f in self.feature_predictions],
           "feature_prediction_tuples": [{"name": t.name, "type": t.feature_prediction_tuple_type.name,
                                          "members": [None if f is None else f.name for f in t.feature_predictions]}
                                         for t in self.feature_prediction_tuples]}
    # Architecture.py:510-522 (tuple size, output channels of AdjustNumberOfChannels)
    ns2 = {"self": types.SimpleNamespace(feature_prediction_tuple_type=self.feature_prediction_tuple_type,
                                         use_kernel_prediction=arch_json["kernel_prediction"]["use_kernel_prediction"],
                                         number_of_sources_per_target=self.number_of_sources_per_target),
           "FeaturePredictionTupleType": ns["FeaturePredictionTupleType"], "kernel_prediction_json": arch_json["kernel_prediction"]}
    exec(OUTPUT_CHANNELS, ns2)
    out["feature_prediction_tuple_size"] = int(ns2["feature_prediction_tuple_size"])
    out["number_of_output_channels"] = int(ns2["number_of_output_channels"])
    # FeatureFlags.__init__ (EMBEDDING never touches TensorFlow there)
    ns3 = {"Enum": Enum}
    exec(FLAGS, ns3)
    ff = ns3["FeatureFlags"]([t.name for t in self.feature_prediction_tuples], ns3["FeatureFlagMode"]["EMBEDDING"], "channels_last")
    out["feature_flag_names"] = list(ff.feature_flag_names)
    out["vocabulary_size"], out["embedding_dimension"] = int(ff.vocabulary_size), int(ff.embedding_dimension)
    return out


def variant(example, **changes):
    j = copy.deepcopy(example)
    for path, value in changes.items():
        node = j
        keys = path.split("__")
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = value
    return j


def run_recombine(seed, shape, scale):
    rng = np.random.default_rng(seed)
    names = [getattr(RenderPasses, n) for n in (
        "DIFFUSE_DIRECT", "DIFFUSE_INDIRECT", "DIFFUSE_COLOR", "GLOSSY_DIRECT", "GLOSSY_INDIRECT", "GLOSSY_COLOR",
        "SUBSURFACE_DIRECT", "SUBSURFACE_INDIRECT", "SUBSURFACE_COLOR", "TRANSMISSION_DIRECT", "TRANSMISSION_INDIRECT", "TRANSMISSION_COLOR",
        "VOLUME_DIRECT", "VOLUME_INDIRECT", "ENVIRONMENT", "EMISSION", "ALPHA")]
    # HDR-like magnitudes over several decades, some exact zeros and negative values: every add / multiply rounds
    passes = {}
    for n in names:
        v = rng.standard_normal(shape) * np.exp(scale * rng.standard_normal(shape))
        v[rng.random(shape) < 0.05] = 0.0
        passes[n] = v.astype(np.float32)
    ns = {"predictions": {Naming.feature_prediction_name(n): v for n, v in passes.items()}, "Naming": Naming, "RenderPasses": RenderPasses, "np": np}
    exec(RECOMBINE, ns)
    image = ns["image"]
    assert image.dtype == np.float32
    return passes, image


def f32_hex(a):
    return np.ascontiguousarray(a, dtype="<f4").tobytes().hex()


def main():
    example = json.load(open(os.path.join(REF, "ArchitectureExample.json")))
    arch_out = {"_generator": "tests/golden/make_architecture_golden.py",
                "_source": ["TensorFlow/Architecture.py:25-73", "TensorFlow/Architecture.py:76-191", "TensorFlow/Architecture.py:367-473",
                            "TensorFlow/Architecture.py:510-522", "TensorFlow/FeatureFlags.py:12-48", "TensorFlow/ArchitectureExample.json"],
                "cases": []}
    cases = [
        ("example (SINGLE)", {}),
        ("COMBINED", {"architecture__source_encoder__feature_prediction_tuple_type": "COMBINED"}),
        ("SINGLE, preserved source, no kernel prediction",
         {"architecture__kernel_prediction__use_standardized_source_for_kernel_prediction": False,
          "architecture__kernel_prediction__use_kernel_prediction": False}),
        ("COMBINED, 3x3 kernels", {"architecture__source_encoder__feature_prediction_tuple_type": "COMBINED",
                                   "architecture__kernel_prediction__kernel_size": 3}),
    ]
    for label, changes in cases:
        pj = variant(example, **changes)
        arch_out["cases"].append({"label": label, "changes": changes, "result": run_architecture(pj)})
    # the example JSON's own settings, so that the test can rebuild each case's input without /root/reference
    arch_out["example_json"] = example
    with open(os.path.join(HERE, "architecture_golden.json"), "w") as f:
        json.dump(arch_out, f, indent=1, sort_keys=True)

    rec_out = {"_generator": "tests/golden/make_architecture_golden.py", "_source": ["TensorFlow/Prediction.py:443-481"],
               "_inputs": "passes[name] for name in `order`: v = rng.standard_normal(shape) * exp(scale * rng.standard_normal(shape)); "
                          "v[rng.random(shape) < 0.05] = 0; float32 -- rng = numpy.random.default_rng(seed), drawn in `order`",
               "cases": []}
    for seed, shape, scale, full in ((0, (6, 5, 3), 1.5, True), (1, (50, 40, 3), 2.5, False), (2, (270, 480, 3), 3.0, False)):
        passes, image = run_recombine(seed, shape, scale)
        case = {"seed": seed, "shape": list(shape), "scale": scale, "order": list(passes.keys()),
                "image_sha256": hashlib.sha256(np.ascontiguousarray(image, dtype="<f4").tobytes()).hexdigest(),
                "inputs_sha256": hashlib.sha256(b"".join(np.ascontiguousarray(v, dtype="<f4").tobytes() for v in passes.values())).hexdigest()}
        if full:        # small case: the exact bytes of inputs and output (little-endian float32, hex)
            case["inputs_hex"] = {n: f32_hex(v) for n, v in passes.items()}
            case["image_hex"] = f32_hex(image)
        rec_out["cases"].append(case)
    with open(os.path.join(HERE, "recombine_golden.json"), "w") as f:
        json.dump(rec_out, f, indent=1, sort_keys=True)
    print("wrote architecture_golden.json (%d cases) and recombine_golden.json (%d cases)" % (len(arch_out["cases"]), len(rec_out["cases"])))


if __name__ == "__main__":
    main()
