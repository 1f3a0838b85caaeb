"""Generates tests/golden/naming_golden.json from the REFERENCE's own pure-Python
modules (TensorFlow/Naming.py, TensorFlow/RenderPasses.py import without TensorFlow).
Run in the build container only (/root/reference does not exist on the GPU box):
    python tests/golden/make_naming_golden.py
The committed JSON is data (inputs + the reference's outputs), not reference source.
"""
import itertools, json, os, sys
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/TensorFlow")
from Naming import Naming                      # noqa: E402
from RenderPasses import RenderPasses, RenderPassesUsage   # noqa: E402

names = sorted(v for k, v in vars(RenderPasses).items() if k.isupper() and isinstance(v, str))
out = {"pass_constants": {k: v for k, v in vars(RenderPasses).items() if k.isupper() and isinstance(v, str)}}

preds = {}
for fn in ("number_of_channels", "is_combined_feature_render_pass", "is_volume_render_pass",
           "is_direct_or_indirect_render_pass", "is_color_render_pass", "is_rgb_color_render_pass",
           "combined_to_color_render_pass", "combined_to_direct_render_pass", "combined_to_indirect_render_pass"):
    preds[fn] = {n: getattr(RenderPasses, fn)(n) for n in names}
d2c = {}
for n in names:
    try:
        d2c[n] = RenderPasses.direct_or_indirect_to_color_render_pass(n)
    except AttributeError:
        d2c[n] = "__AttributeError__"
preds["direct_or_indirect_to_color_render_pass"] = d2c
out["render_pass_functions"] = preds

usage_cases = []
import inspect
flags = [p for p in inspect.signature(RenderPassesUsage.__init__).parameters if p.startswith("use_")]
for sel in ([], flags, flags[::2], flags[1::3], ["use_normal", "use_alpha", "use_volume_indirect", "use_diffuse_color"]):
    kw = {f: True for f in sel}
    usage_cases.append({"flags": sorted(sel), "passes": RenderPassesUsage(**kw).render_passes()})
out["usage_cases"] = usage_cases
out["usage_flag_order"] = flags

calls = []
for n in ["Normal", "Diffuse", "Diffuse Color", "Glossy", "Volume Direct", "Alpha", "Combined"]:
    for spp, idx, masked in itertools.product([None, 16], [None, 0, 1], [False, True]):
        calls.append(["source_feature_name", [n], {"samples_per_pixel": spp, "index": idx, "masked": masked},
                      Naming.source_feature_name(n, samples_per_pixel=spp, index=idx, masked=masked)])
    for masked in (False, True):
        calls.append(["target_feature_name", [n], {"masked": masked}, Naming.target_feature_name(n, masked=masked)])
    calls.append(["feature_prediction_name", [n], {}, Naming.feature_prediction_name(n)])
    calls.append(["feature_flags_name", [n], {}, Naming.feature_flags_name(n)])
    calls.append(["tensorboard_name", [n], {}, Naming.tensorboard_name(n)])
    for fn in ("difference_name", "mean_name", "variation_difference_name", "variation_mean_name"):
        for masked, internal, si in itertools.product([False, True], [False, True], [None, 0, 2]):
            calls.append([fn, [n], {"masked": masked, "internal": internal, "scale_index": si},
                          getattr(Naming, fn)(n, masked=masked, internal=internal, scale_index=si)])
    for masked in (False, True):
        calls.append(["ms_ssim_name", [n], {"masked": masked}, Naming.ms_ssim_name(n, masked=masked)])
out["naming_calls"] = calls

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "naming_golden.json")
with open(path, "w") as f:
    json.dump(out, f, indent=0, sort_keys=True)
print("wrote", path, len(calls), "naming calls")
