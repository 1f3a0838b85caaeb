"""Render-pass name contract.

Mirrors the behaviour of the reference's pass-name helpers
(reference: TensorFlow/RenderPasses.py:6-37 constants, :40-44 channel counts,
:46-112 predicates/mappers, :171-227 usage -> ordered list).  Pass names are
dictionary keys and file-name fragments at the drop-in boundary, so the strings
are part of the contract; tests/golden/naming_golden.json (generated from the
reference module itself) pins them.
"""

# Canonical order = the order in which the reference's RenderPassesUsage emits
# passes (reference: TensorFlow/RenderPasses.py:171-227).
_ORDERED = (
    ("ALPHA", "Alpha"), ("DEPTH", "Depth"), ("MIST", "Mist"), ("NORMAL", "Normal"),
    ("SCREEN_SPACE_NORMAL", "Screen Space Normal"), ("MOTION_VECTOR", "Motion Vector"),
    ("OBJECT_ID", "Object ID"), ("MATERIAL_ID", "Material ID"), ("UV", "UV"),
    ("SHADOW", "Shadow"), ("AMBIENT_OCCLUSION", "Ambient Occlusion"),
    ("EMISSION", "Emission"), ("ENVIRONMENT", "Environment"),
    ("DIFFUSE_COLOR", "Diffuse Color"), ("DIFFUSE_DIRECT", "Diffuse Direct"),
    ("DIFFUSE_INDIRECT", "Diffuse Indirect"),
    ("GLOSSY_COLOR", "Glossy Color"), ("GLOSSY_DIRECT", "Glossy Direct"),
    ("GLOSSY_INDIRECT", "Glossy Indirect"),
    ("TRANSMISSION_COLOR", "Transmission Color"), ("TRANSMISSION_DIRECT", "Transmission Direct"),
    ("TRANSMISSION_INDIRECT", "Transmission Indirect"),
    ("SUBSURFACE_COLOR", "Subsurface Color"), ("SUBSURFACE_DIRECT", "Subsurface Direct"),
    ("SUBSURFACE_INDIRECT", "Subsurface Indirect"),
    ("VOLUME_DIRECT", "Volume Direct"), ("VOLUME_INDIRECT", "Volume Indirect"),
)

_COMBINED_FEATURES = ("Diffuse", "Glossy", "Subsurface", "Transmission")
_NON_RGB = frozenset(("Alpha", "Depth", "Mist", "Normal", "Screen Space Normal",
                      "Motion Vector", "Object ID", "Material ID", "UV"))
_SINGLE_CHANNEL = frozenset(("Alpha", "Depth"))
# passes whose "colour pass" is the pass itself (reference :91-102, :110-121)
_SELF_COLOR_PREFIXES = ("Alpha", "Emission", "Environment", "Ambient Occlusion", "Shadow")


class RenderPasses:
    COMBINED = "Combined"
    COMBINED_DIFFUSE, COMBINED_GLOSSY, COMBINED_SUBSURFACE, COMBINED_TRANSMISSION = _COMBINED_FEATURES

    @staticmethod
    def number_of_channels(render_pass_name):
        return 1 if render_pass_name in _SINGLE_CHANNEL else 3

    @staticmethod
    def is_combined_feature_render_pass(render_pass_name):
        return render_pass_name in _COMBINED_FEATURES

    @staticmethod
    def is_volume_render_pass(render_pass_name):
        return "Volume" in render_pass_name

    @staticmethod
    def is_direct_or_indirect_render_pass(render_pass_name):
        return render_pass_name.endswith((" Direct", " Indirect"))

    @staticmethod
    def is_color_render_pass(render_pass_name):
        return render_pass_name.endswith(" Color")

    @staticmethod
    def is_rgb_color_render_pass(render_pass_name):
        return render_pass_name not in _NON_RGB

    @staticmethod
    def _self_color_special_case(name):
        for prefix in _SELF_COLOR_PREFIXES:
            if name.startswith(prefix):
                return prefix
        return name

    @staticmethod
    def direct_or_indirect_to_color_render_pass(render_pass_name):
        # Behaviour kept, including the reference's quirk (RenderPasses.py:88 looks for
        # ' Inirect'): an '... Indirect' pass maps to ITSELF, not to its colour pass
        # (SURVEY Appendix E).
        if render_pass_name.endswith(" Direct"):
            result = render_pass_name.replace(" Direct", " Color")
        elif render_pass_name.endswith(" Indirect"):
            result = render_pass_name
        else:
            raise AttributeError("not a direct/indirect pass: %r" % (render_pass_name,))
        return RenderPasses._self_color_special_case(result)

    @staticmethod
    def combined_to_color_render_pass(render_pass_name):
        return RenderPasses._self_color_special_case(render_pass_name + " Color")

    @staticmethod
    def combined_to_direct_render_pass(render_pass_name):
        return render_pass_name + " Direct"

    @staticmethod
    def combined_to_indirect_render_pass(render_pass_name):
        return render_pass_name + " Indirect"


for _attr, _name in _ORDERED:
    setattr(RenderPasses, _attr, _name)


class RenderPassesUsage:
    """use_<pass> flags -> ordered pass list (reference: RenderPasses.py:114-227)."""

    def __init__(self, **flags):
        known = {"use_" + attr.lower() for attr, _ in _ORDERED}
        unknown = set(flags) - known
        if unknown:
            raise TypeError("unexpected keyword argument(s): %s" % sorted(unknown))
        for key in known:
            setattr(self, key, bool(flags.get(key, False)))

    def render_passes(self):
        return [name for attr, name in _ORDERED if getattr(self, "use_" + attr.lower())]
