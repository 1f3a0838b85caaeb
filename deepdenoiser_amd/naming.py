"""Feature-dictionary keys and metric names at the drop-in boundary.

Same strings as the reference produces (reference: TensorFlow/Naming.py:57-81 for
the tensor dictionary keys, :12-51 and :83-102 for metric names).  Pinned by
tests/golden/naming_golden.json, which is generated from the reference module.
"""

from .render_passes import RenderPasses


def _suffixes(name, masked=False, internal=False, scale_index=None):
    if masked:
        name += " Masked"
    if internal:
        name += " Internal"
    if scale_index is not None:
        name += "/" + str(2 ** scale_index)
    return name


class Naming:
    # ---- tensor dictionary keys (reference Naming.py:57-81) ----
    @staticmethod
    def source_feature_name(name, samples_per_pixel=None, index=None, masked=False):
        parts = ["source_image"]
        if samples_per_pixel is not None:
            parts.append(str(samples_per_pixel))
        if index is not None:
            parts.append(str(index))
        parts.append(name)
        return _suffixes("/".join(parts), masked=masked)

    @staticmethod
    def feature_flags_name(name):
        return "feature_flag/" + name

    @staticmethod
    def target_feature_name(name, masked=False):
        return _suffixes("target_image/" + name, masked=masked)

    @staticmethod
    def feature_prediction_name(name):
        return "prediction/" + name

    # ---- metric names (reference Naming.py:12-51) ----
    @staticmethod
    def tensorboard_name(name):
        return name.lower().replace(" ", "_")

    @staticmethod
    def _statistics_name(name, statistics_name, masked=False, internal=False, scale_index=None):
        if RenderPasses.is_combined_feature_render_pass(name):
            name = "Combined " + name
        return Naming.tensorboard_name(
            _suffixes(name + statistics_name, masked=masked, internal=internal, scale_index=scale_index))

    # Only mean_name forwards `internal` in the reference (Naming.py:13-35); kept.
    @staticmethod
    def difference_name(name, masked=False, internal=False, scale_index=None):
        return Naming._statistics_name(name, " Difference", masked=masked, scale_index=scale_index)

    @staticmethod
    def mean_name(name, masked=False, internal=False, scale_index=None):
        return Naming._statistics_name(name, " Mean", masked=masked, internal=internal, scale_index=scale_index)

    @staticmethod
    def variation_difference_name(name, masked=False, internal=False, scale_index=None):
        return Naming._statistics_name(name, " Variation Difference", masked=masked, scale_index=scale_index)

    @staticmethod
    def variation_mean_name(name, masked=False, internal=False, scale_index=None):
        return Naming._statistics_name(name, " Variation Mean", masked=masked, scale_index=scale_index)

    @staticmethod
    def ms_ssim_name(name, masked=False, internal=False):
        return Naming._statistics_name(name, " MS SSIM", masked=masked)
