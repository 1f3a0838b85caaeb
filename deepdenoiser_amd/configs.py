"""Programmatic Architecture / Training JSON documents (same schema the reference parses:
Architecture.__init__, Architecture.py:343-365; Training.main, Training.py:948-989).

The five BASELINE.json configs are built here so tests, bench.py and smoke() share them.
`*_description` / batch-norm / dropout keys of the reference's example files are optional
(the reference never reads them, Architecture.py:506); unknown keys are tolerated.
"""

import copy

_VAR_ON = {"use_variance": True, "variance_mode": "uniform", "relative_variance": True,
           "compute_before_standardization": False, "compress_to_one_channel": True}
_VAR_OFF = dict(_VAR_ON, use_variance=False)


def _handling(use_log1p=True, variance=True, invert=True):
    return {"feature_variance": dict(_VAR_ON if variance else _VAR_OFF),
            "standardization": {"use_log1p": use_log1p, "mean": 0.0, "variance": 1.0},
            "invert_standardization": invert}


def _aux(channels=3, variance=True, use_log1p=False):
    return {"number_of_channels": channels,
            "feature_variance": dict(_VAR_ON if variance else _VAR_OFF),
            "standardization": {"use_log1p": use_log1p, "mean": 0.0, "variance": 1.0}}


_FULL_COMBINED = {
    "Diffuse": {"Color": "Diffuse Color", "Direct": "Diffuse Direct", "Indirect": "Diffuse Indirect"},
    "Glossy": {"Color": "Glossy Color", "Direct": "Glossy Direct", "Indirect": "Glossy Indirect"},
    "Subsurface": {"Color": "Subsurface Color", "Direct": "Subsurface Direct", "Indirect": "Subsurface Indirect"},
    "Transmission": {"Color": "Transmission Color", "Direct": "Transmission Direct", "Indirect": "Transmission Indirect"},
    "Volume": {"Color": "", "Direct": "Volume Direct", "Indirect": "Volume Indirect"},
    "Emission": {"Color": "Emission", "Direct": "", "Indirect": ""},
    "Environment": {"Color": "Environment", "Direct": "", "Indirect": ""},
    "Alpha": {"Color": "Alpha", "Direct": "", "Indirect": ""},
}


def architecture(core="U-Net", filters=(64, 96, 128), convs=4, tuple_type="SINGLE", flag_mode="EMBEDDING",
                 kernel_prediction=True, kernel_size=5, standardized_kp_source=True,
                 multiscale=True, invert_after_multiscale=True, combined=None, auxiliary=None,
                 variance=True, use_log1p=True, model_directory="../Models/Example"):
    combined = copy.deepcopy(_FULL_COMBINED if combined is None else combined)
    auxiliary = {"Normal": _aux()} if auxiliary is None else copy.deepcopy(auxiliary)
    return {
        "model_directory": model_directory,
        "number_of_sources_per_target": 1,
        "architecture": {
            "source_encoder": {"feature_prediction_tuple_type": tuple_type, "feature_flag_mode": flag_mode},
            "core_architecture": {"name": core, "number_of_filters_for_convolution_blocks": list(filters),
                                  "number_of_convolutions_per_block": convs,
                                  "use_batch_normalization": False, "dropout_rate": 0.0},
            "kernel_prediction": {"use_kernel_prediction": kernel_prediction, "kernel_size": kernel_size,
                                  "use_standardized_source_for_kernel_prediction": standardized_kp_source},
            "multiscale_prediction": {"use_multiscale_predictions": multiscale,
                                      "invert_standardization_after_multiscale_predictions": invert_after_multiscale},
        },
        "combined_features": combined,
        "combined_features_handling": {"Color": _handling(use_log1p, variance), "Direct": _handling(use_log1p, variance),
                                       "Indirect": _handling(use_log1p, variance)},
        "auxiliary_features": auxiliary,
    }


def example_architecture():
    """Content-equivalent of the reference's ArchitectureExample.json: SINGLE tuples (17), EMBEDDING flags,
    U-Net [64,96,128]x4, 5x5 KP, multiscale, Normal auxiliary => C_in = 16."""
    return architecture()


def cfg1_small_unet():
    """BASELINE config 1: small U-Net, 3-ch noisy RGB only, 64x64 tiles; direct 3-ch output, no KP / multiscale."""
    return architecture(filters=(16, 32), convs=2, flag_mode="NONE", kernel_prediction=False, multiscale=False,
                        combined={"Emission": {"Color": "Emission", "Direct": "", "Indirect": ""}},
                        auxiliary={}, variance=False, use_log1p=True)


_BENCH_AUX = ("Ambient Occlusion", "Depth", "Motion Vector", "Normal", "Screen Space Normal", "Shadow", "UV")


def cfg2_unet_kpcn(filters=(64, 96, 128), convs=4, core="U-Net"):
    """BASELINE config 2 (the metric's config): U-Net [64,96,128]x4 + 5x5 KP + 3-scale multiscale on a 32-channel
    render-pass stack: one SINGLE tuple = noisy pass (3+1 var) + 7 auxiliaries x (3+1 var) = 32 channels."""
    aux = {n: _aux(channels=1 if n == "Depth" else 3) for n in _BENCH_AUX}
    return architecture(core=core, filters=filters, convs=convs, flag_mode="NONE",
                        combined={"Emission": {"Color": "Emission", "Direct": "", "Indirect": ""}},
                        auxiliary=aux)


def cfg3_tiramisu(filters=(16, 24, 32), convs=4):
    """BASELINE config 3: Tiramisu (FC-DenseNet) + MultiScalePrediction on the 32-channel stack."""
    return cfg2_unet_kpcn(filters=filters, convs=convs, core="Tiramisu")


def training(learning_rate=1e-3, batch_size=8, loss_difference="SMAPE", multiscale_loss=True,
             feature_mean=1.0, combined_mean=5.0, image_mean=10.0, feature_variation=0.0, masked_mean=0.0,
             combined_variation=0.0, image_variation=0.0, combined_masked_mean=0.0):
    """Content-equivalent of the reference's TrainingExample.json (defaults) with a few knobs."""
    stats = {"track_mean": True, "track_variation": False, "track_ms_ssim": False,
             "track_difference_histogram": False, "track_variation_difference_histogram": False}
    stats_off = dict(stats, track_mean=False)

    def w(mean, variation=0.0):
        return {"mean": mean, "variation": variation, "ms_ssim": 0.0}

    return {
        "architecture": "ArchitectureExample.json",
        "base_tfrecords_directory": "../TFRecords/Example",
        "modes": ["training", "validation", "testing"],
        "number_of_source_index_tuples": 8,
        "learning_rate": learning_rate,
        "batch_size": batch_size,
        "data_augmentation": {"use_rotate_90": True, "use_flip_left_right": False,
                              "use_rgb_permutation": True, "use_normal_rotation": True},
        "loss_difference": loss_difference,
        "use_multiscale_loss": multiscale_loss,
        "use_multiscale_metrics": True,
        "combined_image_training_settings": {"loss_weights": w(image_mean, image_variation), "statistics": dict(stats if image_mean > 0 else stats_off)},
        "combined_features_training_settings": {"loss_weights": w(combined_mean, combined_variation), "loss_weights_masked": w(combined_masked_mean),
                                                "statistics": dict(stats if combined_mean > 0 else stats_off),
                                                "statistics_masked": dict(stats_off)},
        "features_training_settings": {"loss_weights": w(feature_mean, feature_variation), "loss_weights_masked": w(masked_mean),
                                       "statistics": dict(stats), "statistics_masked": dict(stats_off)},
    }


def bench_training():
    """Training settings for single-tuple bench architectures (no combined features exist there)."""
    return training(combined_mean=0.0, image_mean=0.0)
