"""`python -m deepdenoiser_amd.predict architecture.json --input <frame directory>` -- the reference's `python Prediction.py architecture.json
--input dir` (TensorFlow/Prediction.py:23-53 argument set, :188-520 main): the directory's per-pass .exr files -> halo tiles -> forward on the
MI355X path (fp16 MFMA by default) -> crop / stitch / recombination -> <Pass>.npy and Combined.npy next to the inputs."""
import argparse
import json
import multiprocessing
import os

import numpy as np
import torch

from . import openexr, tf_checkpoint
from .architecture import Architecture
from .prediction import Predictor


def parser():
    p = argparse.ArgumentParser(description="Prediction for the DeepDenoiser (MI355X-native hot path).")
    p.add_argument("json_filename", help="The json specifying all the relevant details.")
    p.add_argument("--input", type=str, help="Make a prediction for the files in this directory.")
    p.add_argument("--tile_size", default=128, help="Width and heights of the tiles into which the image is split before denoising.")
    p.add_argument("--tile_overlap_size", default=14, help="Border size of the tiles that is overlapping to avoid artifacts.")
    p.add_argument("--threads", default=multiprocessing.cpu_count() + 1, help="Number of threads to use.")
    p.add_argument("--data_format", type=str, default="channels_first", choices=["channels_first", "channels_last"],
                   help="Accepted for compatibility: the MI355X path is NHWC-native, both values give the same results.")
    p.add_argument("--dtype", default="f16", choices=["bf16", "f16", "f32"], help="storage type of activations (f32: the 1e-4 parity path)")
    p.add_argument("--tiles_per_batch", type=int, default=256)
    p.add_argument("--exr", action="store_true", help="also write <Pass>.exr")
    return p


def main(args):
    aj = json.load(open(args.json_filename))
    assert os.path.isdir(args.input)
    arch = Architecture(aj, source_data_format="channels_last", data_format=args.data_format, device="cuda", dtype=args.dtype)
    feats = openexr.load_frame(args.input, arch)                                   # Prediction.py:223-252
    first = next(iter(feats.values()))
    height, width = first.shape[0], first.shape[1]
    predictor = Predictor(arch, tile_size=int(args.tile_size), tile_overlap_size=int(args.tile_overlap_size), tiles_per_batch=args.tiles_per_batch)
    predictor.prepare(height, width)                                               # raises for frames smaller than 16 pixels (Prediction.py:259-261)
    directory = os.path.dirname(os.path.abspath(args.json_filename))
    model_dir = aj["model_directory"] if os.path.isabs(aj["model_directory"]) else os.path.join(directory, aj["model_directory"])
    latest = tf_checkpoint.latest_checkpoint(model_dir) if os.path.isdir(model_dir) else None
    if latest is None:
        raise SystemExit("no checkpoint in %s (train first: python -m deepdenoiser_amd.train ...)" % model_dir)
    tf_checkpoint.load_variables(arch, latest, load_optimizer=False)               # Prediction.py:497-505 restores the Estimator's latest checkpoint
    out = predictor.predict_frame({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in feats.items()})
    for path in openexr.save_predictions(args.input, out, as_exr=args.exr):        # Prediction.py:483-510
        print(path)


if __name__ == "__main__":
    main(parser().parse_args())
