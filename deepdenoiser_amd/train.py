"""`python -m deepdenoiser_amd.train training.json` -- the reference's `python Training.py training.json` (TensorFlow/Training.py:33-61 argument
set, :944-1289 main) on the MI355X path: TFRecord tiles -> device augmentation -> Trainer.step -> TensorFlow-format checkpoints in the
architecture's model_directory.  One process per GPU under torch.distributed.run shards every mini-batch over the ranks.

Not reproduced (out of scope): the Estimator's evaluation / TensorBoard summaries; `--validate` reports the mean validation loss."""
import argparse
import json
import multiprocessing
import os
import random

import numpy as np
import torch

from . import tf_checkpoint, tfrecords
from .architecture import Architecture
from .data_augmentation import DataAugmentation, DataAugmentationUsage
from .naming import Naming
from .tiling import source_index_tuples
from .training import Trainer


def parser():
    p = argparse.ArgumentParser(description="Training for the DeepDenoiser (MI355X-native hot path).")
    p.add_argument("json_filename", help="The json specifying all the relevant details.")
    p.add_argument("--validate", action="store_true", help="Perform a validation step.")
    p.add_argument("--threads", default=multiprocessing.cpu_count() + 1, help="Number of threads to use (host-side decoding)")
    p.add_argument("--train_epochs", type=int, default=10000, help="Number of epochs to train.")
    p.add_argument("--validation_interval", type=int, default=1, help="Number of epochs after which a validation is made.")
    p.add_argument("--data_format", type=str, default="channels_first", choices=["channels_first", "channels_last"],
                   help="Accepted for compatibility: the MI355X path is NHWC-native, both values give the same results.")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"], help="storage type of activations (accumulation is fp32)")
    return p


def batches(base, mode, arch, batch, spp, index_tuples, rank=0, world=1):
    """Mini-batches of `batch` tiles for this rank: (features, labels) dictionaries of device tensors (Training.py:502-604 input_fn)."""
    st = tfrecords.read_settings(base, mode)
    tile = st["tiles_height_width"]
    passes = {f.name: f.number_of_channels for f in arch.feature_predictions + arch.auxiliary_features if f.load_data}
    targets = [f.name for f in arch.feature_predictions if f.load_data and f.is_target]
    required = sorted({i for t in index_tuples for i in t})
    buf, n = [], 0
    for path in tfrecords.list_files(os.path.join(base, mode), mode):
        for rec in tfrecords.read_records(path):
            src, tgt = tfrecords.decode_example(tfrecords.parse_example(rec), passes, tile, [spp], required, targets)
            for tup in index_tuples:                      # one training example per (source index tuple, target): Training.py:544-549
                n += 1
                if (n - 1) // batch % world != rank:      # whole mini-batches round-robin over the ranks
                    continue
                buf.append((src[spp][tup[0]], tgt))
                if len(buf) == batch:
                    feats = {Naming.source_feature_name(k, index=0): torch.from_numpy(np.stack([s[k] for s, _ in buf])).to(arch.device) for k in passes}
                    labels = {Naming.target_feature_name(k): torch.from_numpy(np.stack([t[k] for _, t in buf])).to(arch.device) for k in targets}
                    for f in arch.feature_predictions + arch.auxiliary_features:      # generated passes (Training.py:531-538)
                        if not f.load_data:
                            value = 1.0 if f.feature_prediction_type == "COLOR" else 0.5
                            feats[Naming.source_feature_name(f.name, index=0)] = torch.full((batch, tile, tile, f.number_of_channels), value, device=arch.device)
                            if f.is_target:
                                labels[Naming.target_feature_name(f.name)] = torch.full((batch, tile, tile, f.number_of_channels), value, device=arch.device)
                    yield feats, labels
                    buf = []


def main(args):
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    tj = json.load(open(args.json_filename, encoding="utf-8"))
    directory = os.path.dirname(os.path.abspath(args.json_filename))
    aj = json.load(open(os.path.join(directory, tj["architecture"])))
    arch = Architecture(aj, source_data_format="channels_last", data_format=args.data_format, device="cuda:%d" % local, dtype=args.dtype)
    base = tj["base_tfrecords_directory"] if os.path.isabs(tj["base_tfrecords_directory"]) else os.path.join(directory, tj["base_tfrecords_directory"])
    if "training" not in tj["modes"]:
        raise Exception("No training mode found.")
    st = tfrecords.read_settings(base, "training")
    spp = st["source_samples_per_pixel_list"][0]
    random.seed(0)
    tuples, _ = source_index_tuples(st["number_of_sources_per_example"], tj["number_of_source_index_tuples"], arch.number_of_sources_per_target)
    B, tile = tj["batch_size"], st["tiles_height_width"]
    trainer = Trainer(arch, tj, B, tile, tile, world_size=world)
    model_dir = aj["model_directory"] if os.path.isabs(aj["model_directory"]) else os.path.join(directory, aj["model_directory"])
    step = 0
    latest = tf_checkpoint.latest_checkpoint(model_dir) if os.path.isdir(model_dir) else None
    if latest:
        step = tf_checkpoint.load_variables(arch, latest)["global_step"]
        print("restored %s (global_step %d)" % (latest, step))
    usage = DataAugmentationUsage.from_training_json(tj)
    gen = torch.Generator().manual_seed(1234 + rank)
    for epoch in range(args.train_epochs):
        total, count = 0.0, 0
        for feats, labels in batches(base, "training", arch, B, spp, tuples, rank, world):
            feats, labels = DataAugmentation.apply(feats, labels, DataAugmentation.draw(B, gen), usage)
            trainer.program.set_inputs(feats, labels)
            loss = trainer.step()
            step += 1
            if step % 50 == 0:
                total, count = total + float(loss), count + 1
        if rank == 0:
            print("epoch %d: global_step %d, loss %.5f" % (epoch + 1, step, total / max(count, 1)))
            tf_checkpoint.save_variables(arch, model_dir, global_step=step)
        if args.validate and (epoch + 1) % args.validation_interval == 0 and "validation" in tj["modes"]:
            losses = []
            for feats, labels in batches(base, "validation", arch, B, spp, tuples, rank, world):
                trainer.program.set_inputs(feats, labels)
                trainer.program.zero_grads()
                trainer.program.forward()
                losses.append(float(trainer.program.loss_buf))
            if rank == 0 and losses:
                print("epoch %d: validation loss %.5f over %d batches" % (epoch + 1, float(np.mean(losses)), len(losses)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(parser().parse_args())
