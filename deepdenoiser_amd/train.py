"""`python -m deepdenoiser_amd.train training.json` -- the reference's `python Training.py training.json` (TensorFlow/Training.py:33-61 argument
set, :944-1289 main) on the MI355X path: TFRecord tiles -> device augmentation -> Trainer.step -> TensorFlow-format checkpoints in the
architecture's model_directory.  One process per GPU under torch.distributed.run shards every mini-batch over the ranks.

Input pipeline (Training.py:826-843): the file list is shuffled per epoch, records are decoded by `--threads` background threads, examples pass a
shuffle buffer of 20 x batch_size, mini-batches are stacked into pinned host memory `prefetch` (5) deep and uploaded on a copy stream while the
previous step runs.  Every random choice comes from an RNG seeded by (seed, epoch) that all ranks share, so the ranks see the SAME sequence of
mini-batches and each takes every world-th one; an epoch is cut to a multiple of `world` mini-batches, so every rank runs the same number of
steps (and collectives).  `--validate` is a validation-only run; otherwise every `--validation_interval` epochs are followed by a validation
pass over base/validation/<spp>/ for every validation*.json (Training.py:1230-1284).

Not reproduced (out of scope): the Estimator's evaluation / TensorBoard summaries; validation reports the mean loss."""
import argparse
import json
import multiprocessing
import os
import queue
import random
import threading
import time

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
    p.add_argument("--threads", type=int, default=multiprocessing.cpu_count() + 1, help="Number of threads to use (host-side decoding)")
    p.add_argument("--train_epochs", type=int, default=10000, help="Number of epochs to train.")
    p.add_argument("--validation_interval", type=int, default=1, help="Number of epochs after which a validation is made.")
    p.add_argument("--data_format", type=str, default="channels_first", choices=["channels_first", "channels_last"],
                   help="Accepted for compatibility: the MI355X path is NHWC-native, both values give the same results.")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"], help="storage type of activations (accumulation is fp32)")
    p.add_argument("--seed", type=int, default=0, help="seed of the file / example shuffles and of the source index tuples (shared by all ranks)")
    return p


def evaluation_jsons(base, mode):
    """<mode>*.json next to the tfrecords directories, statistics files excluded (Training.py:916-927)."""
    out = []
    for name in sorted(os.listdir(base)):
        stem, ext = os.path.splitext(name)
        if stem.startswith(mode) and ext == ".json" and "statistics" not in stem and os.path.isfile(os.path.join(base, name)):
            out.append(name)
    return out


class TileStream:
    """Mini-batches of one epoch for this rank: an iterator of (features, labels) dictionaries of pinned host tensors.

    directory: the <mode>[/<spp>] directory of .tfrecords.gz files; `rng`: random.Random shared by all ranks (None: file and record order,
    as validation reads them); `threads` decoder threads each own a file at a time and hand decoded examples to the batcher IN FILE ORDER
    (the shuffled order), so the sequence of examples does not depend on thread timing."""

    def __init__(self, directory, mode, arch, batch, tile, spp, index_tuples, rank=0, world=1, rng=None, threads=4, prefetch=5, pinned=True,
                 pad_last=False):
        """pad_last (validation): the examples left over after the last whole round of `world` mini-batches are evaluated too -- the round is
        filled up by repeating examples from the start of the epoch (`padded` counts them; the reference's dataset.batch() evaluates the
        remainder as a smaller batch, a static launch program cannot).  Without it they are dropped and counted in `dropped`."""
        self.arch, self.batch, self.tile, self.spp, self.tuples = arch, batch, tile, spp, index_tuples
        self.rank, self.world, self.rng, self.prefetch, self.pinned = rank, world, rng, prefetch, pinned
        self.pad_last, self.padded, self.dropped, self._const = pad_last, 0, 0, {}
        self.real_in_last = batch          # real (not repeated) examples in this rank's LAST mini-batch of the epoch (pad_last)
        self.passes = {f.name: f.number_of_channels for f in arch.feature_predictions + arch.auxiliary_features if f.load_data}
        self.targets = [f.name for f in arch.feature_predictions if f.load_data and f.is_target]
        self.required = sorted({i for t in index_tuples for i in t})
        self.files = tfrecords.list_files(directory, mode)
        if rng is not None:
            rng.shuffle(self.files)                                     # Training.py:826-828 (files.shuffle)
        self.threads = max(1, min(int(threads), len(self.files), 32))
        self.decoded = 0

    def _decode_file(self, path):
        out = []
        for rec in tfrecords.read_records(path):
            src, tgt = tfrecords.decode_example(tfrecords.parse_example(rec), self.passes, self.tile, [self.spp], self.required, self.targets)
            for tup in self.tuples:                                     # one training example per (source index tuple, target): Training.py:544-549
                out.append((src[self.spp][tup[0]], tgt))
        return out

    def _examples(self):
        """Decoded examples in (shuffled) file order; files are decoded `threads` at a time, at most 2 x threads files ahead."""
        slots = [queue.Queue(maxsize=1) for _ in self.files]
        ahead = threading.Semaphore(2 * self.threads)
        nxt = iter(range(len(self.files)))
        lock = threading.Lock()

        def worker():
            while True:
                # the look-ahead permit comes BEFORE the index, both under one lock: a permit holder then always owns the lowest outstanding
                # index (taken the other way round the other workers could park 2 x threads later files, take every permit, and leave the
                # worker holding the index the consumer waits for without one: ADVICE r4)
                with lock:
                    ahead.acquire()
                    i = next(nxt, None)
                if i is None:
                    ahead.release()
                    return
                try:
                    slots[i].put(self._decode_file(self.files[i]))
                except BaseException as e:                               # surfaces in the consumer
                    slots[i].put(e)
        pool = [threading.Thread(target=worker, daemon=True) for _ in range(self.threads)]
        for t in pool:
            t.start()
        for i in range(len(self.files)):
            got = slots[i].get()
            ahead.release()
            if isinstance(got, BaseException):
                raise got
            yield from got

    def _shuffled(self):
        """Training.py:836-838: a shuffle buffer of 20 x batch_size examples."""
        if self.rng is None:
            yield from self._examples()
            return
        size, buf = 20 * self.batch, []
        for ex in self._examples():
            if len(buf) < size:
                buf.append(ex)
                continue
            j = self.rng.randrange(size)
            buf[j], ex = ex, buf[j]
            yield ex
        self.rng.shuffle(buf)
        yield from buf

    def _stack(self, group):
        feats, labels = {}, {}
        B, tile = self.batch, self.tile

        def host(shape):      # (allocated pinned: empty(...).pin_memory() allocates pageable memory and copies its garbage into a second buffer)
            return torch.empty(shape, dtype=torch.float32, pin_memory=bool(self.pinned and torch.cuda.is_available()))
        for k, ch in self.passes.items():
            t = host((B, tile, tile, ch))
            np.stack([s[k] for s, _ in group], out=t.numpy())
            feats[Naming.source_feature_name(k, index=0)] = t
        for k in self.targets:
            t = host((B, tile, tile, self.passes[k]))
            np.stack([tg[k] for _, tg in group], out=t.numpy())
            labels[Naming.target_feature_name(k)] = t
        for f in self.arch.feature_predictions + self.arch.auxiliary_features:      # generated passes (Training.py:531-538)
            if not f.load_data:
                value = 1.0 if f.feature_prediction_type == "COLOR" else 0.5
                key = (f.number_of_channels, value)
                if key not in self._const:      # constant passes: one (pinned) tensor for the whole stream, not one per mini-batch
                    self._const[key] = host((B, tile, tile, f.number_of_channels)).fill_(value)
                feats[Naming.source_feature_name(f.name, index=0)] = self._const[key]
                if f.is_target:
                    labels[Naming.target_feature_name(f.name)] = self._const[key]
        return feats, labels

    def __iter__(self):
        """This rank's mini-batches; a background thread keeps `prefetch` of them stacked ahead of the consumer."""
        ready = queue.Queue(maxsize=self.prefetch)
        END = object()

        def producer():
            try:
                group, buf, head = [], [], []
                need = self.batch * self.world

                def feed(ex):
                    nonlocal group, buf
                    buf.append(ex)
                    if len(buf) == self.batch:
                        group.append(buf)
                        buf = []
                        if len(group) == self.world:                     # only whole rounds of `world` mini-batches: every rank gets one
                            ready.put(self._stack(group[self.rank]))
                            group = []
                for ex in self._shuffled():
                    self.decoded += 1
                    if self.pad_last and len(head) < need:
                        head.append(ex)
                    feed(ex)
                left = len(group) * self.batch + len(buf)
                if left and self.pad_last and head:
                    # fill the last round by cycling over ITS OWN examples (not the start of the epoch: those would be counted twice), and
                    # tell the consumer how many of this rank's last mini-batch are real: it weighs that mini-batch's mean by the count, a
                    # mini-batch of repeats only by zero (the fixed-size programme cannot evaluate a smaller remainder batch)
                    tail = [ex for b in group for ex in b] + list(buf)
                    self.real_in_last = max(0, min(self.batch, left - self.rank * self.batch))
                    for k in range(need - left):
                        feed(tail[k % len(tail)])
                    self.padded = need - left
                else:
                    self.dropped = left
                ready.put(END)
            except BaseException as e:
                ready.put(e)
        threading.Thread(target=producer, daemon=True).start()
        while True:
            item = ready.get()
            if item is END:
                return
            if isinstance(item, BaseException):
                raise item
            yield item


class Uploader:
    """Host -> device copies of the next mini-batch on a side stream while the current step runs; the consumer's stream waits on the copy's
    event, the copy waits until the step that read the previous contents of the staging tensors has been enqueued."""

    def __init__(self, device):
        self.device = device
        self.stream = torch.cuda.Stream(device=device)

    def stage(self, feats, labels):
        done = torch.cuda.Event()
        with torch.cuda.stream(self.stream):
            df = {k: v.to(self.device, non_blocking=True) for k, v in feats.items()}
            dl = {k: v.to(self.device, non_blocking=True) for k, v in labels.items()}
            done.record(self.stream)
        return df, dl, done, (feats, labels)                            # (the pinned sources stay alive until the copy has run)


def run_validation(trainer, arch, tj, base, B, rank, world, threads):
    """Training.py:1230-1250 / :1264-1284: every validation*.json, tiles of base/validation/<spp>/, no augmentation, mean loss."""
    import torch.distributed as dist
    results = []
    for name in evaluation_jsons(base, "validation"):
        st = json.load(open(os.path.join(base, name), encoding="utf-8"))
        spp = st["source_samples_per_pixel_list"][0]
        vdir = os.path.join(base, "validation", str(spp))
        if not os.path.isdir(vdir):
            vdir = os.path.join(base, "validation")                      # data sets written without group_by_samples_per_pixel
        tuples, _ = source_index_tuples(st["number_of_sources_per_example"], tj["number_of_source_index_tuples"], arch.number_of_sources_per_target,
                                        rng=random.Random(0))
        if st["tiles_height_width"] != trainer.program.H:
            print("validation set %s: tiles of %d pixels, the program was built for %d -- skipped" % (name, st["tiles_height_width"], trainer.program.H))
            continue
        stream = TileStream(vdir, "validation", arch, B, st["tiles_height_width"], spp, tuples, rank, world, rng=None, threads=threads, pad_last=True)
        total = torch.zeros(2, dtype=torch.float64, device=arch.device)
        losses = []
        for feats, labels in stream:
            trainer.program.set_inputs({k: v.to(arch.device) for k, v in feats.items()}, {k: v.to(arch.device) for k, v in labels.items()})
            trainer.program.zero_grads()
            trainer.program.forward()
            losses.append(trainer.program.loss_buf.double().sum())
        # every example counts once: a mini-batch's mean is weighed by its REAL examples (the last round of the epoch is filled with repeats)
        for i, l in enumerate(losses):
            w = stream.real_in_last if (stream.padded and i == len(losses) - 1) else B
            total[0] += l * w
            total[1] += w
        if world > 1:
            dist.all_reduce(total)
        if float(total[1]) > 0:
            results.append((os.path.splitext(name)[0], float(total[0] / total[1]), int(total[1])))
            if stream.padded and rank == 0:
                print("validation set %s: the last round was filled up with %d repeated example(s) (%d decoded)" % (name, stream.padded, stream.decoded))
    return results


def main(args):
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dist = None
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
    B, tile = tj["batch_size"], st["tiles_height_width"]
    trainer = Trainer(arch, tj, B, tile, tile, world_size=world)
    model_dir = aj["model_directory"] if os.path.isabs(aj["model_directory"]) else os.path.join(directory, aj["model_directory"])
    step = 0
    latest = tf_checkpoint.latest_checkpoint(model_dir) if os.path.isdir(model_dir) else None
    if latest:
        step = tf_checkpoint.load_variables(arch, latest)["global_step"]
        print("restored %s (global_step %d)" % (latest, step))
    validate = "validation" in tj["modes"]

    def report_validation(tag):
        for name, loss, n in run_validation(trainer, arch, tj, base, B, rank, world, args.threads):
            if rank == 0:
                print("%s: %s loss %.5f over %d batches" % (tag, name, loss, n))

    if args.validate:                                                     # Training.py:1230: a validation-only run
        report_validation("validation")
        if dist is not None:
            dist.destroy_process_group()
        return

    usage = DataAugmentationUsage.from_training_json(tj)
    gen = torch.Generator().manual_seed(1234 + rank)
    uploader = Uploader(arch.device)
    main_stream = torch.cuda.current_stream()
    for epoch in range(args.train_epochs):
        rng = random.Random(args.seed * 1000003 + epoch)                 # shared by the ranks: same files, same shuffle, same index tuples
        tuples, _ = source_index_tuples(st["number_of_sources_per_example"], tj["number_of_source_index_tuples"], arch.number_of_sources_per_target,
                                        rng=rng)                         # redrawn every epoch (Training.py:1258)
        stream = TileStream(os.path.join(base, "training"), "training", arch, B, tile, spp, tuples, rank, world, rng=rng, threads=args.threads)
        total, count, t0, steps0 = 0.0, 0, time.time(), step
        staged = None
        it = iter(stream)
        nxt = next(it, None)
        if nxt is not None:
            staged = uploader.stage(*nxt)
        while staged is not None:
            feats, labels, done, keep = staged
            main_stream.wait_event(done)
            for v in list(feats.values()) + list(labels.values()):       # allocated on the copy stream, read on the launch stream
                v.record_stream(main_stream)
            feats, labels = DataAugmentation.apply(feats, labels, DataAugmentation.draw(B, gen), usage)
            trainer.program.set_inputs(feats, labels)
            nxt = next(it, None)                                         # the next mini-batch uploads while this step runs
            staged = uploader.stage(*nxt) if nxt is not None else None
            loss = trainer.step()
            step += 1
            if step % 50 == 0 or staged is None:
                total, count = total + float(loss), count + 1
        if dist is not None:
            dist.barrier()
        if rank == 0:
            dt = max(time.time() - t0, 1e-9)
            print("epoch %d: global_step %d, loss %.5f (%d steps, %.1f tiles/s, %d examples decoded)" % (
                epoch + 1, step, total / max(count, 1), step - steps0, (step - steps0) * B * world / dt, stream.decoded))
            tf_checkpoint.save_variables(arch, model_dir, global_step=step)
        if dist is not None:
            dist.barrier()                                               # nobody runs ahead of the checkpoint
        if validate and (epoch + 1) % args.validation_interval == 0:
            report_validation("epoch %d" % (epoch + 1))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(parser().parse_args())
