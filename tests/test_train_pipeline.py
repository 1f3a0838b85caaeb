"""The training CLI's input pipeline (deepdenoiser_amd/train.py TileStream; Training.py:826-843): per-epoch file shuffle, decoder threads,
20 x batch shuffle buffer, whole rounds of `world` mini-batches so that every rank runs the same number of steps.  Host only."""
import json
import os
import random

import numpy as np
import torch

from deepdenoiser_amd import configs, tfrecords
from deepdenoiser_amd.architecture import Architecture
from deepdenoiser_amd.naming import Naming
from deepdenoiser_amd.train import TileStream, evaluation_jsons

T, SPP = 8, 4


def _dataset(base, arch, n_files, per_file):
    os.makedirs(os.path.join(base, "training"))
    passes = {f.name: f.number_of_channels for f in arch.feature_predictions + arch.auxiliary_features if f.load_data}
    targets = [f.name for f in arch.feature_predictions if f.load_data and f.is_target]
    first = next(iter(passes))
    uid = 0
    for n in range(n_files):
        records = []
        for _ in range(per_file):
            feats = {}
            for name, ch in passes.items():
                img = np.full((T, T, ch), float(uid) if name == first else 0.25, dtype=np.float32)      # the example's id is readable from its first pass
                feats[Naming.source_feature_name(name, samples_per_pixel=SPP, index=0)] = img.tobytes()
            for name in targets:
                feats[Naming.target_feature_name(name)] = np.full((T, T, passes[name]), float(uid), dtype=np.float32).tobytes()
            records.append(tfrecords.serialize_example(feats))
            uid += 1
        tfrecords.write_records(os.path.join(base, "training", "training_%d.tfrecords.gz" % n), records)
    return first, uid


def _ids(stream, first):
    key = Naming.source_feature_name(first, index=0)
    return [[int(v) for v in feats[key][:, 0, 0, 0].tolist()] for feats, _ in stream]


def test_tile_stream_shuffles_shards_evenly_and_is_reproducible(tmp_path):
    aj = configs.architecture(filters=(16, 24), convs=1, flag_mode="NONE")
    arch = Architecture(aj, device="cpu")
    base = str(tmp_path / "data")
    first, n = _dataset(base, arch, n_files=5, per_file=7)            # 35 examples
    B, world = 4, 2

    def run(rank, seed, threads):
        return _ids(TileStream(os.path.join(base, "training"), "training", arch, B, T, SPP, [[0]], rank, world, rng=random.Random(seed),
                               threads=threads, pinned=False), first)
    r0, r1 = run(0, 7, 3), run(1, 7, 1)
    # 35 examples = 8 mini-batches of 4 -> 4 rounds of 2: every rank runs 4 steps; together they hold 32 distinct examples
    assert len(r0) == len(r1) == 4
    flat = [i for b in r0 + r1 for i in b]
    assert len(set(flat)) == len(flat) == 32 and set(flat) <= set(range(n))
    # shuffled: not the file order, and tiles of one file do not stay together
    in_order = [list(range(k, k + B)) for k in range(0, 32, B)]
    assert r0 != in_order[0::2]
    # reproducible for a seed whatever the number of decoder threads; a different seed gives a different order
    assert run(0, 7, 1) == r0 and run(1, 7, 4) == r1
    assert run(0, 8, 3) != r0
    # validation order: no rng -> file and record order
    plain = _ids(TileStream(os.path.join(base, "training"), "training", arch, B, T, SPP, [[0]], 0, 1, rng=None, threads=2, pinned=False), first)
    assert plain == [list(range(k, k + B)) for k in range(0, 32, B)]


def test_evaluation_jsons_lists_the_validation_sets(tmp_path):
    for name in ("validation_4.json", "validation_16.json", "validation_statistics.json", "training_4.json", "validation.txt"):
        (tmp_path / name).write_text("{}")
    assert evaluation_jsons(str(tmp_path), "validation") == ["validation_16.json", "validation_4.json"]


def test_validation_stream_pads_its_last_round_and_training_counts_what_it_drops(tmp_path):
    """35 examples, batch 4, two ranks: training drops the 3 left after 4 rounds and says so; a validation stream (pad_last) fills a fifth round
    with repeats of its own three examples, and weighs the last mini-batch by its real examples, so every example counts exactly once."""
    aj = configs.architecture(filters=(16, 24), convs=1, flag_mode="NONE")
    arch = Architecture(aj, device="cpu")
    base = str(tmp_path / "data")
    first, n = _dataset(base, arch, n_files=5, per_file=7)
    B, world = 4, 2
    tr = TileStream(os.path.join(base, "training"), "training", arch, B, T, SPP, [[0]], 0, world, rng=None, threads=2, pinned=False)
    assert len(_ids(tr, first)) == 4 and tr.dropped == 3 and tr.padded == 0
    seen, real, last_round = [], [], []
    for rank in range(world):
        va = TileStream(os.path.join(base, "training"), "training", arch, B, T, SPP, [[0]], rank, world, rng=None, threads=3, pinned=False, pad_last=True)
        ids = _ids(va, first)
        assert len(ids) == 5 and va.padded == 5 and va.dropped == 0
        seen += [i for b in ids for i in b]
        # (round 6) the weights run_validation gives the mini-batches: B for whole rounds, the REAL examples of the last one
        real += [i for b in ids[:-1] for i in b] + ids[-1][:va.real_in_last]
        last_round += ids[-1]
    assert set(seen) == set(range(n)) and len(seen) == 40
    assert sorted(real) == list(range(n)), "every example must carry weight exactly once"
    assert set(last_round) == set(range(32, 35)), "the last round repeats its own examples, not the start of the epoch"
    # a set smaller than one round still yields one (it yielded nothing before)
    small = str(tmp_path / "small")
    first, n = _dataset(small, arch, n_files=1, per_file=3)
    va = TileStream(os.path.join(small, "training"), "training", arch, B, T, SPP, [[0]], 1, world, rng=None, threads=1, pinned=False, pad_last=True)
    ids = _ids(va, first)
    assert len(ids) == 1 and va.padded == 5 and set(ids[0]) <= set(range(n))


def test_decoder_threads_cannot_starve_the_file_the_consumer_waits_for(tmp_path):
    """More files than look-ahead permits and one decoder thread per permit pair: every run must terminate and keep file order (the permit is taken
    before the file index, under one lock)."""
    aj = configs.architecture(filters=(16, 24), convs=1, flag_mode="NONE")
    arch = Architecture(aj, device="cpu")
    base = str(tmp_path / "data")
    first, n = _dataset(base, arch, n_files=24, per_file=2)
    for threads in (1, 2, 8):
        st = TileStream(os.path.join(base, "training"), "training", arch, 4, T, SPP, [[0]], 0, 1, rng=None, threads=threads, pinned=False)
        flat = [i for b in _ids(st, first) for i in b]
        assert flat == list(range(n))
