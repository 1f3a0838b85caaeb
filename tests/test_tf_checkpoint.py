"""TensorFlow checkpoint bundles without TensorFlow (deepdenoiser_amd/tf_checkpoint.py).  PARITY UNPINNED: no file
written by TensorFlow is available; the reader is checked against an index table assembled BY HAND in this file from the published
leveldb-table / tensor_bundle.proto definitions (independent of the module's writer), and the writer against the reader."""
import os
import struct
import types

import numpy as np
import pytest
import torch

from deepdenoiser_amd import tf_checkpoint as TC
from deepdenoiser_amd.engine import ParamStore
from deepdenoiser_amd.tfrecords import crc32c, masked_crc32c


def _vi(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _block(entries, restart_interval):
    """entries: sorted [(key, value)] -> block bytes with prefix compression, written out longhand."""
    buf, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        if i % restart_interval == 0:
            restarts.append(len(buf))
            shared = 0
        else:
            shared = len(os.path.commonprefix([k, last]))
        buf += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    return bytes(buf) + b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))


def _with_trailer(block, ctype=0):
    return block + bytes([ctype]) + struct.pack("<I", masked_crc32c(block + bytes([ctype])))


def _entry(dtype, shape, offset, size, crc):
    shape_msg = b"".join(b"\x12" + _vi(len(b"\x08" + _vi(d))) + b"\x08" + _vi(d) for d in shape)       # dim = field 2 { size = field 1 }
    out = b"\x08" + _vi(dtype) + b"\x12" + _vi(len(shape_msg)) + shape_msg
    if offset:
        out += b"\x20" + _vi(offset)
    out += b"\x28" + _vi(size) + b"\x35" + struct.pack("<I", crc)
    return out


def _hand_made_bundle(tmp_path, crc_of=masked_crc32c, first_block_type=0):
    a = np.arange(24, dtype="<f4").reshape(2, 3, 4) - 7.5
    b = np.array([1, -2, 3], dtype="<f4")
    step = np.array(1234, dtype="<i8")
    bf = np.array([0x3F80, 0xC000], dtype="<u2")                       # bfloat16 1.0, -2.0
    tensors = [(b"global_step", step, 9), (b"scope/conv2d/bias", b, 1), (b"scope/conv2d/kernel", a, 1), (b"scope/half", bf, 14)]
    data, entries = b"", []
    for name, arr, dt in tensors:
        raw = arr.tobytes()
        entries.append((name, _entry(dt, arr.shape, len(data), len(raw), crc_of(raw))))
        data += raw
    header = b"\x08\x01" + b"\x1a\x02\x08\x01"                          # num_shards 1, version { producer 1 }
    # two data blocks: the second starts inside the run of keys sharing the prefix "scope/conv2d/"
    blocks = [[(b"", header)] + entries[:2], entries[2:]]
    table, index_entries = b"", []
    for blk in blocks:
        bb = _block(blk, restart_interval=2)
        index_entries.append((blk[-1][0], _vi(len(table)) + _vi(len(bb))))
        table += _with_trailer(bb, first_block_type if not table else 0)
    meta = _block([], 1)
    meta_handle = _vi(len(table)) + _vi(len(meta))
    table += _with_trailer(meta)
    ib = _block(index_entries, 1)
    index_handle = _vi(len(table)) + _vi(len(ib))
    table += _with_trailer(ib)
    footer = meta_handle + index_handle
    table += footer + b"\x00" * (40 - len(footer)) + bytes.fromhex("57fb808b247547db")
    prefix = str(tmp_path / "model.ckpt-1234")
    with open(prefix + ".index", "wb") as f:
        f.write(table)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(data)
    return prefix, a, b


def test_reader_on_a_hand_assembled_bundle(tmp_path):
    prefix, a, b = _hand_made_bundle(tmp_path)
    num_shards, index = TC.read_index(prefix)
    assert num_shards == 1
    assert list(index) == ["global_step", "scope/conv2d/bias", "scope/conv2d/kernel", "scope/half"]
    assert index["scope/conv2d/kernel"].shape == (2, 3, 4) and index["scope/conv2d/kernel"].offset == 8 + 12
    assert index["global_step"].shape == () and index["global_step"].dtype == 9
    ck = TC.read_checkpoint(prefix)
    assert ck["global_step"].dtype == np.int64 and int(ck["global_step"]) == 1234
    assert np.array_equal(ck["scope/conv2d/kernel"], a) and np.array_equal(ck["scope/conv2d/bias"], b)
    assert ck["scope/half"].dtype == np.float32 and ck["scope/half"].tolist() == [1.0, -2.0]          # bfloat16 widened
    assert list(TC.read_checkpoint(prefix, names={"scope/conv2d/bias"})) == ["scope/conv2d/bias"]


def test_reader_accepts_unmasked_tensor_checksums_and_rejects_wrong_ones(tmp_path):
    prefix, a, _ = _hand_made_bundle(tmp_path, crc_of=crc32c)
    assert np.array_equal(TC.read_checkpoint(prefix)["scope/conv2d/kernel"], a)
    prefix, _, _ = _hand_made_bundle(tmp_path, crc_of=lambda raw: 12345)
    with pytest.raises(TC.CheckpointError, match="checksum"):
        TC.read_checkpoint(prefix)
    assert "scope/conv2d/kernel" in TC.read_checkpoint(prefix, verify=False)


def test_writer_layout_and_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {"global_step": np.array(7, dtype=np.int64), "beta1_power": np.array(0.9 ** 7, dtype=np.float32)}
    for i in range(300):                                                  # enough entries for several 4 KB index blocks
        tensors["reused_core_architecture/conv2d_%d/kernel" % i] = rng.standard_normal((3, 3, 4, 5)).astype(np.float32)
        tensors["reused_core_architecture/conv2d_%d/bias" % i] = rng.standard_normal(5).astype(np.float32)
    tensors["flags/mask"] = np.array([True, False, True])
    tensors["wide"] = rng.standard_normal((2, 2))                        # float64
    prefix = str(tmp_path / "sub" / "model.ckpt-7")
    TC.write_checkpoint(prefix, tensors)
    raw = open(prefix + ".index", "rb").read()
    assert raw[-8:] == bytes.fromhex("57fb808b247547db") and len(raw) > 3 * 4096
    # first data block starts with the header entry: shared 0, key length 0, value length 6, value = num_shards 1 + version{producer 1}
    assert raw[:9] == bytes([0, 0, 6, 0x08, 0x01, 0x1a, 0x02, 0x08, 0x01])
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(np.asarray(v).nbytes for v in tensors.values())
    back = TC.read_checkpoint(prefix)
    assert list(back) == sorted(tensors, key=lambda s: s.encode())
    for k, v in tensors.items():
        assert back[k].dtype == np.asarray(v).dtype and np.array_equal(back[k], v), k
    # a flipped bit anywhere in the index or the data is noticed
    corrupt = bytearray(raw)
    corrupt[100] ^= 0x10
    open(prefix + ".index", "wb").write(bytes(corrupt))
    with pytest.raises(TC.CheckpointError):
        TC.read_checkpoint(prefix)
    open(prefix + ".index", "wb").write(raw)
    with open(prefix + ".data-00000-of-00001", "r+b") as f:
        f.seek(1000)
        byte = f.read(1)
        f.seek(1000)
        f.write(bytes([byte[0] ^ 1]))
    with pytest.raises(TC.CheckpointError, match="checksum"):
        TC.read_checkpoint(prefix)


def test_not_a_bundle_and_compressed_blocks_are_rejected(tmp_path):
    p = str(tmp_path / "x")
    open(p + ".index", "wb").write(b"\x00" * 64)
    with pytest.raises(TC.CheckpointError, match="magic"):
        TC.read_index(p)
    open(p + ".index", "wb").write(b"\x00" * 10)
    with pytest.raises(TC.CheckpointError, match="footer"):
        TC.read_index(p)
    prefix, _, _ = _hand_made_bundle(tmp_path, first_block_type=1)          # type byte 1 = snappy, checksum consistent
    with pytest.raises(TC.CheckpointError, match="compressed"):
        TC.read_index(prefix)


def _toy_arch(seed):
    ps = ParamStore()
    ps.get("embedding/feature_flags_embedding_matrix", (17, 8), 17, 8)
    ps.get("reused_core_architecture/conv2d/kernel", (3, 3, 16, 16), 144, 144)
    ps.get("reused_core_architecture/conv2d/bias", (16,))
    ps.get("reused_core_architecture/conv2d_transpose/kernel", (2, 2, 16, 24), 64, 96)
    ps.get("reused_core_architecture/conv2d_transpose/bias", (16,))
    ps.get("reused_compose_scales/conv2d_5/kernel", (1, 1, 24, 1), 24, 1)      # 24 values: exercises the 16-byte arena padding
    ps.get("reused_compose_scales/conv2d_5/bias", (1,))
    ps.finalize("cpu", seed=seed)
    return types.SimpleNamespace(params=ps)


def test_parameter_arena_round_trip_with_adam_state(tmp_path):
    a = _toy_arch(seed=3)
    g = torch.Generator().manual_seed(0)
    a.params.m.copy_(torch.randn(a.params.m.shape, generator=g))
    a.params.v.copy_(torch.rand(a.params.v.shape, generator=g))
    a.adam_step = 41
    model_dir = str(tmp_path / "model")
    assert TC.latest_checkpoint(model_dir) is None
    prefix = TC.save_variables(a, model_dir, global_step=41)
    assert os.path.basename(prefix) == "model.ckpt-41" and TC.latest_checkpoint(model_dir) == prefix
    names = list(TC.read_index(prefix)[1])
    assert "reused_core_architecture/conv2d/kernel/Adam_1" in names and "beta2_power" in names and "global_step" in names
    assert TC.read_index(prefix)[1]["reused_core_architecture/conv2d_transpose/kernel"].shape == (2, 2, 16, 24)

    b = _toy_arch(seed=4)
    assert not torch.equal(a.params.values, b.params.values)
    info = TC.load_variables(b, prefix)
    assert info == {"global_step": 41, "adam_step": 41, "missing": [], "unused": []}
    assert b.adam_step == 41
    for arena in ("values", "m", "v"):
        for p in a.params.params:           # the padding between parameters is not part of the checkpoint
            sl = slice(p.offset, p.offset + p.size)
            assert torch.equal(getattr(a.params, arena)[sl], getattr(b.params, arena)[sl]), (arena, p.name)

    # weights only (what Prediction.py restores): slots untouched
    c = _toy_arch(seed=5)
    TC.save_variables(a, model_dir, global_step=50, save_optimizer=False)
    assert TC.latest_checkpoint(model_dir).endswith("model.ckpt-50")
    assert open(os.path.join(model_dir, "checkpoint")).read().count("all_model_checkpoint_paths") == 2
    info = TC.load_variables(c, TC.latest_checkpoint(model_dir))
    assert info["adam_step"] is None and float(c.params.m.abs().max()) == 0.0
    assert torch.equal(c.params.value(c.params.params[1]), a.params.value(a.params.params[1]))


def test_load_variables_reports_missing_and_mismatched_variables(tmp_path):
    a = _toy_arch(seed=3)
    prefix = TC.save_variables(a, str(tmp_path), global_step=1, save_optimizer=False)
    ck = dict(TC.read_checkpoint(prefix))
    del ck["reused_compose_scales/conv2d_5/bias"]
    ck["extra/variable"] = np.zeros(2, dtype=np.float32)
    TC.write_checkpoint(prefix, ck)
    with pytest.raises(TC.CheckpointError, match="lacks 1 model variable"):
        TC.load_variables(_toy_arch(seed=1), prefix)
    info = TC.load_variables(_toy_arch(seed=1), prefix, strict=False)
    assert info["missing"] == ["reused_compose_scales/conv2d_5/bias"] and info["unused"] == ["extra/variable"]
    ck["reused_compose_scales/conv2d_5/bias"] = np.zeros(3, dtype=np.float32)
    TC.write_checkpoint(prefix, ck)
    with pytest.raises(TC.CheckpointError, match="shape"):
        TC.load_variables(_toy_arch(seed=1), prefix)
    unbuilt = types.SimpleNamespace(params=ParamStore())
    with pytest.raises(RuntimeError, match="before load_variables"):
        TC.load_variables(unbuilt, prefix)


@pytest.mark.parametrize("t", [0, 1, 41, 700, 5000, 60000, 300000])
def test_adam_step_survives_the_checkpoint_at_any_length_of_run(tmp_path, t):
    """tf.train.AdamOptimizer stores beta ** (t + 1) in float32: a fresh model saves beta (never 1.0, TF divides by 1 - beta1_power),
    0.9 ** t underflows near t = 1000 and 0.999 ** t near t = 1e5 -- the step must come back exactly in every regime."""
    a = _toy_arch(seed=3)
    a.adam_step = t
    prefix = TC.save_variables(a, str(tmp_path), global_step=t)
    ck = TC.read_checkpoint(prefix)
    assert float(ck["beta1_power"]) == float(np.float32(0.9 ** (t + 1))) and float(ck["beta2_power"]) == float(np.float32(0.999 ** (t + 1)))
    if t == 0:
        assert float(ck["beta1_power"]) == float(np.float32(0.9))
    b = _toy_arch(seed=4)
    info = TC.load_variables(b, prefix)
    assert info["adam_step"] == t and b.adam_step == t


def test_adam_step_of_a_tensorflow_written_checkpoint_is_not_off_by_one(tmp_path):
    """After 7 updates TF holds beta1_power = 0.9 ** 8; global_step (counted by the Estimator) may lag or be absent."""
    a = _toy_arch(seed=3)
    prefix = TC.save_variables(a, str(tmp_path), global_step=7)
    ck = dict(TC.read_checkpoint(prefix))
    ck["beta1_power"] = np.array(0.9 ** 8, dtype=np.float32)
    ck["beta2_power"] = np.array(0.999 ** 8, dtype=np.float32)
    del ck["global_step"]
    TC.write_checkpoint(prefix, ck)
    assert TC.load_variables(_toy_arch(seed=1), prefix)["adam_step"] == 7
    # both powers underflowed and no global_step: refuse instead of restarting the bias correction on warm moments
    ck["beta1_power"] = np.array(0.0, dtype=np.float32)
    ck["beta2_power"] = np.array(0.0, dtype=np.float32)
    TC.write_checkpoint(prefix, ck)
    with pytest.raises(TC.CheckpointError, match="Adam step is unknown"):
        TC.load_variables(_toy_arch(seed=1), prefix)
    # beta2_power alone (beta1_power underflowed, ~2000 steps)
    ck["beta2_power"] = np.array(0.999 ** 2001, dtype=np.float32)
    TC.write_checkpoint(prefix, ck)
    assert abs(TC.load_variables(_toy_arch(seed=1), prefix)["adam_step"] - 2000) <= 2
