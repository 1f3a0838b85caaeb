"""TFRecord / tf.train.Example wire formats against published check values and hand-assembled bytes."""
import struct

import numpy as np
import pytest

from deepdenoiser_amd import tfrecords as R
from deepdenoiser_amd.naming import Naming


def test_crc32c_check_values():
    assert R.crc32c(b"123456789") == 0xE3069283                       # the CRC-32C (Castagnoli) check value, RFC 3720 appendix B.4
    assert R.crc32c(b"\x00" * 32) == 0x8A9136AA                       # RFC 3720 B.4: 32 bytes of zeros
    assert R.crc32c(b"\xff" * 32) == 0x62A8AB43                       # RFC 3720 B.4: 32 bytes of ones
    c = R.crc32c(b"123456789")
    assert R.masked_crc32c(b"123456789") == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def test_example_wire_format_hand_assembled():
    # Example{features{feature{key:"a" value{bytes_list{value:"xy"}}}}} written out by hand from the protobuf encoding rules
    feature = bytes([0x0A, 0x04, 0x0A, 0x02]) + b"xy"                 # Feature: field 1 (bytes_list) len 4 { field 1 (value) len 2 "xy" }
    entry = bytes([0x0A, 0x01]) + b"a" + bytes([0x12, len(feature)]) + feature
    features = bytes([0x0A, len(entry)]) + entry
    example = bytes([0x0A, len(features)]) + features
    assert R.serialize_example({"a": b"xy"}) == example
    assert R.parse_example(example) == {"a": b"xy"}
    # unknown fields and other feature kinds are skipped: add a float_list feature "f" and an unknown varint field 7 at the top level
    fl = bytes([0x12, 0x06, 0x0A, 0x04]) + struct.pack("<f", 1.5)
    entry2 = bytes([0x0A, 0x01]) + b"f" + bytes([0x12, len(fl)]) + fl
    features2 = features + bytes([0x0A, len(entry2)]) + entry2
    example2 = bytes([0x0A, len(features2)]) + features2 + bytes([0x38, 0x05])
    assert R.parse_example(example2) == {"a": b"xy"}


@pytest.mark.parametrize("ext", [".tfrecords", ".tfrecords.gz"])
def test_record_framing_round_trip_and_corruption(tmp_path, ext):
    recs = [b"", b"abc", bytes(range(256)) * 3]
    path = str(tmp_path / ("training_0" + ext))
    R.write_records(path, recs)
    assert list(R.read_records(path, verify_payload_crc=True)) == recs
    if ext == ".tfrecords":
        raw = bytearray(open(path, "rb").read())
        assert struct.unpack("<Q", raw[:8])[0] == 0 and len(raw) == 3 * 16 + 0 + 3 + 768
        raw[16 + 12] ^= 1                                              # flip a payload bit of the second record
        open(path, "wb").write(raw)
        with pytest.raises(IOError):
            list(R.read_records(path, verify_payload_crc=True))
        raw[16 + 12] ^= 1
        raw[16] ^= 1                                                   # corrupt its length field
        open(path, "wb").write(raw)
        with pytest.raises(IOError):
            list(R.read_records(path))


def test_tile_data_set_round_trip(tmp_path):
    """A miniature data set in the reference's layout: <dir>/training/training_<n>.tfrecords.gz + <dir>/training.json."""
    import json
    import os
    T = 8
    passes = {"Diffuse Color": 3, "Depth": 1, "Normal": 3}
    rng = np.random.default_rng(0)
    os.makedirs(tmp_path / "training")
    json.dump({"tiles_height_width": T, "number_of_sources_per_example": 2, "source_samples_per_pixel_list": [16]}, open(tmp_path / "training.json", "w"))
    examples = []
    for n in range(3):
        feats, arrays = {}, {}
        for name, ch in passes.items():
            for idx in range(2):
                a = rng.standard_normal((T, T, ch)).astype(np.float32)
                arrays[("s", idx, name)] = a
                feats[Naming.source_feature_name(name, samples_per_pixel=16, index=idx)] = a.tobytes()
        a = rng.standard_normal((T, T, 3)).astype(np.float32)
        arrays[("t", "Diffuse Color")] = a
        feats[Naming.target_feature_name("Diffuse Color")] = a.tobytes()
        examples.append((feats, arrays))
    R.write_records(str(tmp_path / "training" / "training_0.tfrecords.gz"), [R.serialize_example(f) for f, _ in examples[:2]])
    R.write_records(str(tmp_path / "training" / "training_10.tfrecords.gz"), [R.serialize_example(examples[2][0])])
    R.write_records(str(tmp_path / "training" / "training_2.tfrecords.gz"), [])
    st = R.read_settings(str(tmp_path), "training")
    files = R.list_files(str(tmp_path / "training"), "training")
    assert [f.split("_")[-1] for f in files] == ["0.tfrecords.gz", "2.tfrecords.gz", "10.tfrecords.gz"]     # numeric, not lexicographic
    got = [R.parse_example(r) for f in files for r in R.read_records(f, verify_payload_crc=True)]
    assert len(got) == 3
    for parsed, (_, arrays) in zip(got, examples):
        src, tgt = R.decode_example(parsed, passes, st["tiles_height_width"], st["source_samples_per_pixel_list"], [0, 1], ["Diffuse Color"])
        for name in passes:
            for idx in range(2):
                assert np.array_equal(src[16][idx][name], arrays[("s", idx, name)])
        assert np.array_equal(tgt["Diffuse Color"], arrays[("t", "Diffuse Color")])
