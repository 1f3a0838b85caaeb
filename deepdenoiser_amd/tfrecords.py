"""Reader (and a small writer) for the reference's training data files: GZIP-compressed TFRecord files of `tf.train.Example` protos whose
features are raw little-endian float32 tiles (TensorFlow/TFRecordsCreator.py:135-158 writes them, :221-252 frames and compresses them;
TensorFlow/Training.py:502-524 parses them, :826-843 opens them).  No TensorFlow: the two wire formats are restated here.

TFRecord framing (tensorflow/core/lib/io/record_writer.cc):  uint64 length | uint32 masked_crc32c(length) | data | uint32 masked_crc32c(data)
tf.train.Example (tensorflow/core/example/example.proto, feature.proto):
    Example { Features features = 1; }   Features { map<string, Feature> feature = 1; }   Feature { oneof { BytesList bytes_list = 1; ... } }
    BytesList { repeated bytes value = 1; }
Only bytes_list features occur in this data set; other kinds are skipped.
"""
import gzip
import json
import os
import struct

import numpy as np

from .naming import Naming

# ---------------------------------------------------------------------------------------------------- CRC-32C (Castagnoli), masked
_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ _POLY if _c & 1 else _c >> 1
    _TABLE.append(_c)


_native = None     # dd_crc32c of libdd_hip.so (host code, slicing-by-8) once loaded; False if the library is not built


def _crc32c_python(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def crc32c(data):
    """CRC-32C of a bytes-like object.  Multi-megabyte payloads (tiles, checkpoint tensors) go through the library's host routine;
    the table loop above is the definition and the fallback for short inputs or a missing library."""
    global _native
    if len(data) >= 4096 and _native is not False:
        if _native is None:
            try:
                from . import _lib
                _native = _lib.load().dd_crc32c
            except (RuntimeError, OSError):
                _native = False
        if _native:
            import ctypes as C
            out = C.c_uint32()
            raw = bytes(data) if not isinstance(data, bytes) else data
            if _native(raw, len(raw), 0, C.byref(out)) != 0:
                raise RuntimeError("dd_crc32c failed")
            return out.value
    return _crc32c_python(data)


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------------- TFRecord framing
def _open(path, mode):
    return gzip.open(path, mode) if path.endswith(".gz") else open(path, mode)


def read_records(path, verify_payload_crc=True):
    """Yield the raw records of a (.gz) TFRecord file.  Length header and payload are CRC-checked, as tf.data.TFRecordDataset does
    (crc32c() hands payloads >= 4 KiB to the native dd_crc32c); verify_payload_crc=False skips the payload check when the native
    library is unavailable and the pure-Python CRC of multi-megabyte tiles is too slow."""
    with _open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) != 12:
                raise IOError("%s: truncated record header" % path)
            (length,), (hcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if masked_crc32c(head[:8]) != hcrc:
                raise IOError("%s: corrupt record length" % path)
            data = f.read(length)
            tail = f.read(4)
            if len(data) != length or len(tail) != 4:
                raise IOError("%s: truncated record" % path)
            if verify_payload_crc and masked_crc32c(data) != struct.unpack("<I", tail)[0]:
                raise IOError("%s: corrupt record payload" % path)
            yield data


def write_records(path, records):
    with _open(path, "wb") as f:
        for data in records:
            head = struct.pack("<Q", len(data))
            f.write(head + struct.pack("<I", masked_crc32c(head)) + data + struct.pack("<I", masked_crc32c(data)))


# ---------------------------------------------------------------------------------------------------- protobuf wire format (subset)
def _varint(buf, pos):
    shift, val = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _fields(buf):
    """(field number, wire type, value) of one message; length-delimited values are memoryview slices."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 2:
            ln, pos = _varint(buf, pos)
            yield num, wt, buf[pos:pos + ln]
            pos += ln
        elif wt == 0:
            v, pos = _varint(buf, pos)
            yield num, wt, v
        elif wt == 1:
            yield num, wt, buf[pos:pos + 8]
            pos += 8
        elif wt == 5:
            yield num, wt, buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)


def parse_example(record):
    """tf.train.Example bytes -> {feature name: bytes of the first bytes_list value} (what tf.parse_single_example with
    FixedLenFeature([], tf.string) returns, Training.py:507-515)."""
    out = {}
    buf = memoryview(record)
    for num, wt, features in _fields(buf):
        if num != 1 or wt != 2:
            continue
        for fnum, fwt, entry in _fields(features):            # map<string, Feature> entries
            if fnum != 1 or fwt != 2:
                continue
            name, feature = None, None
            for enum, ewt, val in _fields(entry):
                if enum == 1 and ewt == 2:
                    name = bytes(val).decode("utf-8")
                elif enum == 2 and ewt == 2:
                    feature = val
            if name is None or feature is None:
                continue
            for knum, kwt, blist in _fields(feature):
                if knum == 1 and kwt == 2:                      # bytes_list
                    for vnum, vwt, value in _fields(blist):
                        if vnum == 1 and vwt == 2:
                            out[name] = bytes(value)
                            break
    return out


def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(num, payload):
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def serialize_example(features):
    """{name: bytes} -> tf.train.Example bytes (TFRecordsCreator._bytes_feature + tf.train.Example, TFRecordsCreator.py:221-231).
    Map entries are written in sorted key order (protobuf map order is unspecified; readers must not depend on it)."""
    body = b""
    for name in sorted(features):
        feature = _ld(1, _ld(1, features[name]))                # Feature{bytes_list{value}}
        body += _ld(1, _ld(1, name.encode("utf-8")) + _ld(2, feature))
    return _ld(1, body)


# ---------------------------------------------------------------------------------------------------- the data set
def read_settings(base_directory, mode, samples_per_pixel=None):
    """The side-car JSON next to the tfrecords directory (TFRecordsCreator.py:166-178): tiles_height_width,
    number_of_sources_per_example, source_samples_per_pixel_list."""
    name = mode if samples_per_pixel is None else "%s_%d" % (mode, samples_per_pixel)
    with open(os.path.join(base_directory, name + ".json"), encoding="utf-8") as f:
        return json.load(f)


def decode_example(parsed, passes, tile, source_samples_per_pixel_list, required_indices, targets):
    """Training.py:518-524: raw bytes -> float32 [tile, tile, channels] arrays.
    passes: {pass name: channels}; targets: names of the passes that are also targets.
    Returns (sources {spp: {index: {name: array}}}, targets {name: array})."""
    src, tgt = {}, {}
    for spp in source_samples_per_pixel_list:
        src[spp] = {}
        for index in required_indices:
            d = src[spp][index] = {}
            for name, ch in passes.items():
                raw = parsed[Naming.source_feature_name(name, samples_per_pixel=spp, index=index)]
                d[name] = np.frombuffer(raw, dtype="<f4").reshape(tile, tile, ch)
    for name in targets:
        tgt[name] = np.frombuffer(parsed[Naming.target_feature_name(name)], dtype="<f4").reshape(tile, tile, passes[name])
    return src, tgt


def list_files(tfrecords_directory, mode):
    """<mode>_<n>.tfrecords.gz in numeric order (TFRecordsCreator.py:206-210, :233-247)."""
    pre = mode + "_"
    names = [n for n in os.listdir(tfrecords_directory) if n.startswith(pre) and n.endswith(".tfrecords.gz")]
    return [os.path.join(tfrecords_directory, n) for n in sorted(names, key=lambda n: int(n[len(pre):-len(".tfrecords.gz")]))]
