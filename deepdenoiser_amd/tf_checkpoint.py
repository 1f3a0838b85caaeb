"""TensorFlow checkpoints (V2 "tensor bundle": <prefix>.index + <prefix>.data-00000-of-00001) without TensorFlow: the weights the
reference's Estimator saves into --model_directory (TensorFlow/Training.py:1209-1232 builds the Estimator, :700-702 the optimizer whose
slots land in the same files) and that Prediction.py:497-505 restores.  SURVEY 8f rank 4.

PARITY UNPINNED: TensorFlow is not installable here and the reference ships no checkpoint, so nothing below has been checked against
a file written by TensorFlow.  The formats are restated from their published definitions:

  <prefix>.index   a leveldb-format sorted string table (tensorflow/core/lib/io/table*.cc, format.cc; a port of leveldb's table/):
                     data blocks | metaindex block | index block | 48-byte footer
                     block   = entries, uint32 restart offsets, uint32 restart count; then 1 type byte (0 = uncompressed, 1 = snappy)
                               and uint32 masked crc32c(block + type byte)
                     entry   = varint shared-key-bytes, varint unshared-key-bytes, varint value-bytes, key suffix, value
                     footer  = BlockHandle(metaindex), BlockHandle(index) as varint (offset, size) pairs, zero padding to 40 bytes,
                               uint64 magic 0xdb4775248b80fb57
                   key ""           -> BundleHeaderProto  { num_shards = 1; endianness = 2; version = 3 { producer = 1 } }
                   key <var name>   -> BundleEntryProto   { dtype = 1; shape = 2 { dim = 2 { size = 1 } }; shard_id = 3; offset = 4;
                                                            size = 5; crc32c = 6 (fixed32, masked); slices = 7 }
                   (tensorflow/core/protobuf/tensor_bundle.proto; tensorflow/core/util/tensor_bundle/tensor_bundle.cc)
  <prefix>.data-<shard>-of-<shards>   the tensors' little-endian bytes back to back at (offset, size)
  checkpoint       text CheckpointState:  model_checkpoint_path: "model.ckpt-1234"  (+ all_model_checkpoint_paths lines)

Variable names are the TF creation names the parameter arena already uses: `<scope>/conv2d_7/kernel`, `.../bias`,
`embedding/feature_flags_embedding_matrix`; Adam adds `<var>/Adam` (m), `<var>/Adam_1` (v), `beta1_power`, `beta2_power`; the
Estimator adds `global_step` (int64).  Kernels are HWIO / transpose-conv [kh,kw,C_out,C_in] exactly as in the arena, so import and
export are copies.  Snappy-compressed index blocks (never written by TensorFlow's BundleWriter) and partitioned variables (`slices`)
are rejected with an error rather than guessed at.
"""
import collections
import os
import struct

import numpy as np

from .tfrecords import _enc_varint, _fields, _ld, _varint, crc32c

_MAGIC = 0xdb4775248b80fb57
_FOOTER = 48
_MASK_DELTA = 0xA282EAD8

# DataType enum (tensorflow/core/framework/types.proto) -> numpy; DT_BFLOAT16 is widened to float32 on read
_DT_FLOAT, _DT_DOUBLE, _DT_INT32, _DT_INT64, _DT_BOOL, _DT_BFLOAT16, _DT_HALF = 1, 2, 3, 9, 10, 14, 19
_NP_OF_DT = {_DT_FLOAT: np.dtype("<f4"), _DT_DOUBLE: np.dtype("<f8"), _DT_INT32: np.dtype("<i4"), _DT_INT64: np.dtype("<i8"),
             _DT_BOOL: np.dtype("bool"), _DT_HALF: np.dtype("<f2"), _DT_BFLOAT16: np.dtype("<u2")}
_DT_OF_NP = {np.dtype("float32"): _DT_FLOAT, np.dtype("float64"): _DT_DOUBLE, np.dtype("int32"): _DT_INT32, np.dtype("int64"): _DT_INT64,
             np.dtype("bool"): _DT_BOOL, np.dtype("float16"): _DT_HALF}


class CheckpointError(IOError):
    pass


def _mask(c):
    return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xFFFFFFFF


Entry = collections.namedtuple("Entry", "dtype shape shard_id offset size crc32c")


# ---------------------------------------------------------------------------------------------------- table reader
def _read_block(buf, offset, size, verify):
    if offset + size + 5 > len(buf):
        raise CheckpointError("index block [%d, +%d) runs past the end of the file" % (offset, size))
    contents = bytes(buf[offset:offset + size])
    ctype = buf[offset + size]
    (crc,) = struct.unpack_from("<I", buf, offset + size + 1)
    if verify and _mask(crc32c(contents + bytes([ctype]))) != crc:
        raise CheckpointError("index block at %d: checksum mismatch" % offset)
    if ctype != 0:
        raise CheckpointError("index block at %d is compressed (type %d); only uncompressed tables are supported" % (offset, ctype))
    return contents


def _block_entries(block):
    if len(block) < 4:
        raise CheckpointError("index block too short")
    (n_restarts,) = struct.unpack_from("<I", block, len(block) - 4)
    limit = len(block) - 4 * (n_restarts + 1)
    if limit < 0:
        raise CheckpointError("index block restart array is larger than the block")
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _varint(block, pos)
        unshared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        if shared > len(key) or pos + unshared + vlen > limit:
            raise CheckpointError("malformed index block entry")
        key = key[:shared] + block[pos:pos + unshared]
        pos += unshared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _parse_entry(value):
    dtype, shape, shard, offset, size, crc, sliced = 0, [], 0, 0, 0, None, False
    for num, wt, v in _fields(memoryview(value)):
        if num == 1 and wt == 0:
            dtype = v
        elif num == 2 and wt == 2:
            for snum, swt, dim in _fields(v):
                if snum == 2 and swt == 2:
                    d = 0
                    for dnum, dwt, dv in _fields(dim):
                        if dnum == 1 and dwt == 0:
                            d = dv
                    shape.append(d)
        elif num == 3 and wt == 0:
            shard = v
        elif num == 4 and wt == 0:
            offset = v
        elif num == 5 and wt == 0:
            size = v
        elif num == 6 and wt == 5:
            (crc,) = struct.unpack("<I", bytes(v))
        elif num == 7:
            sliced = True
    if sliced:
        raise CheckpointError("partitioned variables (BundleEntryProto.slices) are not supported")
    return Entry(dtype, tuple(shape), shard, offset, size, crc)


def read_index(prefix, verify=True):
    """-> (num_shards, OrderedDict name -> Entry) of <prefix>.index, in key order."""
    path = prefix + ".index"
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < _FOOTER:
        raise CheckpointError("%s: shorter than a table footer" % path)
    footer = buf[-_FOOTER:]
    if struct.unpack("<Q", footer[-8:])[0] != _MAGIC:
        raise CheckpointError("%s: not a TensorFlow tensor-bundle index (bad magic number)" % path)
    pos = 0
    _, pos = _varint(footer, pos)      # metaindex handle (unused)
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)
    isize, pos = _varint(footer, pos)
    entries, num_shards = collections.OrderedDict(), 1
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, hp = _varint(handle, 0)
        bsize, hp = _varint(handle, hp)
        for key, value in _block_entries(_read_block(buf, boff, bsize, verify)):
            if key == b"":
                for num, wt, v in _fields(memoryview(value)):
                    if num == 1 and wt == 0:
                        num_shards = v
                    elif num == 2 and wt == 0 and v != 0:
                        raise CheckpointError("%s: big-endian bundle" % path)
                continue
            entries[key.decode("utf-8")] = _parse_entry(value)
    return num_shards, entries


def read_checkpoint(prefix, names=None, verify=True):
    """-> OrderedDict variable name -> numpy array (bfloat16 widened to float32), in key order.  `names`: only these."""
    num_shards, entries = read_index(prefix, verify)
    shards, out = {}, collections.OrderedDict()
    for name, e in entries.items():
        if names is not None and name not in names:
            continue
        if e.dtype not in _NP_OF_DT:
            continue                                # strings, resources, ...: nothing this model stores
        if e.shard_id not in shards:
            path = "%s.data-%05d-of-%05d" % (prefix, e.shard_id, num_shards)
            shards[e.shard_id] = (np.memmap(path, dtype=np.uint8, mode="r") if os.path.getsize(path) else np.zeros(0, dtype=np.uint8))     # (an empty file cannot be mapped)
        data = shards[e.shard_id]
        dt = _NP_OF_DT[e.dtype]
        count = int(np.prod(e.shape, dtype=np.int64)) if e.shape else 1
        if e.size != count * dt.itemsize or e.offset + e.size > data.shape[0]:
            raise CheckpointError("%s: size %d at offset %d does not match shape %s of %s" % (name, e.size, e.offset, e.shape, dt))
        raw = bytes(data[e.offset:e.offset + e.size])
        if verify and e.crc32c is not None and e.crc32c not in (_mask(crc32c(raw)), crc32c(raw)):
            raise CheckpointError("%s: tensor checksum mismatch" % name)
        arr = np.frombuffer(raw, dtype=dt).reshape(e.shape)
        if e.dtype == _DT_BFLOAT16:
            arr = (arr.astype(np.uint32) << 16).view(np.float32)
        out[name] = arr
    return out


# ---------------------------------------------------------------------------------------------------- table writer
class _BlockBuilder:
    def __init__(self, restart_interval):
        self.interval, self.buf, self.restarts, self.count, self.last = restart_interval, bytearray(), [0], 0, b""

    def add(self, key, value):
        shared = 0
        if self.count % self.interval == 0:
            if self.count:
                self.restarts.append(len(self.buf))
        else:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _enc_varint(shared) + _enc_varint(len(key) - shared) + _enc_varint(len(value)) + key[shared:] + value
        self.last, self.count = key, self.count + 1

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _vint_field(num, v):
    return _enc_varint(num << 3) + _enc_varint(v)


def _entry_proto(dtype, shape, offset, size, crc):
    dims = b"".join(_ld(2, _vint_field(1, d)) for d in shape)
    out = _vint_field(1, dtype) + _ld(2, dims)
    if offset:
        out += _vint_field(4, offset)                # proto3: zero-valued scalars (shard_id 0, offset 0) are not serialized
    if size:
        out += _vint_field(5, size)
    return out + _enc_varint((6 << 3) | 5) + struct.pack("<I", crc)


def write_checkpoint(prefix, tensors, block_size=4096):
    """Write {name: array} as a one-shard bundle.  float32/float64/int32/int64/bool/float16 arrays; names are sorted bytewise."""
    items = sorted((name.encode("utf-8"), np.asarray(arr, order="C")) for name, arr in tensors.items())     # (ascontiguousarray makes 0-d arrays 1-d)
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    table, index = bytearray(), _BlockBuilder(1)

    def emit(block_bytes):
        off = len(table)
        table.extend(block_bytes + b"\x00" + struct.pack("<I", _mask(crc32c(block_bytes + b"\x00"))))
        return _enc_varint(off) + _enc_varint(len(block_bytes))

    header = _vint_field(1, 1) + _ld(3, _vint_field(1, 1))        # num_shards 1, little endian (default 0), version.producer 1
    block = _BlockBuilder(16)
    block.add(b"", header)
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as data:
        for key, arr in items:
            if arr.dtype not in _DT_OF_NP:
                raise CheckpointError("%s: dtype %s cannot be stored" % (key.decode(), arr.dtype))
            raw = arr.astype(arr.dtype.newbyteorder("<"), copy=False).tobytes()
            data.write(raw)
            if len(block.buf) >= block_size:
                index.add(block.last, emit(block.finish()))
                block = _BlockBuilder(16)
            block.add(key, _entry_proto(_DT_OF_NP[arr.dtype], arr.shape, offset, len(raw), _mask(crc32c(raw))))
            offset += len(raw)
    index.add(block.last, emit(block.finish()))
    meta = emit(_BlockBuilder(1).finish())
    idx = emit(index.finish())
    footer = meta + idx
    table.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC))
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(table))


# ---------------------------------------------------------------------------------------------------- CheckpointState file
def latest_checkpoint(model_directory):
    """Prefix named by <model_directory>/checkpoint (what tf.train.latest_checkpoint returns), or None."""
    state = os.path.join(model_directory, "checkpoint")
    if not os.path.exists(state):
        return None
    with open(state) as f:
        for line in f:
            if line.startswith("model_checkpoint_path:"):
                name = line.split(":", 1)[1].strip().strip('"')
                return name if os.path.isabs(name) else os.path.join(model_directory, name)
    return None


def _update_state(model_directory, name):
    state = os.path.join(model_directory, "checkpoint")
    older = []
    if os.path.exists(state):
        with open(state) as f:
            older = [ln.split(":", 1)[1].strip().strip('"') for ln in f if ln.startswith("all_model_checkpoint_paths:")]
    paths = [p for p in older if p != name] + [name]
    with open(state, "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % name)
        for p in paths:
            f.write('all_model_checkpoint_paths: "%s"\n' % p)


# ---------------------------------------------------------------------------------------------------- the parameter arena
def _adam_step_from_powers(ck, beta1, beta2, global_step):
    """Number of Adam updates t behind a checkpoint.  TF stores beta1_power = beta1 ** (t + 1) and beta2_power = beta2 ** (t + 1)
    (float32): beta1_power resolves t exactly while it is a normal number (t < ~800 for 0.9), beta2_power up to ~8e4 steps;
    beyond that (both underflowed or too flat to invert) the step is the Estimator's global_step, saved in the same file.  The step is
    never silently reset: bias correction restarting on warm moments would cut the effective learning rate to ~0.3x."""
    b1p = float(ck["beta1_power"].reshape(-1)[0])
    b2p = float(ck["beta2_power"].reshape(-1)[0]) if "beta2_power" in ck else 0.0
    tiny = float(np.finfo(np.float32).tiny)
    if tiny * 1e3 < b1p < 1.0:
        return max(int(round(np.log(b1p) / np.log(beta1))) - 1, 0)
    if b1p == 1.0:      # a step-0 file written by the first revisions of this module (they stored beta ** t; TensorFlow stores beta ** (t + 1))
        return 0
    if b1p > 1.0:
        raise CheckpointError("beta1_power = %r is not a power of beta1 = %r" % (b1p, beta1))
    if tiny * 1e3 < b2p < 1.0:
        t = int(round(np.log(b2p) / np.log(beta2))) - 1
        if global_step is not None and abs(global_step - t) <= max(2, int(2e-3 * t)):
            return int(global_step)      # float32 beta2_power resolves t to ~1e-3 relative; global_step is exact when it agrees
        return max(t, 0)
    if global_step is not None:
        return int(global_step)
    raise CheckpointError("beta1_power / beta2_power have underflowed and the checkpoint has no global_step: the Adam step is unknown")


def load_variables(arch, prefix, load_optimizer=True, strict=True, beta1=0.9, beta2=0.999):
    """Copy a checkpoint into the architecture's parameter arena (built programs first: Architecture.program(...) or
    Predictor.prepare(...) create the variables).  With load_optimizer the Adam slots and step count are restored as well when the
    checkpoint has them.  strict: every variable of the model must be in the checkpoint with the same shape.
    -> {'global_step', 'adam_step', 'missing': [...], 'unused': [...]}"""
    import torch
    ps = arch.params
    if ps.values is None:
        raise RuntimeError("parameters are created when the first program is built: call Architecture.program(...) / "
                           "Predictor.prepare(H, W) before load_variables()")
    ck = read_checkpoint(prefix)
    missing, used = [], set()
    host = {"values": ps.values.cpu(), "m": ps.m.cpu(), "v": ps.v.cpu()}
    have_slots = load_optimizer and all((p.name + "/Adam") in ck and (p.name + "/Adam_1") in ck for p in ps.params)
    for p in ps.params:
        if p.name not in ck:
            missing.append(p.name)
            continue
        for arena, key in (("values", p.name),) + ((("m", p.name + "/Adam"), ("v", p.name + "/Adam_1")) if have_slots else ()):
            arr = ck[key]
            if tuple(arr.shape) != tuple(p.shape):
                raise CheckpointError("%s: checkpoint shape %s, model shape %s" % (key, tuple(arr.shape), tuple(p.shape)))
            host[arena][p.offset:p.offset + p.size] = torch.from_numpy(np.array(arr, dtype=np.float32)).reshape(-1)
            used.add(key)
    if missing and strict:
        raise CheckpointError("%s lacks %d model variable(s): %s" % (prefix, len(missing), ", ".join(missing[:5])))
    ps.values.copy_(host["values"])
    info = {"global_step": int(ck["global_step"].reshape(-1)[0]) if "global_step" in ck else None, "adam_step": None, "missing": missing}
    if have_slots:
        ps.m.copy_(host["m"])
        ps.v.copy_(host["v"])
        if "beta1_power" in ck:
            used.add("beta1_power")
            used.add("beta2_power")
            info["adam_step"] = _adam_step_from_powers(ck, beta1, beta2, info["global_step"])
            arch.adam_step = info["adam_step"]
    info["unused"] = [k for k in ck if k not in used and k != "global_step"]
    return info


def save_variables(arch, model_directory, global_step, save_optimizer=True, beta1=0.9, beta2=0.999, basename="model.ckpt"):
    """Write <model_directory>/<basename>-<global_step>.{index,data-00000-of-00001} and update the `checkpoint` state file, with the
    names, shapes and dtypes an Estimator training the reference graph produces.  -> the checkpoint prefix."""
    ps = arch.params
    if ps.values is None:
        raise RuntimeError("no parameters yet: build a program first")
    values, m, v = ps.values.cpu().numpy(), ps.m.cpu().numpy(), ps.v.cpu().numpy()
    tensors = {"global_step": np.array(global_step, dtype=np.int64)}
    for p in ps.params:
        sl = slice(p.offset, p.offset + p.size)
        tensors[p.name] = values[sl].reshape(p.shape)
        if save_optimizer:
            tensors[p.name + "/Adam"] = m[sl].reshape(p.shape)
            tensors[p.name + "/Adam_1"] = v[sl].reshape(p.shape)
    if save_optimizer:
        # tf.train.AdamOptimizer creates beta{1,2}_power with initial value beta and multiplies by beta in _finish(): after t
        # updates the variables hold beta ** (t + 1) (a fresh model saves beta, never 1.0 -- TF computes 1 - beta1_power)
        t = getattr(arch, "adam_step", 0)
        tensors["beta1_power"] = np.array(beta1 ** (t + 1), dtype=np.float32)
        tensors["beta2_power"] = np.array(beta2 ** (t + 1), dtype=np.float32)
    name = "%s-%d" % (basename, global_step)
    prefix = os.path.join(model_directory, name)
    write_checkpoint(prefix, tensors)
    _update_state(model_directory, name)
    return prefix
