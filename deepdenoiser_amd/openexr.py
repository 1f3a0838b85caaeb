"""OpenEXR frames without OpenCV / the OpenEXR library: the render-pass images either side of the inference path
(TensorFlow/OpenEXRDirectory.py:54-151 loads one .exr per render pass through cv2.imdecode and hands RGB float32 to Prediction.py:223-252;
SURVEY 8f rank 4).

PARITY UNPINNED: neither OpenCV nor the OpenEXR library is installed and the reference ships no image, so nothing here has been checked
against a file written by Blender.  The container format is restated from the published "OpenEXR File Layout" document:

    int32 magic 20000630 | int32 version (low byte 2; bit 0x200 tiled, 0x800 deep, 0x1000 multi-part)
    header    = attributes (name\\0 type\\0 int32 size, value) ... \\0
                channels (chlist: name\\0 int32 type{0 uint,1 half,2 float} uint8 pLinear 3 reserved int32 xSampling int32 ySampling ... \\0),
                compression (uint8), dataWindow / displayWindow (box2i), lineOrder, pixelAspectRatio, screenWindowCenter, screenWindowWidth
    uint64 offset per chunk, chunks of 1 (none, RLE, ZIPS) or 16 (ZIP) scan lines:
    chunk     = int32 y | int32 size | data;   data = per scan line, per channel in name order, one row of that channel's samples
    ZIP / ZIPS / RLE transform the chunk bytes first: split into even and odd bytes (first half, second half), then byte deltas
    (t[i] - t[i-1] + 128), then deflate (or run-length code); a chunk that would not shrink is stored raw.

Supported: single-part scan-line files, compression NONE / RLE / ZIPS / ZIP (Blender's default), UINT / HALF / FLOAT channels without
subsampling.  Tiled, deep, multi-part files and the PIZ / PXR24 / B44 / DWA codecs raise ExrError naming what was found.
"""
import os
import struct
import zlib

import numpy as np

MAGIC = 20000630
NO_COMPRESSION, RLE_COMPRESSION, ZIPS_COMPRESSION, ZIP_COMPRESSION = 0, 1, 2, 3
_CODEC_NAMES = {0: "NONE", 1: "RLE", 2: "ZIPS", 3: "ZIP", 4: "PIZ", 5: "PXR24", 6: "B44", 7: "B44A", 8: "DWAA", 9: "DWAB"}
_LINES = {NO_COMPRESSION: 1, RLE_COMPRESSION: 1, ZIPS_COMPRESSION: 1, ZIP_COMPRESSION: 16}
_PIXEL = {0: np.dtype("<u4"), 1: np.dtype("<f2"), 2: np.dtype("<f4")}
_PIXEL_CODE = {np.dtype("uint32"): 0, np.dtype("float16"): 1, np.dtype("float32"): 2}


class ExrError(IOError):
    pass


# ---------------------------------------------------------------------------------------------------- chunk codecs
def _undo_predictor_and_interleave(t):
    """bytes after inflate / run-length decoding -> original chunk bytes"""
    d = np.frombuffer(t, dtype=np.uint8).astype(np.int64)
    if d.size == 0:
        return b""
    d[1:] -= 128
    s = (np.cumsum(d) & 0xFF).astype(np.uint8)
    half = (s.size + 1) // 2
    out = np.empty(s.size, dtype=np.uint8)
    out[0::2] = s[:half]
    out[1::2] = s[half:]
    return out.tobytes()


def _interleave_and_predict(raw):
    """original chunk bytes -> bytes handed to deflate / the run-length coder"""
    b = np.frombuffer(raw, dtype=np.uint8)
    if b.size == 0:
        return b""
    t = np.concatenate([b[0::2], b[1::2]]).astype(np.int64)
    d = t.copy()
    d[1:] = t[1:] - t[:-1] + 128
    return (d & 0xFF).astype(np.uint8).tobytes()


def _rle_decode(data, expected):
    out, pos, n = bytearray(), 0, len(data)
    while pos < n:
        count = data[pos] - 256 if data[pos] > 127 else data[pos]
        pos += 1
        if count < 0:
            out += data[pos:pos - count]
            pos -= count
        else:
            out += bytes([data[pos]]) * (count + 1)
            pos += 1
    if len(out) != expected:
        raise ExrError("run-length coded chunk expands to %d bytes, expected %d" % (len(out), expected))
    return bytes(out)


def _rle_encode(data):
    out, i, n = bytearray(), 0, len(data)
    while i < n:
        j = i + 1
        while j < n and data[j] == data[i] and j - i < 128:
            j += 1
        if j - i >= 3:                                     # a run
            out += bytes([j - i - 1, data[i]])
            i = j
            continue
        j = i                                               # literals up to the next run of 3 (or 127 bytes)
        while j < n and j - i < 127 and not (j + 2 < n and data[j] == data[j + 1] == data[j + 2]):
            j += 1
        out += bytes([256 - (j - i)]) + data[i:j]
        i = j
    return bytes(out)


def _decode_chunk(data, compression, expected):
    if len(data) == expected:                               # stored raw because coding did not shrink it (or NO_COMPRESSION)
        return data
    if compression in (ZIP_COMPRESSION, ZIPS_COMPRESSION):
        t = zlib.decompress(data)
    elif compression == RLE_COMPRESSION:
        t = _rle_decode(data, expected)
    else:
        raise ExrError("chunk of %d bytes where %d are expected" % (len(data), expected))
    if len(t) != expected:
        raise ExrError("chunk expands to %d bytes, expected %d" % (len(t), expected))
    return _undo_predictor_and_interleave(t)


# ---------------------------------------------------------------------------------------------------- header
def _cstr(buf, pos):
    end = buf.index(b"\0", pos)
    return buf[pos:end].decode("latin-1"), end + 1


def _parse_header(buf):
    if len(buf) < 8:
        raise ExrError("not an OpenEXR file (too short)")
    magic, version = struct.unpack_from("<ii", buf, 0)
    if magic != MAGIC:
        raise ExrError("not an OpenEXR file (magic %d)" % magic)
    if version & 0xFF != 2:
        raise ExrError("OpenEXR format version %d is not supported" % (version & 0xFF))
    for bit, what in ((0x200, "tiled"), (0x800, "deep"), (0x1000, "multi-part")):
        if version & bit:
            raise ExrError("%s OpenEXR files are not supported (single-part scan-line files only)" % what)
    pos, attrs = 8, {}
    while True:
        name, pos = _cstr(buf, pos)
        if not name:
            break
        kind, pos = _cstr(buf, pos)
        (size,) = struct.unpack_from("<i", buf, pos)
        pos += 4
        attrs[name] = (kind, buf[pos:pos + size])
        pos += size
    for need in ("channels", "compression", "dataWindow"):
        if need not in attrs:
            raise ExrError("header lacks the %s attribute" % need)
    channels, cp, cl = [], 0, attrs["channels"][1]
    while cl[cp] != 0:
        cname, cp = _cstr(cl, cp)
        ptype, _plinear, xs, ys = struct.unpack_from("<iB3xii", cl, cp)
        cp += 16
        if ptype not in _PIXEL:
            raise ExrError("channel %s: unknown pixel type %d" % (cname, ptype))
        if xs != 1 or ys != 1:
            raise ExrError("channel %s is subsampled (%d x %d); not supported" % (cname, xs, ys))
        channels.append((cname, ptype))
    compression = attrs["compression"][1][0]
    if compression not in _LINES:
        raise ExrError("compression %s is not supported (NONE, RLE, ZIPS, ZIP are)" % _CODEC_NAMES.get(compression, compression))
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    return {"channels": channels, "compression": compression, "data_window": (x0, y0, x1, y1), "attributes": attrs}, pos


def read_exr(path):
    """-> ({channel name: [H,W] array (float32 for HALF/FLOAT, uint32 for UINT)}, header dict) of the file's data window."""
    with open(path, "rb") as f:
        buf = f.read()
    try:
        head, pos = _parse_header(buf)
    except (ValueError, struct.error, IndexError) as e:
        raise ExrError("%s: truncated or malformed header (%s)" % (path, e))
    x0, y0, x1, y1 = head["data_window"]
    W, H = x1 - x0 + 1, y1 - y0 + 1
    chans = head["channels"]                                  # stored sorted by name; the chunk layout follows this order
    lines = _LINES[head["compression"]]
    n_chunks = (H + lines - 1) // lines
    if pos + 8 * n_chunks > len(buf):
        raise ExrError("%s: truncated offset table" % path)
    offsets = struct.unpack_from("<%dQ" % n_chunks, buf, pos)
    row_bytes = sum(W * _PIXEL[t].itemsize for _, t in chans)
    planes = {name: np.empty((H, W), dtype=_PIXEL[t]) for name, t in chans}
    for off in offsets:
        if off + 8 > len(buf):
            raise ExrError("%s: chunk offset %d past the end of the file" % (path, off))
        y, size = struct.unpack_from("<ii", buf, off)
        r0 = y - y0
        nl = min(lines, H - r0)
        if r0 < 0 or nl <= 0 or off + 8 + size > len(buf):
            raise ExrError("%s: bad chunk (y %d, %d bytes)" % (path, y, size))
        raw = _decode_chunk(buf[off + 8:off + 8 + size], head["compression"], nl * row_bytes)
        p = 0
        for r in range(r0, r0 + nl):
            for name, t in chans:
                n = W * _PIXEL[t].itemsize
                planes[name][r] = np.frombuffer(raw, dtype=_PIXEL[t], count=W, offset=p)
                p += n
    out = {name: (planes[name] if t == 0 else planes[name].astype(np.float32)) for name, t in chans}
    return out, head


def read_image(path):
    """What OpenEXRDirectory._load_exr returns (OpenEXRDirectory.py:125-151): float32 [H,W,3] in R,G,B order.  Channel names may carry a
    layer prefix ("ViewLayer.Combined.R"); a file with a single channel (Alpha, Depth written as Y / A / Z / V) is replicated to 3."""
    chans, _ = read_exr(path)
    by_suffix = {}
    for name in chans:
        by_suffix.setdefault(name.rsplit(".", 1)[-1].upper(), name)
    if all(c in by_suffix for c in "RGB"):
        return np.stack([chans[by_suffix[c]] for c in "RGB"], axis=-1).astype(np.float32)
    if len(chans) == 1:
        (only,) = chans.values()
        return np.repeat(only.astype(np.float32)[..., None], 3, axis=-1)
    raise ExrError("%s: no R, G, B channels among %s" % (path, sorted(chans)))


# ---------------------------------------------------------------------------------------------------- writer
def _attr(name, kind, value):
    return name.encode("latin-1") + b"\0" + kind.encode("latin-1") + b"\0" + struct.pack("<i", len(value)) + value


def write_exr(path, channels, compression=ZIP_COMPRESSION):
    """{channel name: [H,W] float32 / float16 / uint32 array} -> single-part scan-line file."""
    if compression not in _LINES:
        raise ExrError("cannot write compression %s" % _CODEC_NAMES.get(compression, compression))
    names = sorted(channels)
    arrs = [np.asarray(channels[n]) for n in names]
    H, W = arrs[0].shape
    chlist = b""
    for n, a in zip(names, arrs):
        if a.shape != (H, W) or a.dtype not in _PIXEL_CODE:
            raise ExrError("channel %s: need a [%d,%d] float32 / float16 / uint32 array, got %s %s" % (n, H, W, a.shape, a.dtype))
        chlist += n.encode("latin-1") + b"\0" + struct.pack("<iB3xii", _PIXEL_CODE[a.dtype], 0, 1, 1)
    chlist += b"\0"
    box = struct.pack("<4i", 0, 0, W - 1, H - 1)
    header = struct.pack("<ii", MAGIC, 2)
    header += _attr("channels", "chlist", chlist) + _attr("compression", "compression", bytes([compression]))
    header += _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box) + _attr("lineOrder", "lineOrder", b"\0")
    header += _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + _attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0))
    header += _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lines = _LINES[compression]
    chunks = []
    for r0 in range(0, H, lines):
        raw = b"".join(a[r].astype(a.dtype.newbyteorder("<"), copy=False).tobytes() for r in range(r0, min(H, r0 + lines)) for a in arrs)
        data = raw
        if compression in (ZIP_COMPRESSION, ZIPS_COMPRESSION):
            data = zlib.compress(_interleave_and_predict(raw))
        elif compression == RLE_COMPRESSION:
            data = _rle_encode(_interleave_and_predict(raw))
        if len(data) >= len(raw):
            data = raw
        chunks.append(struct.pack("<ii", r0, len(data)) + data)
    pos = len(header) + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", pos)
        pos += len(c)
    with open(path, "wb") as f:
        f.write(header + table + b"".join(chunks))


def write_image(path, image, compression=ZIP_COMPRESSION):
    """[H,W,3] (R,G,B) or [H,W] (written as Y) float array -> .exr with FLOAT channels."""
    image = np.asarray(image, dtype=np.float32)
    if image.ndim == 2:
        return write_exr(path, {"Y": image}, compression)
    if image.ndim != 3 or image.shape[2] != 3:
        raise ExrError("write_image takes [H,W,3] or [H,W], got %s" % (image.shape,))
    return write_exr(path, {c: np.ascontiguousarray(image[..., i]) for i, c in enumerate("RGB")}, compression)


# ---------------------------------------------------------------------------------------------------- a directory of render passes
class OpenEXRDirectory:
    """One rendered frame: a directory with one .exr per render pass, matched by '_<Pass>_' in the file name (the underscores keep
    'Normal' and 'Screen Space Normal' apart, OpenEXRDirectory.py:29-53); 1-channel passes keep channel 0 (:66-68).  A missing or
    ambiguous pass and non-finite samples raise ExrError (the reference logs and marks the directory invalid)."""

    def __init__(self, directory):
        self.directory = directory
        self.render_pass_to_image = {}

    def exr_files(self):
        return sorted(os.path.join(self.directory, n) for n in os.listdir(self.directory) if n.endswith(".exr"))

    def file_of(self, render_pass):
        hits = [p for p in self.exr_files() if "_" + render_pass + "_" in os.path.basename(p)]
        if not hits:
            raise ExrError("%s does not contain an exr file for %s" % (self.directory, render_pass))
        if len(hits) > 1:
            raise ExrError("more than one file in %s could be used for the %s pass" % (self.directory, render_pass))
        return hits[0]

    def load_images(self, render_passes, single_channel=()):
        for render_pass in render_passes:
            path = self.file_of(render_pass)
            image = read_image(path)
            if render_pass in single_channel:
                image = image[:, :, 0]
            if not np.isfinite(image).all():
                raise ExrError("there is at least one value in %s which is not finite" % path)
            self.render_pass_to_image[render_pass] = image
        return self.render_pass_to_image

    def size_of_loaded_images(self):
        for image in self.render_pass_to_image.values():
            return image.shape[0], image.shape[1]
        return 0, 0


# ---------------------------------------------------------------------------------------------------- a frame for the Predictor
def load_frame(directory, architecture):
    """The feature dictionary Prediction.main builds from a directory of .exr files (Prediction.py:223-252): one [H,W,C] float32 array
    per required feature under 'source_image/0/<Pass>'.  Passes that are not loaded (`load_data` false) are constant 1.0 (colour) or
    0.5 (direct / indirect).  A pass's file is the one whose name contains '_<Pass>_'; the reference's looser rule -- the first
    listed file whose path contains the pass name -- is the fallback, made deterministic by taking the shortest such name."""
    from .naming import Naming
    frame = OpenEXRDirectory(directory)
    files = frame.exr_files()
    features, size = {}, None
    required = list(architecture.auxiliary_features) + list(architecture.feature_predictions)
    for f in required:
        if not f.load_data:
            continue
        try:
            path = frame.file_of(f.name)
        except ExrError:
            loose = sorted((p for p in files if f.name in os.path.basename(p)), key=lambda p: (len(os.path.basename(p)), p))
            if not loose:
                raise ExrError("image for '%s' could not be loaded or does not exist in %s" % (f.name, directory))
            path = loose[0]
        image = read_image(path)
        if size is None:
            size = image.shape[:2]
        elif size != image.shape[:2]:
            raise ExrError("%s is %dx%d, the other passes are %dx%d" % (path, image.shape[1], image.shape[0], size[1], size[0]))
        features[Naming.source_feature_name(f.name, index=0)] = image[..., :f.number_of_channels]
    if size is None:
        raise ExrError("no pass of the architecture is loaded from files")
    for f in required:
        if not f.load_data:
            value = 1.0 if f.feature_prediction_type == "COLOR" else 0.5        # Prediction.py:244-249
            features[Naming.source_feature_name(f.name, index=0)] = np.full(size + (f.number_of_channels,), value, dtype=np.float32)
    return features


def save_predictions(directory, predictions, as_exr=False):
    """Prediction.py:483-510 stores every denoised pass and the combined image as <directory>/<Pass>.npy for the Blender side;
    as_exr additionally writes <Pass>.exr.  `predictions`: the dictionary Predictor.predict_frame returns."""
    written = []
    for key, value in predictions.items():
        name = key.split("/", 1)[1] if key.startswith("prediction/") else key
        arr = value.detach().cpu().numpy() if hasattr(value, "detach") else np.asarray(value)
        path = os.path.join(directory, name + ".npy")
        np.save(path, arr)
        written.append(path)
        if as_exr:
            write_image(os.path.join(directory, name + ".exr"), arr if arr.shape[-1] == 3 else arr[..., 0])
            written.append(os.path.join(directory, name + ".exr"))
    return written
