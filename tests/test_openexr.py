"""OpenEXR frames without OpenCV (deepdenoiser_amd/openexr.py).  PARITY UNPINNED: no Blender-written file is
available; the reader is checked against files assembled BY HAND here from the published file layout (loops, independent of the
module's writer and of its numpy codec), the writer against the reader."""
import struct
import zlib

import numpy as np
import pytest

from deepdenoiser_amd import openexr as X


def _attr(name, kind, value):
    return name + b"\0" + kind + b"\0" + struct.pack("<i", len(value)) + value


def _header(channels, compression, window, version=2, extra=b""):
    chlist = b"".join(n + b"\0" + struct.pack("<i", t) + b"\0\0\0\0" + struct.pack("<ii", 1, 1) for n, t in channels) + b"\0"
    box = struct.pack("<4i", *window)
    return (struct.pack("<ii", 20000630, version) + _attr(b"channels", b"chlist", chlist) + _attr(b"compression", b"compression", bytes([compression]))
            + _attr(b"dataWindow", b"box2i", box) + _attr(b"displayWindow", b"box2i", box) + _attr(b"lineOrder", b"lineOrder", b"\0")
            + _attr(b"pixelAspectRatio", b"float", struct.pack("<f", 1.0)) + extra + b"\0")


def _zip_longhand(raw):
    """the ZIP chunk transform of the file-layout document, one byte at a time"""
    half1 = bytes(raw[i] for i in range(0, len(raw), 2))
    half2 = bytes(raw[i] for i in range(1, len(raw), 2))
    t = half1 + half2
    d = bytearray([t[0]])
    for i in range(1, len(t)):
        d.append((t[i] - t[i - 1] + 128) & 0xFF)
    return zlib.compress(bytes(d))


def _assemble(path, channels, rows_of, H, W, compression, lines, window=None, coder=None, decreasing=False):
    """channels: sorted [(name, type code, numpy dtype)]; rows_of[name][r] -> row array."""
    window = window or (0, 0, W - 1, H - 1)
    head = _header([(n, t) for n, t, _ in channels], compression, window)
    chunks = []
    for r0 in range(0, H, lines):
        raw = b""
        for r in range(r0, min(H, r0 + lines)):
            for n, _, dt in channels:
                raw += np.asarray(rows_of[n][r], dtype=dt).tobytes()
        data = coder(raw) if coder else raw
        if len(data) >= len(raw):
            data = raw
        chunks.append(struct.pack("<ii", window[1] + r0, len(data)) + data)
    pos = len(head) + 8 * len(chunks)
    offsets = []
    for c in chunks:
        offsets.append(pos)
        pos += len(c)
    body = b"".join(chunks)
    with open(path, "wb") as f:
        f.write(head + struct.pack("<%dQ" % len(offsets), *offsets) + body)


def _planes(H, W, seed):
    rng = np.random.default_rng(seed)
    return {b"B": rng.standard_normal((H, W)).astype(np.float32), b"G": (rng.standard_normal((H, W)) * 4).astype(np.float16),
            b"R": np.linspace(0, 1000, H * W, dtype=np.float32).reshape(H, W), b"id": rng.integers(0, 2 ** 32, (H, W), dtype=np.uint32)}


CHANNELS = [(b"B", 2, "<f4"), (b"G", 1, "<f2"), (b"R", 2, "<f4"), (b"id", 0, "<u4")]


@pytest.mark.parametrize("compression,lines,coder", [(0, 1, None), (3, 16, _zip_longhand), (2, 1, _zip_longhand)])
def test_reader_on_hand_assembled_files(tmp_path, compression, lines, coder):
    H, W = 37, 23                                          # 37 rows: ZIP's last chunk holds 5 scan lines
    planes = _planes(H, W, compression)
    path = str(tmp_path / "f.exr")
    _assemble(path, CHANNELS, planes, H, W, compression, lines, window=(5, -3, 5 + W - 1, -3 + H - 1), coder=coder)
    got, head = X.read_exr(path)
    assert head["compression"] == compression and head["data_window"] == (5, -3, 27, 33)
    assert sorted(got) == ["B", "G", "R", "id"]
    assert got["B"].dtype == np.float32 and np.array_equal(got["B"], planes[b"B"])
    assert got["G"].dtype == np.float32 and np.array_equal(got["G"], planes[b"G"].astype(np.float32))       # HALF widened
    assert got["id"].dtype == np.uint32 and np.array_equal(got["id"], planes[b"id"])
    rgb = X.read_image(path)
    assert rgb.shape == (H, W, 3) and rgb.dtype == np.float32
    assert np.array_equal(rgb[..., 0], planes[b"R"]) and np.array_equal(rgb[..., 2], planes[b"B"])


def test_known_bytes_of_a_two_pixel_file(tmp_path):
    """every byte written out: 2x1 image, one FLOAT channel 'Y', no compression"""
    raw = bytes.fromhex("762f3101" "02000000")
    raw += b"channels\0chlist\0" + struct.pack("<i", 19) + b"Y\0" + bytes.fromhex("02000000" "00" "000000" "01000000" "01000000") + b"\0"
    raw += b"compression\0compression\0" + struct.pack("<i", 1) + b"\0"
    raw += b"dataWindow\0box2i\0" + struct.pack("<i", 16) + struct.pack("<4i", 0, 0, 1, 0)
    raw += b"\0"
    offset = len(raw) + 8
    raw += struct.pack("<Q", offset) + struct.pack("<ii", 0, 8) + struct.pack("<2f", 1.5, -2.0)
    path = str(tmp_path / "two.exr")
    open(path, "wb").write(raw)
    got, _ = X.read_exr(path)
    assert got["Y"].tolist() == [[1.5, -2.0]]
    assert X.read_image(path).tolist() == [[[1.5] * 3, [-2.0] * 3]]            # a single channel is replicated (Alpha / Depth files)


@pytest.mark.parametrize("compression", [X.NO_COMPRESSION, X.RLE_COMPRESSION, X.ZIPS_COMPRESSION, X.ZIP_COMPRESSION])
def test_writer_round_trip(tmp_path, compression):
    H, W = 50, 31
    rng = np.random.default_rng(1)
    img = rng.standard_normal((H, W, 3)).astype(np.float32) * np.exp(rng.standard_normal((H, W, 1))).astype(np.float32)
    img[10:30, 5:25] = 0.25                                  # flat region: runs for RLE, good deflate ratio
    path = str(tmp_path / "w.exr")
    X.write_image(path, img, compression)
    assert np.array_equal(X.read_image(path), img)
    chans = {"ViewLayer.Combined.R": img[..., 0], "ViewLayer.Combined.G": img[..., 1].astype(np.float16), "ViewLayer.Combined.B": img[..., 2],
             "ViewLayer.IndexOB.X": rng.integers(0, 100, (H, W)).astype(np.uint32)}
    X.write_exr(path, chans, compression)
    got, head = X.read_exr(path)
    assert [n for n, _ in head["channels"]] == sorted(chans)
    for k, v in chans.items():
        assert np.array_equal(got[k], v.astype(got[k].dtype)), k
    assert np.array_equal(X.read_image(path)[..., 0], img[..., 0])           # layer-prefixed R, G, B are found
    if compression == X.ZIP_COMPRESSION:
        import os
        assert os.path.getsize(path) < H * W * (4 + 2 + 4 + 4)


def test_rle_coder_longhand_vector():
    data = bytes([7, 7, 7, 7, 1, 2, 3, 3, 9, 9, 9]) + bytes([5]) * 200
    coded = X._rle_encode(data)
    assert coded[:2] == bytes([3, 7])                                         # run of 4 -> count 3
    assert coded[2] == 256 - 4 and coded[3:7] == bytes([1, 2, 3, 3])           # 4 literals (count byte -4), the pair of 3s is no run
    assert coded[7:9] == bytes([2, 9])                                          # run of 3
    assert coded[9:] == bytes([127, 5, 71, 5])                                  # 200 = 128 + 72: a run holds at most 128 bytes
    assert X._rle_decode(coded, len(data)) == data
    with pytest.raises(X.ExrError):
        X._rle_decode(coded, len(data) + 1)


def test_unsupported_files_say_what_they_are(tmp_path):
    p = str(tmp_path / "x.exr")
    open(p, "wb").write(b"\x89PNG\r\n\x1a\n" + b"\0" * 32)
    with pytest.raises(X.ExrError, match="not an OpenEXR file"):
        X.read_exr(p)
    for version, word in ((2 | 0x200, "tiled"), (2 | 0x1000, "multi-part"), (2 | 0x800, "deep")):
        open(p, "wb").write(_header([(b"R", 2)], 0, (0, 0, 0, 0), version=version))
        with pytest.raises(X.ExrError, match=word):
            X.read_exr(p)
    open(p, "wb").write(_header([(b"R", 2)], 4, (0, 0, 0, 0)))
    with pytest.raises(X.ExrError, match="PIZ"):
        X.read_exr(p)
    open(p, "wb").write(_header([(b"R", 2)], 0, (0, 0, 3, 3)))                  # header only: the offset table is missing
    with pytest.raises(X.ExrError, match="truncated"):
        X.read_exr(p)
    X.write_exr(p, {"U": np.zeros((2, 2), np.float32), "V": np.zeros((2, 2), np.float32)})
    with pytest.raises(X.ExrError, match="no R, G, B"):
        X.read_image(p)
    with pytest.raises(X.ExrError, match="need a"):
        X.write_exr(p, {"R": np.zeros((2, 2), np.float64)})


def test_render_pass_directory(tmp_path):
    d = tmp_path / "scene_0001_16_0_0"
    d.mkdir()
    rng = np.random.default_rng(2)
    imgs = {n: rng.random((6, 8, 3)).astype(np.float32) for n in ("Normal", "Screen Space Normal", "Diffuse Color", "Alpha")}
    for n, im in imgs.items():
        X.write_image(str(d / ("frame_%s_0001.exr" % n)), im)
    (d / "notes.txt").write_text("x")
    frame = X.OpenEXRDirectory(str(d))
    got = frame.load_images(["Normal", "Screen Space Normal", "Alpha"], single_channel=("Alpha",))
    assert np.array_equal(got["Normal"], imgs["Normal"]) and np.array_equal(got["Screen Space Normal"], imgs["Screen Space Normal"])
    assert got["Alpha"].shape == (6, 8) and np.array_equal(got["Alpha"], imgs["Alpha"][..., 0])
    assert frame.size_of_loaded_images() == (6, 8)
    with pytest.raises(X.ExrError, match="does not contain"):
        frame.file_of("Glossy Direct")
    X.write_image(str(d / "other_Normal_0002.exr"), imgs["Normal"])
    with pytest.raises(X.ExrError, match="more than one"):
        frame.file_of("Normal")
    bad = imgs["Diffuse Color"].copy()
    bad[0, 0, 0] = np.inf
    X.write_image(str(d / "frame_Diffuse Color_0001.exr"), bad)
    with pytest.raises(X.ExrError, match="not finite"):
        frame.load_images(["Diffuse Color"])


def test_load_frame_and_save_predictions(tmp_path):
    import types
    from deepdenoiser_amd.configs import cfg2_unet_kpcn
    from deepdenoiser_amd.architecture import Architecture
    arch = Architecture(cfg2_unet_kpcn(), device="cpu", dtype="bf16")            # parsing only: no program is built on the CPU
    rng = np.random.default_rng(3)
    truth = {}
    for f in arch.auxiliary_features + arch.feature_predictions:
        if f.load_data:
            truth[f.name] = rng.random((10, 12, 3)).astype(np.float32)
            X.write_image(str(tmp_path / ("render_%s_0007.exr" % f.name)), truth[f.name])
    feats = X.load_frame(str(tmp_path), arch)
    assert sorted(feats) == sorted(arch.required_source_names())
    for f in arch.auxiliary_features + arch.feature_predictions:
        got = feats["source_image/0/" + f.name]
        assert got.shape == (10, 12, f.number_of_channels) and got.dtype == np.float32
        if f.load_data:
            assert np.array_equal(got, truth[f.name][..., :f.number_of_channels])
        else:
            assert np.all(got == (1.0 if f.feature_prediction_type == "COLOR" else 0.5))
    # a pass of another size is refused; a missing pass is named
    some = next(f.name for f in arch.feature_predictions if f.load_data)
    X.write_image(str(tmp_path / ("render_%s_0007.exr" % some)), np.zeros((4, 4, 3), np.float32))
    with pytest.raises(X.ExrError, match="other passes"):
        X.load_frame(str(tmp_path), arch)
    import os
    os.remove(str(tmp_path / ("render_%s_0007.exr" % some)))
    with pytest.raises(X.ExrError, match="could not be loaded"):
        X.load_frame(str(tmp_path), arch)
    out = tmp_path / "out"
    out.mkdir()
    import torch
    preds = {"prediction/Diffuse Direct": torch.rand(5, 6, 3), "Combined": np.ones((5, 6, 3), np.float32), "prediction/Alpha": torch.rand(5, 6, 1)}
    written = X.save_predictions(str(out), preds, as_exr=True)
    assert len(written) == 6 and np.array_equal(np.load(str(out / "Diffuse Direct.npy")), preds["prediction/Diffuse Direct"].numpy())
    assert np.array_equal(X.read_image(str(out / "Combined.exr")), preds["Combined"])
    assert np.array_equal(X.read_image(str(out / "Alpha.exr"))[..., 0], preds["prediction/Alpha"].numpy()[..., 0])
