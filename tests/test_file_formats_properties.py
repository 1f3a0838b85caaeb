"""Property tests (hypothesis) of the host-side file formats: whatever the writers produce, the readers give back bit for bit --
OpenEXR frames over every supported codec / pixel type / image size, checkpoint bundles over arbitrary names, shapes and dtypes."""
import numpy as np
from hypothesis import given, settings, strategies as st

from deepdenoiser_amd import openexr as X
from deepdenoiser_amd import tf_checkpoint as TC

_DTYPES = [np.float32, np.float16, np.uint32]


@settings(max_examples=40, deadline=None)
@given(h=st.integers(1, 37), w=st.integers(1, 29), compression=st.sampled_from([0, 1, 2, 3]),
       kinds=st.lists(st.sampled_from([0, 1, 2]), min_size=1, max_size=5), seed=st.integers(0, 2 ** 31 - 1), flat=st.booleans())
def test_exr_round_trip(tmp_path_factory, h, w, compression, kinds, seed, flat):
    rng = np.random.default_rng(seed)
    chans = {}
    for i, k in enumerate(kinds):
        dt = _DTYPES[k]
        if dt is np.uint32:
            a = rng.integers(0, 2 ** 32, (h, w), dtype=np.uint32)
        else:
            a = (rng.standard_normal((h, w)) * np.exp(rng.standard_normal((h, w)) * 3)).astype(dt)
        if flat:
            a[: h // 2 + 1] = a[0, 0]                  # long runs: exercises the run-length coder and the "stored raw" fallback both ways
        chans["layer%d.%s" % (i, "RGBAZ"[i])] = a
    path = str(tmp_path_factory.mktemp("exr") / "p.exr")
    X.write_exr(path, chans, compression)
    got, head = X.read_exr(path)
    assert head["data_window"] == (0, 0, w - 1, h - 1) and sorted(got) == sorted(chans)
    for k, v in chans.items():
        want = v if v.dtype == np.uint32 else v.astype(np.float32)
        assert got[k].dtype == want.dtype and np.array_equal(got[k], want, equal_nan=True), k


_names = st.text(alphabet="abcdefghijklmnopqrstuvwxyz_/0123456789", min_size=1, max_size=40)


@settings(max_examples=25, deadline=None)
@given(entries=st.dictionaries(_names, st.tuples(st.lists(st.integers(0, 5), min_size=0, max_size=4), st.sampled_from(["f4", "f8", "i4", "i8", "?", "f2"])),
                               min_size=1, max_size=30), seed=st.integers(0, 2 ** 31 - 1), block=st.sampled_from([64, 512, 4096]))
def test_checkpoint_round_trip(tmp_path_factory, entries, seed, block):
    rng = np.random.default_rng(seed)
    tensors = {}
    for name, (shape, dt) in entries.items():
        a = rng.standard_normal(shape) * 100
        tensors[name] = (a > 0) if dt == "?" else a.astype(dt)
    prefix = str(tmp_path_factory.mktemp("ckpt") / "model.ckpt-1")
    TC.write_checkpoint(prefix, tensors, block_size=block)
    back = TC.read_checkpoint(prefix)
    assert list(back) == sorted(tensors, key=lambda s: s.encode())
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k


@settings(max_examples=30, deadline=None)
@given(examples=st.lists(st.dictionaries(st.text(min_size=1, max_size=30), st.binary(min_size=0, max_size=300), min_size=0, max_size=6), min_size=0, max_size=5),
       gz=st.booleans())
def test_tfrecord_examples_round_trip(tmp_path_factory, examples, gz):
    from deepdenoiser_amd import tfrecords as R
    path = str(tmp_path_factory.mktemp("rec") / ("t_0.tfrecords" + (".gz" if gz else "")))
    R.write_records(path, [R.serialize_example(e) for e in examples])
    got = [R.parse_example(r) for r in R.read_records(path, verify_payload_crc=True)]
    assert got == examples
