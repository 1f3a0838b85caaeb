"""Data augmentation row: the oracle against hand-derived known answers (CPU) and the HIP kernel against the oracle (GPU)."""
import numpy as np
import pytest
import torch

from oracle import augment_ref as A


def test_flip_rot_known_answers():
    x = np.arange(4, dtype=np.float32).reshape(2, 2, 1)            # [[0,1],[2,3]]
    assert A.flip_left_right(x, "Diffuse Color", 1)[..., 0].tolist() == [[1, 0], [3, 2]]
    assert A.flip_left_right(x, "Diffuse Color", 0)[..., 0].tolist() == [[0, 1], [2, 3]]
    assert A.rotate_90(x, 1, "Depth")[..., 0].tolist() == [[1, 3], [0, 2]]        # counter-clockwise (tf.image.rot90)
    assert A.rotate_90(x, 2, "Depth")[..., 0].tolist() == [[3, 2], [1, 0]]
    assert A.rotate_90(x, 3, "Depth")[..., 0].tolist() == [[2, 0], [3, 1]]
    with pytest.raises(Exception):
        A.flip_left_right(np.zeros((2, 2, 3), np.float32), "Normal", 0)


def test_screen_space_normal_and_permutation_known_answers():
    v = np.array([[[1.0, 2.0, 3.0]]], dtype=np.float32)
    assert A.flip_left_right(v, "Screen Space Normal", 1)[0, 0].tolist() == [-1, 2, 3]
    assert A.rotate_90(v, 1, "Screen Space Normal")[0, 0].tolist() == [-2, 1, 3]
    assert A.rotate_90(v, 2, "Screen Space Normal")[0, 0].tolist() == [-1, -2, 3]
    assert A.rotate_90(v, 3, "Screen Space Normal")[0, 0].tolist() == [2, -1, 3]
    want = {0: [1, 2, 3], 1: [1, 3, 2], 2: [2, 1, 3], 3: [2, 3, 1], 4: [3, 1, 2], 5: [3, 2, 1]}
    for p, w in want.items():
        assert A.permute_rgb(v, p)[0, 0].tolist() == w


def test_random_rotation_matrix_is_a_rotation():
    rng = np.random.default_rng(0)
    for _ in range(20):
        m = A.random_rotation_matrix(rng.random(3), dtype=np.float64)
        assert np.allclose(m @ m.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(m) - 1.0) < 1e-12
    assert np.allclose(A.random_rotation_matrix([0.0, 0.0, 0.0], dtype=np.float64), np.diag([-1.0, -1.0, 1.0]), atol=1e-12)


def test_pass_rules():
    usage = {"use_flip_left_right": False, "use_rotate_90": True, "use_rgb_permutation": True, "use_normal_rotation": True}
    draw = {"flip": 0, "rotate": 0, "permute": 3, "normal_rotation": np.diag([-1.0, -1.0, 1.0]).astype(np.float32)}
    v = np.array([[[1.0, 2.0, 3.0]]], dtype=np.float32)
    assert A.augment_example("Diffuse Color", v, draw, usage)[0, 0].tolist() == [2, 3, 1]       # rgb passes are permuted
    assert A.augment_example("Normal", v, draw, usage)[0, 0].tolist() == [-1, -2, 3]            # normals are rotated, not permuted
    assert A.augment_example("Depth", v[..., :1], draw, usage)[0, 0].tolist() == [1]            # non-rgb passes: geometry only
    assert A.augment_example("Motion Vector", v, draw, usage)[0, 0].tolist() == [1, 2, 3]


@pytest.mark.gpu
@pytest.mark.parametrize("flip_on", [False, True])
def test_device_augmentation_matches_oracle(flip_on):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from deepdenoiser_amd.data_augmentation import DataAugmentation, DataAugmentationUsage
    from deepdenoiser_amd.naming import Naming
    B, T = 24, 20                     # every (flip, rotate, permute) combination occurs at least once
    g = torch.Generator().manual_seed(5)
    passes = {"Diffuse Color": 3, "Depth": 1, "Screen Space Normal": 3, "Motion Vector": 3, "Emission": 3}
    if not flip_on:
        passes["Normal"] = 3
    feats = {Naming.source_feature_name(n, index=0): torch.randn(B, T, T, c, generator=g) for n, c in passes.items()}
    labels = {Naming.target_feature_name("Diffuse Color"): torch.randn(B, T, T, 3, generator=g)}
    draws = DataAugmentation.draw(B, generator=g)
    draws["flip"] = (np.arange(B) % 2).astype(np.int32)
    draws["rotate"] = ((np.arange(B) // 2) % 4).astype(np.int32)
    draws["permute"] = ((np.arange(B) // 4) % 6).astype(np.int32)
    usage = DataAugmentationUsage(True, flip_on, True, True)
    out_f, out_l = DataAugmentation.apply({k: v.cuda() for k, v in feats.items()}, {k: v.cuda() for k, v in labels.items()}, draws, usage)
    torch.cuda.synchronize()
    u = {"use_flip_left_right": flip_on, "use_rotate_90": True, "use_rgb_permutation": True, "use_normal_rotation": True}
    for table, got in ((feats, out_f), (labels, out_l)):
        for key, t in table.items():
            name = key.split("/")[-1]
            for b in range(B):
                d = {"flip": int(draws["flip"][b]), "rotate": int(draws["rotate"][b]), "permute": int(draws["permute"][b]),
                     "normal_rotation": draws["normal_rotation"][b]}
                want = A.augment_example(name, t[b].numpy(), d, u)
                have = got[key][b].cpu().numpy()
                if name == "Normal":
                    assert np.allclose(have, want, rtol=0, atol=2e-6), (key, b)
                else:
                    assert np.array_equal(have, want), (key, b, d)     # pure data movement / sign flips: bit exact
    if flip_on:
        with pytest.raises(Exception):
            DataAugmentation.apply({Naming.source_feature_name("Normal", index=0): torch.zeros(1, 4, 4, 3).cuda()}, None,
                                   DataAugmentation.draw(1, generator=g), usage)
