"""CPU restatement of the reference's training-time data augmentation (TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product).

Follows TensorFlow/DataAugmentation.py:10-200 function by function and TensorFlow/Training.py:551-604 / :794-821 for which pass gets which
operation.  PARITY UNPINNED against live TensorFlow (TF is not installable here); the TF ops involved are pinned by their documented numpy
equivalents: tf.image.flip_left_right == x[:, ::-1], tf.image.rot90(x, k) == np.rot90(x, k) (counter-clockwise), tf.matmul == @.
All functions take ONE example [H, W, C] (the reference maps them over a tf.data pipeline of single tiles).
"""
import math

import numpy as np

NORMAL, SCREEN_SPACE_NORMAL = "Normal", "Screen Space Normal"            # RenderPasses.py:14-15
_NON_RGB = ("Alpha", "Depth", "Mist", "Normal", "Screen Space Normal", "Motion Vector", "Object ID", "Material ID", "UV")   # RenderPasses.py:66-79
_PERMUTATIONS = {1: (0, 2, 1), 2: (1, 0, 2), 3: (1, 2, 0), 4: (2, 0, 1), 5: (2, 1, 0)}                                    # DataAugmentation.py:124-129


def flip_left_right(x, name, flip):
    """DataAugmentation.py:10-29."""
    if name == NORMAL:
        raise Exception("Flipping for normals is not supported.")       # :22-23 (raised at graph construction, whatever `flip` is)
    if flip > 0:
        x = x[:, ::-1]
        if name == SCREEN_SPACE_NORMAL:                                  # :31-43
            x = np.concatenate([-x[..., 0:1], x[..., 1:2], x[..., 2:3]], axis=-1)
    return x


def rotate_90(x, k, name):
    """DataAugmentation.py:45-62 and :64-111."""
    x = np.rot90(x, k)
    if name == SCREEN_SPACE_NORMAL:
        nx, ny, nz = x[..., 0:1], x[..., 1:2], x[..., 2:3]
        if k == 1:
            nx, ny = -ny, nx
        elif k == 2:
            nx, ny = -nx, -ny
        elif k == 3:
            nx, ny = ny, -nx
        x = np.concatenate([nx, ny, nz], axis=-1)
    return x


def permute_rgb(x, permute):
    """DataAugmentation.py:113-132."""
    if permute in _PERMUTATIONS:
        p = _PERMUTATIONS[permute]
        x = np.concatenate([x[..., p[0]:p[0] + 1], x[..., p[1]:p[1] + 1], x[..., p[2]:p[2] + 1]], axis=-1)
    return x


def random_rotation_matrix(random_vector, dtype=np.float32):
    """DataAugmentation.py:134-186 (Graphics Gems III rand_rotation), evaluated in `dtype` like the TF graph (float32)."""
    f = dtype
    two_pi = f(2.0) * f(math.pi)
    theta, phi, z = f(random_vector[0]) * two_pi, f(random_vector[1]) * two_pi, f(random_vector[2]) * f(2.0)
    r = np.sqrt(z)
    vx, vy, vz = np.sin(phi) * r, np.cos(phi) * r, np.sqrt(f(2.0) - z)
    st, ct = np.sin(theta), np.cos(theta)
    sx, sy = vx * ct - vy * st, vx * st + vy * ct
    m = [vx * sx - ct, vx * sy - st, vx * vz,
         vy * sx + st, vy * sy - ct, vy * vz,
         vz * sx, vz * sy, f(1.0) - z]
    return np.asarray(m, dtype=dtype).reshape(3, 3)


def rotate_normal(x, rotation_matrix):
    """DataAugmentation.py:188-200."""
    h, w = x.shape[0], x.shape[1]
    return (x.reshape(h * w, 3) @ rotation_matrix).reshape(h, w, 3)


def augment_example(name, x, draw, usage):
    """One pass of one example through FeatureTrainingAugmentation (Training.py:568-594) in the order of Training.py:806-816.
    draw: dict(flip, rotate, permute, normal_rotation[3,3]); usage: dict(use_flip_left_right, use_rotate_90, use_rgb_permutation, use_normal_rotation)."""
    if usage["use_flip_left_right"]:
        x = flip_left_right(x, name, draw["flip"])
    if usage["use_rotate_90"]:
        x = rotate_90(x, draw["rotate"], name)
    if usage["use_rgb_permutation"] and name not in _NON_RGB:
        x = permute_rgb(x, draw["permute"])
    if usage["use_normal_rotation"] and name == NORMAL:
        x = rotate_normal(x, draw["normal_rotation"])
    return x
