"""Training-time data augmentation on the device: rot90 / flip / RGB permutation / normal rotation, applied consistently to every pass of
a tile with one set of random draws per tile -- the MI355X counterpart of the reference's DataAugmentation (TensorFlow/DataAugmentation.py)
as driven by Training.data_augmentation and FeatureTrainingAugmentation (TensorFlow/Training.py:551-604, :794-821).

The reference maps single tiles through a tf.data pipeline; here a whole batch of tiles is augmented per pass by one launch of
`dd_augment` with a per-tile draw record.  TF's random stream cannot be reproduced, so the draws are explicit (`draw()` uses a torch
generator with the reference's distributions: flip in {0,1}, rotate in {0..3}, permute in {0..5}, three uniform floats for the rotation).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L
from .naming import Naming
from .render_passes import RenderPasses

NORMAL, SCREEN_SPACE_NORMAL = "Normal", "Screen Space Normal"


class DataAugmentationUsage:
    """DataAugmentation.py:203-209 (same argument order)."""

    def __init__(self, use_rotate_90, use_flip_left_right, use_rgb_permutation, use_normal_rotation):
        self.use_rotate_90, self.use_flip_left_right = bool(use_rotate_90), bool(use_flip_left_right)
        self.use_rgb_permutation, self.use_normal_rotation = bool(use_rgb_permutation), bool(use_normal_rotation)

    @staticmethod
    def from_training_json(training_json):
        """Training.py:977-979."""
        j = training_json["data_augmentation"]
        return DataAugmentationUsage(j["use_rotate_90"], j["use_flip_left_right"], j["use_rgb_permutation"], j["use_normal_rotation"])


class DataAugmentation:
    @staticmethod
    def random_rotation_matrix(random_vector):
        """DataAugmentation.py:134-186 (Graphics Gems III rand_rotation) in float32, row-major [3,3]."""
        f = np.float32
        two_pi = f(2.0) * f(math.pi)
        theta, phi, z = f(random_vector[0]) * two_pi, f(random_vector[1]) * two_pi, f(random_vector[2]) * f(2.0)
        r = np.sqrt(z)
        vx, vy, vz = np.sin(phi) * r, np.cos(phi) * r, np.sqrt(f(2.0) - z)
        st, ct = np.sin(theta), np.cos(theta)
        sx, sy = vx * ct - vy * st, vx * st + vy * ct
        return np.asarray([vx * sx - ct, vx * sy - st, vx * vz, vy * sx + st, vy * sy - ct, vy * vz, vz * sx, vz * sy, f(1.0) - z],
                          dtype=np.float32).reshape(3, 3)

    @staticmethod
    def draw(batch, generator=None):
        """Per-tile random draws with the reference's distributions (Training.py:796-801)."""
        flip = torch.randint(0, 2, (batch,), generator=generator)
        rotate = torch.randint(0, 4, (batch,), generator=generator)
        permute = torch.randint(0, 6, (batch,), generator=generator)
        vec = torch.rand((batch, 3), generator=generator)
        mats = np.stack([DataAugmentation.random_rotation_matrix(v) for v in vec.numpy()])
        return {"flip": flip.numpy().astype(np.int32), "rotate": rotate.numpy().astype(np.int32), "permute": permute.numpy().astype(np.int32),
                "normal_rotation": mats}

    @staticmethod
    def _kind(name, channels):
        if channels != 3:
            return L.AUG_PLAIN
        if name == NORMAL:
            return L.AUG_NORMAL
        if name == SCREEN_SPACE_NORMAL:
            return L.AUG_SCREEN_NORMAL
        return L.AUG_RGB if RenderPasses.is_rgb_color_render_pass(name) else L.AUG_PLAIN

    @staticmethod
    def apply(features, labels, draws, usage, stream=None):
        """Augment every tensor of the source / target dictionaries ({'source_image/<i>/<Pass>': [B,H,W,C] float32 device tensor}).
        Returns new dictionaries; the inputs are not modified."""
        lib = L.load()
        any_t = next(iter(features.values()))
        B, dev = int(any_t.shape[0]), any_t.device
        table = (L.AugmentDraw * B)()
        for b in range(B):
            table[b].flip, table[b].rotate, table[b].permute = int(draws["flip"][b]), int(draws["rotate"][b]), int(draws["permute"][b])
            for k, v in enumerate(np.asarray(draws["normal_rotation"][b], dtype=np.float32).reshape(9)):
                table[b].normal_rotation[k] = float(v)
        tdev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(dev)
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream

        def one(key, t):
            name = key.split("/")[-1]
            if usage.use_flip_left_right and name == NORMAL:
                raise Exception("Flipping for normals is not supported.")          # DataAugmentation.py:22-23
            x = t.to(torch.float32).contiguous()
            _, H, W, Cn = x.shape
            out = torch.empty_like(x)
            L.check(lib.dd_augment(x.data_ptr(), out.data_ptr(), Cn, B, H, W, tdev.data_ptr(), DataAugmentation._kind(name, Cn),
                                   int(usage.use_flip_left_right), int(usage.use_rotate_90), int(usage.use_rgb_permutation),
                                   int(usage.use_normal_rotation), s))
            return out

        # feature-flag planes ('feature_flag/<name>') are constant over a tile and stay as they are
        out_f = {k: (one(k, v) if k.startswith("source_image/") else v) for k, v in features.items()}
        out_l = {k: one(k, v) for k, v in labels.items()} if labels is not None else None
        # (the draw table is freed in stream order by the caching allocator: the launches above read it before any later reuse)
        return out_f, out_l
