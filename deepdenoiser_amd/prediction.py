"""Full-frame inference: halo tiling -> batched forward on the HIP path -> on-device crop/stitch -> recombination.

Mirrors the reference's Prediction.main (TensorFlow/Prediction.py:188-520) for the part that is on the hot path:
the integer tile plan and crop windows (bit-exact, tiling.py), row-major tile order (:325-326, :380-382), stitch
(:384-441) and recombination (:443-481).  Differences by design: tiles are batched
(the reference's batch-1 Estimator.predict and its temporary TFRecord round trip, :316-338, are not part of the
contract), and crop/stitch/recombine run on the device.  EXR decode (cv2) is out of scope: frames are given as
tensors keyed by the reference's feature names.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from .naming import Naming
from .tiling import tile_plan

_COMBINED = ("Diffuse", "Glossy", "Subsurface", "Transmission")
_SINGLES = ("Volume Direct", "Volume Indirect", "Environment", "Emission")


class Predictor:
    def __init__(self, architecture, tile_size=128, tile_overlap_size=14, tiles_per_batch=16, use_graph=True):
        self.arch, self.tile_size, self.tile_overlap_size, self.tiles_per_batch = architecture, tile_size, tile_overlap_size, tiles_per_batch
        self.lib = L.load()
        self.use_graph = use_graph
        self._plans, self._graphs = {}, {}
        self.profile = None          # a list: predict_frame appends (start, before forward, after forward, end) timing events per tile batch / frame

    def prepare(self, H, W):
        """Build (and cache) the tile program for an HxW frame; the network parameters exist after this call, so weights are loaded
        with `architecture.params.load_list(...)` between prepare() and predict_frame()."""
        self._frame_plan(H, W)

    def _frame_plan(self, H, W):
        """Tile plan, gather indices and stitch tables of a frame size (cached: every frame of a sequence shares them)."""
        key = (H, W)
        if key in self._plans:
            return self._plans[key]
        dev = self.arch.device
        plan = tile_plan(H, W, self.tile_size, self.tile_overlap_size)
        T = plan.tile
        n_batches = -(-plan.count // max(1, self.tiles_per_batch))
        Bt = -(-plan.count // n_batches)          # balanced batches: 209 tiles at <= 64 per batch -> 4 x 53, not 3 x 64 + 17
        prog = self.arch.program(Bt, T, T)
        NF = prog.NF
        # ONE_HOT_ENCODING: the reference's prediction input_fn adds the constant one-hot planes of every flag name to the source
        # dictionary (Prediction.py:97-98 -> FeatureFlags.add_to_source_dictionary); they are written once here.  A caller may still
        # pass 'feature_flag/<name>' frames, which predict_frame() tiles like any other input.
        for name, buf in prog.flags_raw.items():
            buf.zero_()
            buf[..., self.arch.feature_flag_names.index(name)] = 1.0
        origins = plan.windows()                                                              # row-major (Prediction.py:380-382)
        grid = [(hi, wi) for hi in range(plan.rows.count) for wi in range(plan.cols.count)]
        chunks = []
        for start in range(0, plan.count, Bt):
            ids = list(range(start, min(start + Bt, plan.count)))
            pad = ids + [ids[-1]] * (Bt - len(ids))                                           # ragged last batch: repeat a tile, never stitched
            oyx = torch.tensor([[origins[t][0], origins[t][1]] for t in pad], dtype=torch.int32, device=dev)     # dd_extract_tiles table
            table = (L.StitchEntry * (len(ids) * NF))()
            n = 0
            for f in range(NF):
                for slot, ti in enumerate(ids):
                    hi, wi = grid[ti]
                    (cy0, cy1), (cx0, cx1) = plan.rows.crops[hi], plan.cols.crops[wi]
                    table[n] = L.StitchEntry(f * Bt + slot, cy0, cy1, cx0, cx1, f, plan.rows.offsets[hi], plan.cols.offsets[wi])
                    n += 1
            tdev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(dev)
            chunks.append((oyx, tdev, n))
        self._plans[key] = (plan, prog, chunks)
        return self._plans[key]

    def predict_frame(self, features):
        """features: {'source_image/0/<Pass>': [H,W,C] float32 tensor} -> {'prediction/<Pass>': [H,W,C]} (+ 'Combined' if all passes exist)."""
        arch, lib = self.arch, self.lib
        dev = arch.device
        names = arch.required_source_names()
        frame = {k: torch.as_tensor(features[k], dtype=torch.float32).to(dev).contiguous() for k in names}
        H, W = frame[names[0]].shape[0], frame[names[0]].shape[1]
        plan, prog, chunks = self._frame_plan(H, W)
        T, NF = plan.tile, prog.NF
        frames = torch.empty((NF, H, W, 3), dtype=torch.float32, device=dev)      # (the crops of the tile plan cover every pixel exactly once: tests/test_tiling_golden.py)
        # the MFMA operand images are re-packed only when the weights may have changed since this program last packed them (a frame sequence
        # runs on fixed weights: 30 us per 1080p frame); DD_PACK_EVERY_FRAME=1: always
        key = arch.params.state_key()
        if self.__dict__.setdefault("_packed", {}).get(id(prog)) != key or os.environ.get("DD_PACK_EVERY_FRAME", "0") == "1":
            prog.pack_weights()
            self._packed[id(prog)] = key
        stream = prog.g.stream_ptr()
        feats = prog.head + arch.auxiliary_features
        for f in feats:
            fr = frame[Naming.source_feature_name(f.name, index=0)]
            if fr.dim() != 3 or fr.shape[2] < f.number_of_channels or tuple(fr.shape[:2]) != (H, W):
                raise ValueError("%s: expected a [%d,%d,>=%d] frame, got %s" % (f.name, H, W, f.number_of_channels, tuple(fr.shape)))
        flag_frames = {}
        for name, buf in prog.flags_raw.items():
            key = Naming.feature_flags_name(name)
            if key not in features:      # the constant one-hot plane (Prediction.py:108): rewritten per frame, a previous caller's planes must not linger
                buf.zero_()
                buf[..., arch.feature_flag_names.index(name)] = 1.0
            if key in features:
                fr = torch.as_tensor(features[key], dtype=torch.float32).to(dev).contiguous()
                if tuple(fr.shape) != (H, W, prog.flags_raw[name].shape[3]):
                    raise ValueError("%s: expected a [%d,%d,%d] frame, got %s" % (key, H, W, prog.flags_raw[name].shape[3], tuple(fr.shape)))
                flag_frames[name] = fr
        # Render passes whose frames have exactly the pass's channels are read in place by the input assembly (no tile copies); anything else
        # (wider frames, DD_FRAME_INPUT=0, the unfused input path) goes through dd_extract_tiles as before.
        direct = (prog.fused_input and os.environ.get("DD_FRAME_INPUT", "1") != "0"
                  and all(frame[Naming.source_feature_name(f.name, index=0)].shape[2] == f.number_of_channels for f in feats))
        if direct:
            if prog.frame_input != (H, W):      # (programs are cached per (tiles per batch, tile): another frame size may share this one)
                prog.enable_frame_input(H, W)
            prog.set_frame_sources({f.name: frame[Naming.source_feature_name(f.name, index=0)] for f in feats})
        elif prog.frame_input is not None:      # (programs are cached per architecture: an earlier frame may have been read in place)
            prog.disable_frame_input()
        ev = None
        if self.profile is not None:      # (bench.py: where a frame's time goes on the device, and how long the device waits for the host between frames)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        for oyx, tdev, n in chunks:
            if direct:
                prog.frame_origins.copy_(oyx, non_blocking=True)
            for name, fr in flag_frames.items():
                raw = prog.flags_raw[name]
                L.check(lib.dd_extract_tiles(fr.data_ptr(), H, W, fr.shape[2], fr.shape[2], raw.data_ptr(), T, raw.shape[3],
                                             oyx.data_ptr(), oyx.shape[0], stream))
            for f in (() if direct else feats):                         # halo tiles straight into the program's input buffers
                fr = frame[Naming.source_feature_name(f.name, index=0)]
                raw = prog.raw[f.name]
                L.check(lib.dd_extract_tiles(fr.data_ptr(), H, W, fr.shape[2], f.number_of_channels, raw.data_ptr(), T, raw.shape[3],
                                             oyx.data_ptr(), oyx.shape[0], stream))
            if ev is not None and oyx is chunks[0][0]:
                ev[1].record()
            self._forward(prog)
            if ev is not None and oyx is chunks[-1][0]:
                ev[2].record()
            tiles = prog.predictions[0]                                  # [NF*Bt, T, T, 3], feature-major
            L.check(lib.dd_stitch(tiles.ptr, T, 3, frames.data_ptr(), H, W, 3, 3, tdev.data_ptr(), n, stream))
        out = {}
        for f in arch.feature_predictions:
            if f.is_target and f.load_data:
                out[Naming.feature_prediction_name(f.name)] = frames[prog.head_index[f.name]][..., :f.number_of_channels]
        self._recombine(prog, frames, out, H * W, stream)
        if ev is not None:
            ev[3].record()
            self.profile.append(ev)
        return out

    def _forward(self, prog):
        """The tile program's forward launches, replayed from a hipGraph after one eager pass (the ~80 launches of a batch of tiles
        cost more on the host than on the device once the kernels are fast)."""
        if not self.use_graph:
            prog.forward(pack=False)
            return
        key = (id(prog), prog.frame_input)             # (a graph replays the input-assembly launch it was captured with)
        gr = self._graphs.get(key)
        if gr is None:
            prog.forward(pack=False)                 # eager once: binds every launch, warms the caches
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                prog.forward(pack=False)
            self._graphs[key] = gr
        gr.replay()

    def _recombine(self, prog, frames, out, npix, stream):
        idx = prog.head_index
        need = [c + s for c in _COMBINED for s in (" Color", " Direct", " Indirect")] + list(_SINGLES)
        if not all(n in idx and prog.head[idx[n]].load_data for n in need):
            return
        comb = torch.zeros((len(_COMBINED) + 1,) + tuple(frames.shape[1:]), dtype=torch.float32, device=frames.device)
        d = L.RecombineDesc()
        d.n_triples = len(_COMBINED)
        for k, c in enumerate(_COMBINED):
            d.color[k], d.direct[k], d.indirect[k] = (frames[idx[c + s]].data_ptr() for s in (" Color", " Direct", " Indirect"))
            d.combined[k] = comb[k].data_ptr()
            out[Naming.feature_prediction_name(c)] = comb[k]
        d.n_singles = len(_SINGLES)
        for j, s in enumerate(_SINGLES):
            d.single[j] = frames[idx[s]].data_ptr()
        d.image = comb[len(_COMBINED)].data_ptr()
        L.check(self.lib.dd_recombine(C.byref(d), npix, stream))
        out["Combined"] = comb[len(_COMBINED)]
