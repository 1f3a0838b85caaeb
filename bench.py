#!/usr/bin/env python
"""Headline benchmark: train tiles/s of the U-Net KPCN hot path (BASELINE.json config 2) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one full training step of the hot path over one batch of synthetic render-pass tiles already resident in HBM:
standardize+variance+assemble (32 ch) -> U-Net [64,96,128]x4 -> 1x1 x2 -> 5x5 kernel-prediction apply at 3 scales ->
multiscale compose x2 -> inverse standardization -> multi-scale SMAPE loss -> full backward -> (RCCL gradient all-reduce
for N>1) -> Adam.  Unit: 128x128 tile-passes per second, whole job.  bf16 storage / MFMA with fp32
accumulate, fp32 master weights and optimizer state.  Weak scaling: per-GPU batch fixed.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel = the MFMA implicit-GEMM
conv, timed per launch with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle -- a PyTorch-CPU restatement
of the TensorFlow graph, TF itself is unavailable -- timed on a bounded sample of the same workload; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense MFMA peak, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0          # HBM3E, same guide


def synthetic_inputs(arch, B, H, W, device, seed):
    from deepdenoiser_amd.naming import Naming
    g = torch.Generator(device="cpu").manual_seed(seed)
    feats, labels = {}, {}
    for f in arch.feature_predictions + arch.auxiliary_features:
        v = torch.randn(B, H, W, f.number_of_channels, generator=g)
        if f.name != "Normal":
            v = v.abs() * torch.exp(0.5 * torch.randn(B, H, W, 1, generator=g))      # HDR-like radiance proxy
        feats[Naming.source_feature_name(f.name, index=0)] = v.to(device)
    for f in arch.feature_predictions:
        labels[Naming.target_feature_name(f.name)] = torch.randn(B, H, W, f.number_of_channels, generator=g).abs().to(device)
    return feats, labels


def measured_traffic(B, dtype, family="conv_igemm"):
    """HBM bytes per conv_igemm launch from the committed PMC measurement of this configuration (profiles/*_hbm_traffic.json: FETCH_SIZE /
    WRITE_SIZE passes of rocprofv3, collected and corrected as MI355X_MICROARCH.md prescribes).  Counters cannot be collected from inside the
    timed run, so the figure is reported only when a measurement of the same batch size and dtype exists; otherwise null."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")), reverse=True):
        try:
            m = json.load(open(path))
        except (OSError, ValueError):
            continue
        if family == "conv_bwd":      # conv_bwd_kernel + conv_bwd96_kernel (the PMC table lists them under one substring)
            if m.get("tiles_per_gpu_per_step") == B and m.get("dtype") == dtype and "conv_bwd" in m:
                c = m["conv_bwd"]
                return {"bytes_per_launch": 1e6 * (c["fetch_MB_per_launch"] + c["write_MB_per_launch"]), "source": os.path.relpath(path, ROOT)}
            continue
        if m.get("tiles_per_gpu_per_step") == B and m.get("dtype") == dtype and ("conv_igemm" in m or "conv_rw" in m):      # (round 4: every launch of the family is a register-weight kernel)
            # the bench's conv_igemm launch family = the LDS-weight kernels plus the register-weight kernel dd_conv_igemm forwards to (PMC lists
            # them by kernel name): dispatch-weighted mean of the two
            fams = [m[k] for k in ("conv_igemm", "conv_rw") if k in m]
            n = sum(c["dispatches"] for c in fams)
            mb = sum(c["dispatches"] * (c["fetch_MB_per_launch"] + c["write_MB_per_launch"]) for c in fams) / n
            return {"bytes_per_launch": 1e6 * mb, "source": os.path.relpath(path, ROOT)}
    return None


def hbm_copy_rate(device, nbytes=1 << 30, reps=5):
    """What HBM delivers to a plain streaming copy on THIS box (read + write of `nbytes` each way, far beyond the 256 MB infinity cache), GB/s.
    The rooflines keep the guide's 8 TB/s peak; this is the practical ceiling a fused kernel's algorithmic bytes can be read against."""
    src = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=device).normal_()
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    del src, dst
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def family_bytes(launches, family, esz):
    """Algorithmic HBM bytes of the TIMED launches of one family (the per-launch records profile_ops returns -- the same list the family's time
    and flops come from): every input and output element once, plus the output-shaped operands a launch reads (ReLU mask, residual, the
    gradient it accumulates into)."""
    if family == "conv_bwd":      # fused backward: dy (k = C_out) + x (n = C_in) read, dx (n) written unless weights-only, + the gradient it accumulates into
        return sum(r["B"] * r["H"] * r["W"] * (r["k"] + r["n"] * (1 + (0 if r.get("weights_only") else 1) + (1 if r.get("accumulate") else 0))) * esz
                   for tag, r, _ in launches if tag == family and r and "k" in r and "n" in r)
    return sum(r["B"] * r["H"] * r["W"] * (r["k"] + r["n"] * (1 + r.get("extra_reads", 0))) * esz
               for tag, r, _ in launches if tag == family and r and "k" in r and "n" in r)


def cpu_baseline(aj, tj, H, W, budget_s=20.0):
    """The oracle timed on the host cores: fwd + loss + bwd + Adam of the same network on a bounded sample.
    The thread count is chosen by a short sweep (oneDNN with every hardware thread of a 2-socket host on a B=4 batch is far
    slower than a few dozen threads); `cores` reports the threads actually used."""
    from oracle.model import OracleArchitecture
    from oracle import training as OT
    from deepdenoiser_amd.naming import Naming
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    B = 4
    o = OracleArchitecture(aj, dtype=torch.float32, seed=2)
    g = torch.Generator().manual_seed(0)
    feats = {Naming.source_feature_name(f.name, index=0): torch.randn(B, H, W, f.channels, generator=g).abs() for f in o.features + o.auxiliary}
    labels = {Naming.target_feature_name(f.name): torch.randn(B, H, W, f.channels, generator=g).abs() for f in o.features}
    state = ([], [])
    step = [0]

    def one():
        step[0] += 1
        t = time.time()
        OT.train_step(o, aj, tj, feats, labels, state, step[0])
        return time.time() - t

    best_t, best_n = None, None
    sweep_start = time.time()
    for n in sorted({c for c in (8, 16, 32, 64, min(avail, 64)) if c <= avail}):      # all 256 threads of the GPU box: 113 s per step (measured)
        torch.set_num_threads(n)
        one()                                   # warm-up (allocations, oneDNN primitive caches)
        t = one()
        if best_t is None or t < best_t:
            best_t, best_n = t, n
        if time.time() - sweep_start > budget_s:
            break
    torch.set_num_threads(best_n)
    t0, n = time.time(), 0
    while True:
        one()
        n += 1
        if time.time() - t0 > budget_s / 2 or n >= 20:
            break
    dt = time.time() - t0
    return {"value": B * n / dt, "unit": "tiles/s", "cores": best_n, "host_threads_available": avail, "kind": "port",
            "sample": "%d training steps of %d tile(s) %dx%dx32ch, same network, fp32, PyTorch-CPU restatement of the TF graph "
                      "(oracle/; TensorFlow itself is not installable); thread count = best of a short sweep" % (n, B, H, W)}


def inference_frames(device, dtype, tile, batch, steps, warmup, seed, buffer_sets=3, detail=None):
    """Times `steps` full 1080x1920 frames through Predictor on this rank; returns seconds.  A frame SEQUENCE: the frames rotate through
    `buffer_sets` distinct sets of device tensors, so the per-frame upload of the input-assembly table runs (with one set it is skipped, as for a
    caller that refills its tensors in place).  detail (a dict): filled with the device-time breakdown of the timed frames (HIP events inside
    predict_frame): launches in front of the forward, the forward graph, stitch / recombination behind it, and the idle time BETWEEN frames
    (device waiting for the host)."""
    from deepdenoiser_amd import configs
    from deepdenoiser_amd.architecture import Architecture
    from deepdenoiser_amd.naming import Naming
    from deepdenoiser_amd.prediction import Predictor
    H, W = 1080, 1920
    arch = Architecture(configs.cfg2_unet_kpcn(), device=device, dtype=dtype, seed=2)
    pred = Predictor(arch, tile_size=tile, tile_overlap_size=14, tiles_per_batch=batch)
    g = torch.Generator().manual_seed(seed)
    frames = [{Naming.source_feature_name(f.name, index=0): torch.randn(H, W, f.number_of_channels, generator=g).abs().to(device)
               for f in arch.feature_predictions + arch.auxiliary_features} for _ in range(max(1, buffer_sets))]
    for i in range(max(2, warmup)):
        pred.predict_frame(frames[i % len(frames)])
    torch.cuda.synchronize()
    if detail is not None:
        pred.profile = []
    t0 = time.perf_counter()
    for i in range(steps):
        pred.predict_frame(frames[i % len(frames)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if detail is not None and pred.profile:
        ev = pred.profile
        n = len(ev)
        detail["before_forward_ms"] = sum(e[0].elapsed_time(e[1]) for e in ev) / n
        detail["forward_graph_ms"] = sum(e[1].elapsed_time(e[2]) for e in ev) / n
        detail["stitch_recombine_ms"] = sum(e[2].elapsed_time(e[3]) for e in ev) / n
        detail["idle_between_frames_ms"] = sum(ev[i][3].elapsed_time(ev[i + 1][0]) for i in range(n - 1)) / max(1, n - 1)
        detail["frames_timed"] = n
    return dt


def inference_bench(args, device, rank, world):
    """Full-frame inference: 1080x1920x32ch frame -> 209 halo tiles of 128^2 -> U-Net KPCN forward -> stitch.
    A step = one frame; every rank denoises its own frames (replicas only, no collective).  MPix/s counts OUTPUT pixels."""
    H, W = 1080, 1920
    dt = inference_frames(device, args.dtype, args.tile, args.batch, args.steps, args.warmup, 7 + rank)
    if rank == 0:
        print(json.dumps({"metric": "inference MPix/s (1920x1080 frame, halo-tiled 128x128x32ch U-Net KPCN)", "value": world * args.steps * H * W / dt / 1e6,
                          "unit": "MPix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                          "config": {"workload": "BASELINE config 5: full-frame 1920x1080 inference, 209 halo tiles (overlap 14), tiles per batch %d" % args.batch}}))


def _timed_steps(trainer, steps, warmup):
    for _ in range(warmup):
        trainer.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def _conv_roofline(prog, peak):
    """MFMA conv launches of one step of `prog` (HIP events per launch): algorithmic FLOPs / their time, and the step's launch-time total."""
    times, launches = prog.profile_ops(repeats=3, detail=True)
    convs = {k: v for k, v in times.items() if k in ("conv_igemm", "conv_bwd", "conv_wgrad", "convt") and v[1] > 0}
    ms, fl = sum(v[1] for v in convs.values()), sum(v[2] for v in convs.values())
    roof = {"bound": "mfma", "kernel": "all MFMA conv launches of the step", "achieved": fl / (ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
            "frac": fl / (ms * 1e-3) / 1e12 / peak, "conv_ms_per_step": round(ms, 3), "all_launches_ms_per_step": round(sum(v[1] for v in times.values()), 3),
            "algorithmic_gflop_per_step": fl / 1e9, "families_ms": {k: round(v[1], 3) for k, v in times.items()}}
    # which roof binds: algorithmic bytes of the forward / data-gradient launches (every input and output element once, plus the output-shaped
    # operands a launch reads: mask, residual, accumulated gradient) against their flops.  Below the ridge (peak / 8 TB/s = 312 flop/B in
    # bf16) the launches are HBM-bound and `bound`, `achieved`, `peak`, `frac` are restated in bytes; the MFMA view stays in `mfma_view`.
    if "conv_igemm" in times and times["conv_igemm"][1] > 0 and times["conv_igemm"][2] > 0:
        by = family_bytes(launches, "conv_igemm", 4 if prog.arch.dtype == "f32" else 2)
        fam_fl, fam_ms = times["conv_igemm"][2], times["conv_igemm"][1]
        intensity, ridge = fam_fl / by, 1e3 * peak / PEAK_HBM_GBS
        hbm = {"achieved": by / (fam_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": by / (fam_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
               "flop_per_byte": intensity, "ridge_flop_per_byte": ridge, "launch_family": "conv_igemm (forward + data gradients)",
               "ms_per_step": round(fam_ms, 3), "algorithmic_bytes_per_step": by}
        if intensity < ridge:
            roof["mfma_view"] = {k: roof[k] for k in ("achieved", "peak", "unit", "frac")}
            roof.update({"bound": "hbm", "kernel": hbm["launch_family"] + " launches of the step", "achieved": hbm["achieved"], "peak": hbm["peak"],
                         "unit": "GB/s", "frac": hbm["frac"]})
        roof["hbm_view"] = hbm
    return roof


def extras(device, B, H, W):
    """Secondary measurements carried by the default line so that the driver's record holds them too (BASELINE.json's metric also names
    "inference MPix/s"; the 1e-4 parity gate applies to the f32 storage path, whose throughput is reported beside the bf16 one; BASELINE
    config 3 = Tiramisu + MultiScalePrediction at 256x256; the literal ArchitectureExample.json = 17 weight-shared tuple passes per tile)."""
    from deepdenoiser_amd import configs
    from deepdenoiser_amd.architecture import Architecture
    from deepdenoiser_amd.prediction import Predictor
    from deepdenoiser_amd.naming import Naming
    from deepdenoiser_amd.training import Trainer
    out = {}
    steps = 10
    detail = {}
    dt = inference_frames(device, "f16", 128, 256, steps, 2, 7, buffer_sets=3, detail=detail)
    dt_same = inference_frames(device, "f16", 128, 256, steps, 2, 7, buffer_sets=1)
    out["inference"] = {"metric": "inference MPix/s (1920x1080 frame sequence, 209 halo tiles of 128x128x32ch, fp16 MFMA path, output pixels)",
                        "value": steps * 1080 * 1920 / dt / 1e6, "unit": "MPix/s", "ms_per_frame": 1e3 * dt / steps, "dtype": "f16", "frames": steps,
                        "distinct_frames": True, "frame_buffer_sets": 3,
                        "same_buffers": {"value": steps * 1080 * 1920 / dt_same / 1e6, "ms_per_frame": 1e3 * dt_same / steps,
                                         "note": "every frame in the same device tensors: the input-assembly table upload is skipped"},
                        "device_time_ms_per_frame": {k: round(v, 4) if isinstance(v, float) else v for k, v in detail.items()}}
    try:        # the same frames with the tile size a user may pass (Prediction.py --tile_size, default 128): 256-pixel tiles recompute 1.26 x the
        # frame's pixels in their halos instead of 1.65 x.  Reported beside the default, never as `value`.
        dt256 = inference_frames(device, "f16", 256, 64, steps, 2, 7, buffer_sets=3)
        out["inference"]["tile_size_256"] = {"value": steps * 1080 * 1920 / dt256 / 1e6, "unit": "MPix/s", "ms_per_frame": 1e3 * dt256 / steps,
                                             "note": "--tile_size 256 --tile_overlap_size 14 (45 tiles); the headline uses the reference's defaults 128 / 14 (209 tiles)"}
    except Exception as e:
        out["inference"]["tile_size_256"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    try:        # roofline of the inference frame: per-launch HIP events of the forward program of the 209-tile batch
        arch = Architecture(configs.cfg2_unet_kpcn(), device=device, dtype="f16", seed=2)
        pred = Predictor(arch, tile_size=128, tile_overlap_size=14, tiles_per_batch=256)
        pred.prepare(1080, 1920)
        prog = pred._plans[(1080, 1920)][1]
        times = prog.profile_ops(repeats=3)
        fl = times.get("conv_igemm", (0, 0.0, 0.0))[2]
        ms_conv = times.get("conv_igemm", (0, 1e-9, 0.0))[1]
        ms_all = sum(v[1] for v in times.values())
        out["inference"]["roofline"] = {
            "bound": "mfma", "kernel": "3x3 / transposed conv launches of the frame's forward (209 tiles, 1.65x halo recompute included in the FLOPs)",
            "achieved": fl / (ms_conv * 1e-3) / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": fl / (ms_conv * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
            "conv_ms_per_frame": round(ms_conv, 3), "whole_frame": {"tflops": fl / (1e-3 * out["inference"]["ms_per_frame"]) / 1e12,
                                                                   "frac": fl / (1e-3 * out["inference"]["ms_per_frame"]) / 1e12 / PEAK_BF16_TFLOPS},
            "families_ms": {k: round(v[1], 3) for k, v in times.items()}, "launch_time_ms_per_frame": round(ms_all, 3)}
        del pred, prog, arch
    except Exception as e:      # the secondary measurement must never take the headline line down
        out["inference"]["roofline"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    Bf = min(B, 32)
    arch = Architecture(configs.cfg2_unet_kpcn(), device=device, dtype="f32", seed=2)
    trainer = Trainer(arch, configs.bench_training(), Bf, H, W, world_size=1, use_graph=True)
    feats, labels = synthetic_inputs(arch, Bf, H, W, device, seed=1000)
    trainer.program.set_inputs(feats, labels)
    dt = _timed_steps(trainer, 5, 3)
    out["f32_path"] = {"metric": "train tiles/sec on the f32 storage path (exact-f32 MFMA; the path the 1e-4 parity gate applies to)",
                       "value": Bf / dt, "unit": "tiles/s", "ms_per_step": 1e3 * dt, "tiles_per_step": Bf, "dtype": "f32"}
    del trainer, arch
    torch.cuda.empty_cache()
    # ---- the headline workload at a larger per-GPU batch.  The headline stays at 128 tile passes per step -- the reference's default step is
    #      batch_size 8 x 17 tuples = 136 tuple passes (TrainingExample.json:15) -- but every launch pays a fixed prologue and
    #      every weight gradient an atomic tail, so throughput still rises with the batch: reported, not used for `value`.
    if B == 128:
        out["batch_256"] = {}
        try:
            arch = Architecture(configs.cfg2_unet_kpcn(), device=device, dtype="bf16", seed=2)
            trainer = Trainer(arch, configs.bench_training(), 256, H, W, world_size=1, use_graph=True)
            feats, labels = synthetic_inputs(arch, 256, H, W, device, seed=1000)
            trainer.program.set_inputs(feats, labels)
            dt = _timed_steps(trainer, 10, 5)
            out["batch_256"] = {"metric": "train tiles/sec, same workload at 256 tile passes per GPU and step", "value": 256 / dt, "unit": "tiles/s",
                                "ms_per_step": 1e3 * dt, "tiles_per_step": 256, "dtype": "bf16"}
            del trainer, arch, feats, labels
        except Exception as e:
            out["batch_256"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    # ---- BASELINE config 3: Tiramisu (FC-DenseNet) + MultiScalePrediction, 256x256 tiles, training step, bf16
    out["cfg3"] = {}
    # (B = 8 and B = 32: BASELINE.md lists both batch sizes; at 8 tiles the light configuration's 107 launches are partly a batch artefact)
    for name, filters, Bc in (("tiramisu_16_24_32", (16, 24, 32), 8), ("tiramisu_16_24_32_b32", (16, 24, 32), 32), ("tiramisu_64_96_128_heavy", (64, 96, 128), 8)):
        try:
            arch = Architecture(configs.cfg3_tiramisu(filters=filters, convs=4), device=device, dtype="bf16", seed=2)
            trainer = Trainer(arch, configs.bench_training(), Bc, 256, 256, world_size=1, use_graph=True)
            feats, labels = synthetic_inputs(arch, Bc, 256, 256, device, seed=1000)
            trainer.program.set_inputs(feats, labels)
            dt = _timed_steps(trainer, 5, 3)
            roof = _conv_roofline(trainer.program, PEAK_BF16_TFLOPS)
            out["cfg3"][name] = {"metric": "train tiles/sec (256x256x32ch Tiramisu F=%s x4 + 5x5 KernelPrediction + 3-scale MultiScalePrediction)" % (list(filters),),
                                 "value": Bc / dt, "unit": "tiles/s", "ms_per_step": 1e3 * dt, "tiles_per_step": Bc, "dtype": "bf16",
                                 "parameters": int(arch.params.total), "roofline": roof,
                                 "whole_step": {"tflops": roof["algorithmic_gflop_per_step"] / (1e3 * dt), "frac": roof["algorithmic_gflop_per_step"] / (1e3 * dt) / PEAK_BF16_TFLOPS}}
            del trainer, arch
        except Exception as e:
            out["cfg3"][name] = {"error": repr(e)}
        torch.cuda.empty_cache()
    # ---- the literal ArchitectureExample.json + TrainingExample.json: 17 SINGLE tuples sharing the backbone weights (folded into the batch),
    #      C_in = 16 with EMBEDDING flags (and their gradient), feature / combined (x5) / image (x10) losses; B = 8 tiles -> 136 tuple passes
    try:
        Be = 8
        arch = Architecture(configs.example_architecture(), device=device, dtype="bf16", seed=2)
        trainer = Trainer(arch, configs.training(), Be, H, W, world_size=1, use_graph=True)
        feats, labels = synthetic_inputs(arch, Be, H, W, device, seed=1000)
        trainer.program.set_inputs(feats, labels)
        dt = _timed_steps(trainer, 10, 3)
        out["example_json"] = {"metric": "train tiles/sec of the literal ArchitectureExample.json / TrainingExample.json (17 tuple passes per tile, measured)",
                               "value": Be / dt, "unit": "tiles/s", "ms_per_step": 1e3 * dt, "tiles_per_step": Be, "tuple_passes_per_step": Be * trainer.program.T,
                               "tuple_passes_per_s": Be * trainer.program.T / dt, "input_channels": arch.input_channels(), "dtype": "bf16",
                               "fused_head": bool(trainer.program.fused_head)}
        del trainer, arch
    except Exception as e:
        out["example_json"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    return out


def augment_bench(args, device, rank, world):
    """Device data augmentation of one batch of cfg-2 tiles: every source pass + the target, one dd_augment launch each.
    HBM-bound: each element is read once and written once (fp32)."""
    from deepdenoiser_amd import configs
    from deepdenoiser_amd.architecture import Architecture
    from deepdenoiser_amd.data_augmentation import DataAugmentation, DataAugmentationUsage
    arch = Architecture(configs.cfg2_unet_kpcn(), device=device, dtype=args.dtype, seed=2)
    B, T = args.batch, args.tile
    feats, labels = synthetic_inputs(arch, B, T, T, device, seed=3 + rank)
    usage = DataAugmentationUsage(True, False, True, True)              # TrainingExample.json:17-23
    draws = DataAugmentation.draw(B, generator=torch.Generator().manual_seed(1))
    for _ in range(max(1, args.warmup)):
        DataAugmentation.apply(feats, labels, draws, usage)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        DataAugmentation.apply(feats, labels, draws, usage)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nbytes = 2 * 4 * sum(v.numel() for v in list(feats.values()) + list(labels.values()))
    if rank == 0:
        gbs = nbytes * args.steps / dt / 1e9
        print(json.dumps({"metric": "data augmentation tiles/s (rot90 + rgb permutation + normal rotation, all passes of a 128x128x32ch tile)",
                          "value": world * B * args.steps / dt, "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": "augmentation of %d tiles x %d tensors per step (incl. host-side launch overhead)" % (B, len(feats) + len(labels))},
                          "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": None}}))


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU) through torch.distributed.run on this node and
    pass their output through -- rank 0 prints the one JSON line with n_gpus = N.  Refuses (non-zero exit) when fewer than N devices are visible
    instead of measuring fewer (DD_FORCE_DEVICE: the test hook that puts several ranks on one GPU over gloo)."""
    import socket
    import subprocess
    if not os.environ.get("DD_FORCE_DEVICE") and torch.cuda.device_count() < n:
        raise SystemExit("--gpus %d: only %d device(s) visible; refusing to measure fewer GPUs than asked for" % (n, torch.cuda.device_count()))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this host driver (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="tile-passes per GPU per step (default 128); inference: tiles per batch "
                                                            "(default 256: the 209 tiles of a 1920x1080 frame go through in one batch)")
    ap.add_argument("--tile", type=int, default=128)
    ap.add_argument("--dtype", default=None, choices=["bf16", "f16", "f32"],
                    help="storage type (MFMA input type; accumulation is fp32): default bf16 for training, f16 for --mode inference (BASELINE config 5)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements of the default line (inference MPix/s, f32-path tiles/s)")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step into a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="train", choices=["train", "inference", "augment"],
                    help="inference: full-frame 1920x1080 halo-tiled prediction (BASELINE config 5), reported as MPix/s -- a secondary line, "
                         "the driver's contract is the default train mode")
    ap.add_argument("--dump-launches", default="", help="write the per-launch records of the profiling pass (family, shape, flops, us) to this JSON file")
    ap.add_argument("--host-inputs", action="store_true",
                    help="PCIe-inclusive variant (NOT the headline value): every step first copies its batch from pinned host memory")
    args = ap.parse_args()

    if args.batch is None:
        args.batch = 256 if args.mode == "inference" else 128
    if args.dtype is None:
        args.dtype = "f16" if args.mode == "inference" else "bf16"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not os.environ.get("DD_FORCE_DEVICE") and torch.cuda.device_count() < world:
        raise SystemExit("--gpus %d: only %d device(s) visible; refusing to measure fewer GPUs than asked for" % (world, torch.cuda.device_count()))
    if os.environ.get("DD_FORCE_DEVICE"):                  # test hook: several ranks on one GPU (with DD_DIST_BACKEND=gloo)
        local_rank = int(os.environ["DD_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # DD_FORCE_COLLECTIVES=1: initialise the process group and issue every collective even in a world of ONE rank (launched through
    # torch.distributed.run --nproc-per-node 1): the RCCL communicator, the side-stream all-reduces next to the captured hipGraph segments and
    # the barriers of this file all run on a one-GPU box (tests/test_gpu_rccl.py).  Not a scaling measurement.
    force_coll = os.environ.get("DD_FORCE_COLLECTIVES", "0") != "0"
    use_dist = world > 1 or force_coll
    if use_dist:
        import torch.distributed as dist
        backend = os.environ.get("DD_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from deepdenoiser_amd import configs
    from deepdenoiser_amd.architecture import Architecture
    from deepdenoiser_amd.training import Trainer

    if args.mode == "inference":
        return inference_bench(args, device, rank, world)
    if args.mode == "augment":
        return augment_bench(args, device, rank, world)

    aj, tj = configs.cfg2_unet_kpcn(), configs.bench_training()
    arch = Architecture(aj, device=device, dtype=args.dtype, seed=2)       # identical init on every rank
    B, H, W = args.batch, args.tile, args.tile
    trainer = Trainer(arch, tj, B, H, W, world_size=world, use_graph=not args.no_graph, force_collectives=force_coll)
    feats, labels = synthetic_inputs(arch, B, H, W, device, seed=1000 + rank)    # per-rank data shard
    trainer.program.set_inputs(feats, labels)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.host_inputs:
        hf = {k: v.cpu().pin_memory() for k, v in feats.items()}
        hl = {k: v.cpu().pin_memory() for k, v in labels.items()}

        # double-buffered staging on a copy stream: the upload of step k + 1 runs beside step k (deepdenoiser_amd.program.HostInputStager)
        stager = trainer.program.host_stager()
        stager.stage(hf, hl)

        def one_step():
            stager.consume()
            stager.stage(hf, hl)
            trainer.step()
    else:
        one_step = trainer.step

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    loss = float(trainer.program.loss_buf)
    if use_dist:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)

    # ---- the gradient all-reduces of a step on their side stream, and the part the launch stream waited for (every rank runs these steps)
    coll = trainer.measure_collectives(5) if use_dist else None
    if use_dist:
        barrier()

    # ---- roofline of the dominant kernel: per-launch HIP-event timing on the launch stream (outside the timed region)
    roof = None
    if rank == 0:
        times, launches = trainer.program.profile_ops(detail=True)           # {kernel family: (launches, total_ms, flops)}, per-launch records
        if args.dump_launches:
            # every launch of the step in program order: family, shape record (B, H, W, taps, n = C_out, k = C_in, flops, flags), HIP-event us --
            # what `roofline` is computed from, so that `frac` can be recomputed per layer without reading engine.py
            with open(args.dump_launches, "w") as fh:
                json.dump([{"family": tag, "us": round(us, 2), **({k: v for k, v in info.items()} if info else {})} for tag, info, us in launches], fh, indent=0)
        # the DOMINANT kernel family of the step = the one with the most time (round 6: with the 96-channel level's backward fused, the fused
        # data + weight gradient launches -- conv_bwd_kernel, conv_bwd96_kernel -- take more of the step than the dd_conv_igemm launches, which
        # were the dominant family through round 5; both stay in `mfma_families`)
        fam = max((k for k in ("conv_igemm", "conv_bwd") if k in times and times[k][2] > 0), key=lambda k: times[k][1])
        n, ms, flops = times[fam]
        # the family's work is the sum over the launches that were timed; the engine's own records must say the same (round 4: they did not --
        # the layer-wise compose net recorded launches its fused replacement never runs)
        recs = trainer.program.g.conv_records if fam == "conv_igemm" else trainer.program.g.bwd_records
        rec_flops = sum(r["flops"] for r in recs)
        assert abs(rec_flops - flops) <= 1e-9 * flops, ("%s records %.1f GFLOP != timed launches %.1f GFLOP" % (fam, rec_flops / 1e9, flops / 1e9))
        achieved = flops / (ms * 1e-3) / 1e12
        peak = PEAK_F32_TFLOPS if args.dtype == "f32" else PEAK_BF16_TFLOPS      # fp16 and bf16 MFMA run at the same rate
        # algorithmic HBM bytes of the same launches: every input and output element once, plus the output-shaped operands some launches
        # read (the ReLU mask of a dgrad, a residual, the gradient a launch accumulates into)
        esz = 4 if args.dtype == "f32" else 2
        alg_bytes = family_bytes(launches, fam, esz)
        tr = measured_traffic(B, args.dtype, fam)
        kname = {"conv_igemm": "dd_conv_igemm launches <%s>: conv_igemm_ws_kernel, conv_rw_kernel, conv_rw8_kernel (forward + the data gradients no fused "
                               "backward kernel covers)" % args.dtype,
                 "conv_bwd": "dd_conv3x3_bwd launches <%s>: conv_bwd_kernel (<= 64 output channels), conv_bwd96_kernel (65 - 96): data + weight + bias "
                             "gradient of a 3x3 layer in one launch" % args.dtype}[fam]
        roof = {"bound": "mfma", "kernel": kname, "family": fam, "achieved": achieved, "peak": peak,
                "unit": "TFLOP/s", "frac": achieved / peak, "traffic": tr["bytes_per_launch"] if tr else None,
                "traffic_source": tr["source"] if tr else None, "algorithmic_bytes_per_launch": alg_bytes / n,
                "launches_per_step": n, "avg_launch_us": 1e3 * ms / n,
                "algorithmic_gflop_per_step": flops / 1e9,
                # the same launches seen from the memory side: a 64->64 3x3 conv in bf16 moves 2*64*2 B per pixel for 2*9*64*64 flop,
                # 288 flop/B against a ridge of 2500/8 = 312 flop/B, so the U-Net's widest layers sit AT the ridge and every
                # narrower layer (24/25/32 channels) is HBM-bound: both fractions are reported, the larger one is the binding roof
                "hbm_view": {"achieved": alg_bytes / n / (1e-3 * ms / n) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": alg_bytes / n / (1e-3 * ms / n) / 1e9 / PEAK_HBM_GBS,
                             "flop_per_byte": flops / alg_bytes, "ridge_flop_per_byte": 1e3 * peak / PEAK_HBM_GBS},
                # the practical ceilings of this box, measured in the same process: a plain 1 GiB copy (torch, read + write), and the family's
                # slowest-roof estimate -- its algorithmic bytes at that copy rate against its time (a launch that is also at the MFMA ridge
                # cannot beat either).  The 64 -> 64 layers at 128 x 128 move 537 MB in ~125 us: the copy rate itself
                "practical": None,
                "other_kernels_ms_per_step": {k: round(v[1], 3) for k, v in times.items() if k != fam}}
        # every MFMA conv family of the step (conv_bwd = the fused data + weight gradient launch of csrc/dd_conv_bwd.hip, conv_wgrad = the
        # weight-gradient launches of the layers it does not cover) and their aggregate: the step's conv FLOPs over the time of all of them
        try:
            cr = hbm_copy_rate(device)
            roof["practical"] = {"hbm_copy_GBs": cr, "hbm_copy_frac_of_peak": cr / PEAK_HBM_GBS,
                                 "family_bytes_at_copy_rate_ms": alg_bytes / cr / 1e6, "family_ms": ms,
                                 "frac_of_copy_rate": (alg_bytes / (1e-3 * ms) / 1e9) / cr}
        except Exception as e:
            roof["practical"] = {"error": repr(e)}
        convs = {k: times[k] for k in ("conv_igemm", "conv_bwd", "conv_wgrad") if k in times and times[k][1] > 0}
        roof["mfma_families"] = {k: {"launches_per_step": v[0], "ms_per_step": round(v[1], 3), "tflops": v[2] / (v[1] * 1e-3) / 1e12,
                                     "frac": v[2] / (v[1] * 1e-3) / 1e12 / peak} for k, v in convs.items()}
        tot_ms, tot_fl = sum(v[1] for v in convs.values()), sum(v[2] for v in convs.values())
        roof["all_conv_launches"] = {"ms_per_step": round(tot_ms, 3), "tflops": tot_fl / (tot_ms * 1e-3) / 1e12, "frac": tot_fl / (tot_ms * 1e-3) / 1e12 / peak}
        # the whole step against the same peak: algorithmic flops of forward + backward (every conv family incl. the transposed convs, the
        # compose net 41 808 flop per fine pixel and the kernel-prediction head 2 C K + 2 K K + 6 K per pixel, each x 3 for fwd + bwd) over
        # the TIMED step (value above), not over the profiled launches
        NT_ = trainer.program.NF * B
        extra_fl = sum(3.0 * NT_ * (H >> sc) * (W >> sc) * 41808.0 for sc in range(2)) + \
            sum(3.0 * NT_ * (H >> sc) * (W >> sc) * (2.0 * c * 25 + 2.0 * 25 * 25 + 150.0) for sc, c in enumerate((64, 96, 128)))
        step_fl = sum(v[2] for v in times.values()) + extra_fl
        roof["whole_step"] = {"algorithmic_gflop_per_step": step_fl / 1e9, "ms_per_step": 1e3 * dt / args.steps,
                              "tflops": step_fl / (dt / args.steps) / 1e12, "frac": step_fl / (dt / args.steps) / 1e12 / peak}
    if rank == 0:
        out = {
            "metric": "train tiles/sec (128x128x32ch U-Net KPCN)", "value": world * B * args.steps / dt, "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            # physical devices behind the ranks: == n_gpus except under the DD_FORCE_DEVICE test hook (several ranks sharing one GPU)
            "devices": 1 if os.environ.get("DD_FORCE_DEVICE") else world,
            "collectives": (os.environ.get("DD_DIST_BACKEND", "nccl") if use_dist else None),
            # per step: time of the bucketed gradient all-reduces on the side stream and how much of it was NOT hidden behind the backward
            "allreduce": coll,
            "config": {"workload": "BASELINE config 2: U-Net [64,96,128]x4 + 5x5 KernelPrediction + 3-scale MultiScalePrediction, "
                                   "32-channel render-pass stack, %dx%d tiles, full training step (fwd+SMAPE loss+bwd+Adam)" % (H, W),
                       "tiles_per_gpu_per_step": B, "global_batch": world * B, "parallelism": "dp%d" % world,
                       "hipgraph": not args.no_graph, "final_loss": loss,
                       # a tile of the reference's example JSON is 17 SINGLE tuple passes through the network
                       "example_json_tiles_per_s": world * B * args.steps / dt / 17.0, "inputs": "pinned host, copied every step" if args.host_inputs else "resident in HBM"},
            "roofline": roof,
        }
        if world == 1 and not args.no_extras:
            out["extras"] = extras(device, B, H, W)
            if not args.host_inputs:
                # the same step fed from pinned host memory every step (PCIe-inclusive; never `value`): double-buffered staging on a copy stream
                hf = {k: v.cpu().pin_memory() for k, v in feats.items()}
                hl = {k: v.cpu().pin_memory() for k, v in labels.items()}
                stager = trainer.program.host_stager()
                stager.stage(hf, hl)
                for i in range(args.warmup + args.steps):
                    if i == args.warmup:
                        torch.cuda.synchronize()
                        th = time.perf_counter()
                    stager.consume()
                    stager.stage(hf, hl)
                    trainer.step()
                torch.cuda.synchronize()
                dth = time.perf_counter() - th
                out["extras"]["host_inputs"] = {"value": B * args.steps / dth, "unit": "tiles/s", "ms_per_step": 1e3 * dth / args.steps,
                                                "frac_of_resident": (B * args.steps / dth) / out["value"],
                                                "bytes_per_step": sum(v.numel() * v.element_size() for v in list(hf.values()) + list(hl.values())),
                                                "inputs": "pinned host fp32, copied every step on a copy stream (2 staging slots), then device-to-device into the program's buffers"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(aj, tj, H, W)
        print(json.dumps(out))
    if use_dist:
        barrier()                      # rank 0 may still be timing single launches for the roofline: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
