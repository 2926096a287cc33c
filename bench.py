"""bench.py — frames/s of DVIS++ R50 offline inference on synthetic 720p clips (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames T]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`--gpus N` without a launcher (no WORLD_SIZE in the environment) re-executes itself as N ranks under torch.distributed.run
on 127.0.0.1 (one rank per GPU, RCCL); under a launcher WORLD_SIZE must equal --gpus.

A "step" is one pass of the whole hot path over one synthetic clip of T=30 uint8 720x1280 frames that is already
resident in HBM (step i gets its own clip, seed 1234 + i): normalise/pad -> R50 backbone -> MSDeformAttn pixel decoder -> masked-attention decoder ->
(all-gather of per-frame queries when N > 1) -> referring tracker (incl. host assignment) -> temporal refiner ->
mask contraction -> panoptic post-processing to integer masks on the device.  With N GPUs the SAME clips are sharded by
frame (strong scaling); value = T * K / max-over-ranks(wall time).  Weights: deterministic random init with the
reference's init rules (no checkpoints offline).  fp32 throughout (the parity target is the fp32 path).  The timed pass
runs in the product's default strict mode: a glue op that would quietly take a torch formulation on the GPU raises instead.

N > 1: `value` is north_star's split (frames sharded, ONE all-gather of the per-frame queries per clip, tracker + refiner
replicated); `owner_rounds` holds a second timed pass with stream()'s tracker-owner rounds; `dist` lists world size,
backend and every rank's device.

Also reported on the one JSON line:
  latency_ms    per-clip latency (first enqueue of the clip -> its outputs complete), p10 / p50 / p90 over the timed clips;
  stages_ms     stage-synchronised breakdown of one clip (untimed extra pass);
  candidates_100  the same workload with every non-void query sent to the panoptic stage (short second timed pass);
  roofline      the dominant hand-written kernel of the path, MSDeformAttn forward: algorithmic bytes (61 824 000 B
                per frame-layer) / its mean launch time (HIP events on the launch stream) vs 8 TB/s.
  roofline_tracker  the referring tracker's recurrence (its captured hipGraph replayed alone): the tracker's parameter bytes
                x frames / replay time vs 8 TB/s — SURVEY.md §8(d)'s weight-streaming bound for the latency-bound chain.
  cpu_baseline  the oracle's restatement of the reference pipeline (torch fp32 CPU ops, all host cores) on a bounded
                sample of the same clip, in the metric's own unit, plus the oracle's C port of the MSDeformAttn kernel
                (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MSDA_BYTES_PER_FRAME_LAYER = 4 * (19320 * 8 * 32 + 19320 * 8 * 3 * 4 * 2 + 19320 * 8 * 3 * 4 + 19320 * 8 * 32)  # 61 824 000
# what the sampling actually moves ON CHIP: every (query, head, level, point) gathers 4 corner lines of 128 B through the vector
# L1 / texture-address path into VGPRs — 949.6 MB per frame-layer, 15x the compulsory HBM bytes; that path delivers 64 B per clock
# and CU (x 256 CUs x 2.1 GHz under load = 34.4 TB/s): the roof this kernel is actually under
MSDA_GATHERED_BYTES_PER_FRAME_LAYER = 19320 * 8 * 3 * 4 * 4 * 128
L1_PATH_PEAK_GBS = 64 * 256 * 2.1
HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TF = 157.3      # dense fp32 matrix peak (MI355X_MICROARCH.md)
MFMA_F16_PEAK_TF = 2500.0     # dense f16 / bf16 matrix peak (MI355X_MICROARCH.md)


def synthetic_clip(T, device, seed=1234):
    """T uint8 (3, 720, 1280) frames: uniform noise blended with a low-frequency pattern (non-degenerate masks)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    noise = torch.randint(0, 256, (T, 3, 720, 1280), generator=g, dtype=torch.uint8)
    yy, xx = torch.meshgrid(torch.linspace(0, 6.28, 720), torch.linspace(0, 6.28, 1280), indexing="ij")
    frames = []
    for t in range(T):
        pat = (127 + 100 * torch.sin(xx * (1 + t % 3) + 0.2 * t) * torch.cos(yy * 2 + 0.1 * t)).clamp(0, 255)
        frames.append(((noise[t].float() * 0.3 + pat[None] * 0.7)).to(torch.uint8))
    return torch.stack(frames).to(device)


def tracker_roofline(model, n=10):
    """"tracker: latency-bound (report achieved vs weight-streaming bound 101 MB/frame)".  Replays the
    captured hipGraph of the referring tracker's recurrence (the strictly sequential part: dvis_plus_amd/tracker.py
    _recurrence, dvis_Plus/tracker.py:277-318) on the current stream between two HIP events: algorithmic bytes = the
    tracker's fp32 parameters, which every frame of the recurrence streams once, x frames per replay."""
    trk = getattr(model, "tracker", None)
    cache = getattr(getattr(trk, "_graph", None), "_cache", None)
    if not cache:
        return None
    graph, static_in, _ = next(reversed(cache.values()))
    T, Q, B, C = static_in[0].shape
    params = sum(p.numel() for p in trk.parameters())
    for _ in range(2):
        graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / n
    gbs = 4.0 * params * T / sec / 1e9
    return {"bound": "hbm (weight streaming; latency-bound in practice)",
            "kernel": "referring tracker recurrence (hipGraph: dvis_gemm_ln / dvis_gemm_nt / attn_short chain)",
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": None, "alg_bytes_per_frame": 4 * params, "frames_per_replay": int(T), "clips_per_replay": int(B),
            "ms_per_replay": round(sec * 1e3, 3), "us_per_frame": round(sec * 1e6 / T, 1), "replays_timed": n,
            "dependent_launches_per_frame": 36 if getattr(trk, "fused_chain", False) else 65,
            "note": "the recurrence alone, nothing else on the device; 36 launches per frame = 6 (reference MLP, 48-head "
                    "cross-attention, batched out-projection) + 6 layers x 5 (profiles/r04_tracker_timeline.txt)"}


class MsdaTimer:
    """Times every MSDeformAttn launch of the timed region with HIP events recorded on the launch stream
    (torch's current stream IS the stream the C ABI launches on).  Wraps the product op, does not replace it."""

    def __init__(self):
        from dvis_plus_amd import functions as Fn
        self.Fn, self.orig, self.events, self.frames = Fn, Fn.msda_fused_forward, [], 0

    def __enter__(self):
        def timed(value, *a, **k):
            st = torch.cuda.current_stream(value.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            out = self.orig(value, *a, **k)
            e1.record(st)
            self.events.append((e0, e1, out.shape[0]))          # frames of this launch (the output is (N, Lq, M * D))
            return out
        self.Fn.msda_fused_forward = timed
        return self

    def __exit__(self, *exc):
        self.Fn.msda_fused_forward = self.orig

    def summary(self):
        torch.cuda.synchronize()
        secs = [e0.elapsed_time(e1) / 1e3 for e0, e1, _ in self.events]
        frames = [n for _, _, n in self.events]
        return sum(secs) / max(1, len(secs)), sum(frames) / max(1, len(frames)), len(secs)


class ConvTimer:
    """Times every launch of the own 3x3 convolution in the timed region the same way — the largest convolution kernel of a clip
    (the FPN output convolution and the R50 conv2 layers): csrc/conv1x1_x3.hip's nine-tap form (split-f16 matrix-core
    arithmetic), or with DVIS_X3=0 the exact-fp32 Winograd kernel (csrc/winograd_conv.hip)."""

    def __init__(self):
        from dvis_plus_amd import native, functions
        self.native, self.events, self.x3 = native, [], functions.X3

    def __enter__(self):
        lib = self.native.lib()
        self.lib = lib
        self.name = "dvis_conv3x3_x3" if self.x3 else "dvis_conv3x3_winograd"
        self.orig = getattr(lib, self.name)

        def timed_w(x, uf, bias, y, N, C, K, H, W, relu, stream):
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = self.orig(x, uf, bias, y, N, C, K, H, W, relu, stream)
            e1.record(st)
            self.events.append((e0, e1, 2.0 * 9 * N * C * K * H * W))      # direct-convolution FLOPs of the launch
            return rc

        def timed_x(x, packed, bias, res, y, N, C, K, H, W, stride, *rest):
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = self.orig(x, packed, bias, res, y, N, C, K, H, W, stride, *rest)
            e1.record(st)
            self.events.append((e0, e1, 2.0 * 9 * N * C * K * ((H + stride - 1) // stride) * ((W + stride - 1) // stride)))
            return rc
        setattr(lib, self.name, timed_x if self.x3 else timed_w)
        self.orig1, self.events1 = lib.dvis_conv1x1_x3, []

        def timed_1(x, packed, bias, res, y, N, C, K, H, W, stride, *rest):
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = self.orig1(x, packed, bias, res, y, N, C, K, H, W, stride, *rest)
            e1.record(st)
            self.events1.append((e0, e1, 2.0 * N * C * K * ((H + stride - 1) // stride) * ((W + stride - 1) // stride)))
            return rc
        lib.dvis_conv1x1_x3 = timed_1
        self.orig2 = lib.dvis_conv1x1_x3_dual

        def timed_2(x, x2, packed, bias, res, y, N, C, C2, K, H, W, *rest):        # conv3 + shortcut as one launch of the same kernel
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = self.orig2(x, x2, packed, bias, res, y, N, C, C2, K, H, W, *rest)
            e1.record(st)
            self.events1.append((e0, e1, 2.0 * N * (C + C2) * K * H * W))
            return rc
        lib.dvis_conv1x1_x3_dual = timed_2
        self.orig3 = lib.dvis_conv_x3_image

        def timed_3(ximg, x, packed, bias, res, y, image, N, C, K, H, W, stride, taps, *rest):      # the same kernel with operand images on either side
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = self.orig3(ximg, x, packed, bias, res, y, image, N, C, K, H, W, stride, taps, *rest)
            e1.record(st)
            (self.events if taps == 9 and self.x3 else self.events1).append(
                (e0, e1, 2.0 * taps * N * C * K * ((H + stride - 1) // stride) * ((W + stride - 1) // stride)))
            return rc
        lib.dvis_conv_x3_image = timed_3
        return self

    def __exit__(self, *exc):
        setattr(self.lib, self.name, self.orig)
        self.lib.dvis_conv1x1_x3 = self.orig1
        self.lib.dvis_conv1x1_x3_dual = self.orig2
        self.lib.dvis_conv_x3_image = self.orig3

    def summary(self):
        torch.cuda.synchronize()
        secs = sum(e0.elapsed_time(e1) / 1e3 for e0, e1, _ in self.events)
        return secs, sum(f for _, _, f in self.events), len(self.events)

    def summary_kernel(self):
        """every launch of conv1x1_x3_kernel in the timed region (1 x 1 and nine-tap forms): the kernel a clip spends most time in"""
        torch.cuda.synchronize()
        ev = (self.events if self.x3 else []) + self.events1
        secs = sum(e0.elapsed_time(e1) / 1e3 for e0, e1, _ in ev)
        return secs, sum(f for _, _, f in ev), len(ev)


DTYPE_NOTE = ("f32 (storage, accumulation, results; the segmenter's dense layers - R50 convolutions from 128 channels on, pixel-decoder "
              "projections and FPN output convolution, the deformable encoder's projections and FFN, the decoder's key / value "
              "projections - multiply each fp32 operand as two f16 terms on the f16 matrix cores, 3 products per pair: error vs fp64 "
              "<= the fp32 GEMM's, tests/test_gemm_x3_gpu.py; exact_f32 = the same run with DVIS_X3=0)")


class FfnTimer:
    """Times every launch of the encoder's fused FFN kernel (csrc/gemm_x3.hip, dvis_x3_ffn_ln) in the timed region the same
    way: the largest kernel of a clip by arithmetic (2 x 579 600 x 256 x 1024 x 2 flops per layer at T = 30)."""

    def __init__(self):
        from dvis_plus_amd import native
        self.native, self.events = native, []

    def __enter__(self):
        lib = self.native.lib()
        self.lib, self.orig = lib, lib.dvis_x3_ffn_ln

        def timed(x, ldx, M, K, H, N, *rest):
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = self.orig(x, ldx, M, K, H, N, *rest)
            e1.record(st)
            self.events.append((e0, e1, 2.0 * M * H * (K + N)))
            return rc
        lib.dvis_x3_ffn_ln = timed
        return self

    def __exit__(self, *exc):
        self.lib.dvis_x3_ffn_ln = self.orig

    def summary(self):
        torch.cuda.synchronize()
        secs = sum(e0.elapsed_time(e1) / 1e3 for e0, e1, _ in self.events)
        return secs, sum(f for _, _, f in self.events), len(self.events)


class LibTimer:
    """Times every launch of one C-ABI entry point in the timed region with HIP events on the launch stream (as MsdaTimer).
    `account(*args)` -> dict of additive per-launch quantities (flops, bytes, ...) or None to leave the launch untimed."""

    def __init__(self, name, account):
        from dvis_plus_amd import native
        self.native, self.name, self.account, self.events = native, name, account, []

    def __enter__(self):
        self.lib = self.native.lib()
        self.orig = getattr(self.lib, self.name)

        def timed(*a):
            acc = self.account(*a)
            if acc is None:
                return self.orig(*a)
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = self.orig(*a)
            e1.record(st)
            self.events.append((e0, e1, acc))
            return rc
        setattr(self.lib, self.name, timed)
        return self

    def __exit__(self, *exc):
        setattr(self.lib, self.name, self.orig)

    def summary(self, pred=lambda acc: True):
        """-> (seconds, summed quantities, launches) over the launches whose accounting dict satisfies `pred`."""
        torch.cuda.synchronize()
        tot, secs, n = {}, 0.0, 0
        for e0, e1, acc in self.events:
            if pred(acc):
                secs += e0.elapsed_time(e1) / 1e3
                n += 1
                for k, v in acc.items():
                    if isinstance(v, (int, float)):
                        tot[k] = tot.get(k, 0.0) + v
        return secs, tot, n


def _acct_mask_logits(pe, pf, B, Q, C, HW, out, stream):
    # flops 2 Q C HW, bytes 4 (Q C + C HW + Q HW) per frame
    return {"flops": 2.0 * B * Q * C * HW, "bytes": 4.0 * B * (Q * C + C * HW + Q * HW), "frames": B, "queries": Q, "form": 0}


def _acct_mask_pooled(pe, pf, B, Q, C, h, w, mask, allowed, stream):
    # the decoder's attention masks on the level's pooled map: fp32 operands in, ONE byte per (query, pixel) + a count per query out
    return {"flops": 2.0 * B * Q * C * h * w, "bytes": 4.0 * B * (Q * C + C * h * w) + 1.0 * B * Q * h * w + 4.0 * B * Q, "frames": B,
            "queries": Q, "form": 2}


def _acct_bneck(a1, res, x2, packed, b2, b3, b1, y, out, N, H, W, *rest):
    # csrc/bneck_x3.hip, one res2 bottleneck from conv1's map on (+ the next block's conv1): algorithmic HBM bytes per pixel =
    # the 64-channel operand image in (256 B) + the shortcut in (1024 B identity, 256 B projection input) + the block output
    # (1024 B) + the next operand image (256 B); flops = 2 (9 64 64 + 64 256 [+ 64 256] + [256 64]) per pixel
    px = float(N) * H * W
    proj, tail = bool(x2), bool(out)
    return {"bytes": px * (256 + (256 if proj else 1024) + 1024 + (256 if tail else 0)),
            "flops": 2.0 * px * (9 * 64 * 64 + 64 * 256 + (64 * 256 if proj else 0) + (256 * 64 if tail else 0)), "frames": N}


def _acct_attention(q, qs, k, ks, v, vs, out, os_, mask, allowed, B, heads, Lq, Lk, d, scale, ws, stream, kernel):
    # QK^T + PV: 4 Lq Lk d per (batch, head); bytes: q, out (Lq rows), k, v (Lk rows) of heads * d floats + the byte mask
    masked = bool(mask)
    return {"flops": 4.0 * B * heads * Lq * Lk * d, "bytes": 4.0 * B * heads * d * (2 * Lq + 2 * Lk) + (1.0 * B * Lq * Lk if masked else 0.0),
            "masked": masked, "Lk": Lk, "kernel": int(kernel), "long": Lk > 128}


def cpu_baseline_msda(budget_s=8.0):
    """Oracle (C port, OpenMP) of the MSDA forward on a bounded sample: one 720p frame-layer per call."""
    from oracle import msda as omsda
    torch.manual_seed(0)
    shapes = torch.tensor([(23, 40), (46, 80), (92, 160)], dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = 19320
    value = torch.randn(1, S, 8, 32)
    loc = torch.rand(1, S, 8, 3, 4, 2)
    w = torch.softmax(torch.randn(1, S, 8, 12), -1).view(1, S, 8, 3, 4)
    omsda.msda_forward(value, shapes, lsi, loc, w)
    n, t0 = 0, time.time()
    while time.time() - t0 < budget_s and n < 200:
        omsda.msda_forward(value, shapes, lsi, loc, w)
        n += 1
    dt = (time.time() - t0) / n
    return {"value": round(1.0 / dt, 3), "unit": "frame-layers/s (MSDeformAttn fwd, 720p)",
            "sample": f"{n} calls of one 736x1280 frame-layer (S=Lq=19320, M=8, D=32, L=3, P=4), C/OpenMP oracle, "
                      f"{dt * 1e3:.1f} ms each"}


def cpu_baseline(model, clip, frames=3, thr=0.8, windows=3):
    """The metric's own unit on the host cores: the oracle's restatement of the reference pipeline (windowed,
    frame-by-frame tracker, torch ops; oracle/dvis_torch.py) on a bounded sample = the first `frames` frames of the same
    synthetic clip = one window of the reference's loop (TEST.WINDOW_SIZE = 3), same weights, backbone = the same torch
    modules on the CPU.  One warm-up window, `windows` timed windows (median reported, p10 / p50 / p90 beside it)."""
    import copy
    from oracle import dvis_torch as O
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    sd["pixel_mean"], sd["pixel_std"] = model.pixel_mean.cpu(), model.pixel_std.cpu()
    backbone = copy.deepcopy(model.backbone).cpu().eval()
    sample = [f for f in clip[:frames].cpu()]
    # host threads: torch's default on the GPU box (128 of its 256 logical CPUs) runs the oracle's 3-frame window 2.3x SLOWER than 32
    # threads (profiles/r06_oracle_threads.txt: 14.4 s vs 5.8 s for pixel decoder + decoder); the baseline uses the faster setting
    prev_threads = torch.get_num_threads()
    if (os.cpu_count() or 0) >= 64 and "DVIS_CPU_BASELINE_THREADS" not in os.environ:
        torch.set_num_threads(32)
    elif "DVIS_CPU_BASELINE_THREADS" in os.environ:
        torch.set_num_threads(int(os.environ["DVIS_CPU_BASELINE_THREADS"]))
    threads = torch.get_num_threads()

    def run():
        t0 = time.time()
        with torch.no_grad():
            O.dvis_plus_forward(sd, backbone, sample, offline=True, nheads=8, enc_layers=6, dec_layers=9,
                                tracker_layers=6, refiner_layers=6, window_size=3, num_classes=124, n_things=58,
                                task="vps", object_mask_threshold=thr, overlap_threshold=0.8, out_hw=(720, 1280))
        return time.time() - t0
    warm = run()                               # thread pools, oneDNN primitives, allocator
    dts = sorted(run() for _ in range(windows))      # 1 warm-up + 3 timed, median and p10 / p90
    pick = lambda q: dts[min(len(dts) - 1, int(q * len(dts)))]
    med = dts[len(dts) // 2]
    msda_op = cpu_baseline_msda()
    torch.set_num_threads(prev_threads)
    return {"value": round(frames / med, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "p10_p50_p90": [round(frames / pick(0.9), 4), round(frames / med, 4), round(frames / pick(0.1), 4)],
            "window_seconds": [round(d, 2) for d in dts],
            "sample": f"first {frames} frames (one reference window, TEST.WINDOW_SIZE=3) of the same 720p synthetic clip "
                      f"through oracle/dvis_torch.py (fp32 torch CPU ops, {threads} threads): 1 warm-up window "
                      f"({warm:.1f} s) + {windows} timed windows (median {med:.1f} s); value = frames / median window time"
                      f"; {threads} threads is the fastest setting on this host (torch's default of {prev_threads} is slower)",
            "msda_op": msda_op}


def load_x3_traffic():
    """HBM bytes per launch of the split-f16 kernels from the committed rocprofv3 PMC passes (bench.py cannot read PMCs itself):
    profiles/rNN_x3_traffic.json, written by tools/x3_traffic_json.py from the raw counter summaries (separate --pmc passes for
    FETCH_SIZE and WRITE_SIZE, the guide's gfx950 corrections).  {kernel family: {"hbm_bytes_per_launch", "source", ...}}"""
    for name in ("r06b_x3_traffic.json", "r06_x3_traffic.json", "r05_x3_traffic.json"):
        tj = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tj):
            try:
                return json.load(open(tj))
            except ValueError:
                pass
    return {}


def round_sizes(n_clips, per_round):
    """stream() takes clips in rounds of `per_round`: sizes of the rounds for n_clips clips."""
    return [min(per_round, n_clips - i) for i in range(0, n_clips, per_round)]


def warmup_clip_count(warmup, steps, world, owner_rounds, streamed, tracker_batch=1):
    """Clips to run untimed so that EVERY segmenter batch shape of the timed pass has been seen (MIOpen searches its
    solvers, seconds per shape, the first time a convolution shape appears).  One clip per step on a single GPU.  With
    several ranks stream() batches a rank's frames of a whole ROUND of clips into one segmenter call — `world` clips with
    tracker-owner rounds, `tracker_batch` clips with the replicated tracker — so the shapes are those of a full round and of
    the last, partial round (steps mod round size clips): the warm-up replays exactly that round structure."""
    size = world if owner_rounds else tracker_batch
    if not (streamed and world > 1 and size > 1):
        return warmup
    r, full = steps % size, steps >= size
    k = max(1 if full else 0, -(-(warmup - r) // size))          # smallest k with k * size + r >= warmup
    if r == 0:
        k = max(k, 1)
    return k * size + r


def calibrate_threshold(model, inputs, candidates, slack=0):
    """Random-init class logits are near-uniform (max prob ~ 1/125), so the reference's 0.8 score threshold would keep no
    query and the panoptic stage would be skipped.  One untimed pass finds the score threshold that sends `candidates`
    queries of THIS clip there (mid-way between the k-th and the (k+1)-th best non-void score).  slack > 0 (comparisons
    between schedules / against the oracle): k anywhere in candidates +- slack, at the LARGEST gap between neighbouring scores
    — neighbouring random-init scores can sit 1e-7 apart, and a threshold between two such scores is decided by rounding."""
    from dvis_plus_amd import postprocess as PP
    seen = {}
    orig_sel = PP.vps_select

    def spy(pred_cls, num_classes, thr, aux=None):
        scores, labels, keep = orig_sel(pred_cls, num_classes, thr, aux)
        seen["s"] = scores[labels.ne(num_classes)].sort(descending=True)[0]
        return scores, labels, keep
    PP.vps_select = spy
    probe = [dict(inputs[0], object_mask_threshold=2.0)]     # keep nothing: cheap calibration pass
    try:
        model(probe)
    finally:
        PP.vps_select = orig_sel
    s = seen["s"]
    if candidates >= s.numel():
        return 0.0                             # every non-void query
    k = max(0, min(candidates, s.numel() - 1))
    if slack > 0 and k > 0:
        lo, hi = max(1, k - slack), min(s.numel() - 1, k + slack)
        gaps = s[lo - 1:hi] - s[lo:hi + 1]
        k = lo + int(gaps.argmax())
    return float((s[k - 1] + s[k]) / 2) if k > 0 else 2.0


def stage_breakdown(model, video, task):
    """One clip with a device synchronisation after every stage (ms).  Untimed extra pass, offline mode."""
    from dvis_plus_amd import postprocess as PP
    acc = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        acc[name] = round((time.perf_counter() - t0) * 1e3, 2)
        return out
    m = model
    with torch.no_grad():
        images, img_size = timed("preprocess", lambda: m.preprocess(video["image"]))
        feats = timed("backbone", lambda: m.backbone(images))
        mf, _, ms = timed("pixel_decoder", lambda: m.sem_seg_head.pixel_decoder.forward_features(feats))
        e, e_nn, lg = timed("decoder", lambda: m.decode(ms, mf))
        to_bctq = lambda z: z.permute(2, 0, 1).unsqueeze(0)
        track = timed("tracker", lambda: m.tracker(to_bctq(e), None, resume=False, frame_embeds_no_norm=to_bctq(e_nn),
                                                   need_masks=False))
        if m.refiner is None:
            return acc
        ref = timed("refiner", lambda: m.refiner(track["pred_embds"], to_bctq(e_nn), None, need_masks=False))
        cls, aux = PP.mean_logits(ref["pred_logits"], track["pred_logits"])
        mask_fn = lambda idx: m.refiner.predict_masks(ref["mask_embed"], mf.unsqueeze(0), idx)[0]
        timed("masks+postprocess", lambda: m._task_output(cls, aux, mask_fn, img_size, (720, 1280), images.shape[-2:],
                                                          len(images), video))
    return acc


def other_configurations(model, clips, device, args):
    """BASELINE.json's other single-GPU-runnable configurations as short passes AFTER the headline's timed region (same protocol:
    inputs resident in HBM, warm-up, K clips between synchronisations): #2 online T=5, #4's clip length T=64 (on one GPU, unsharded),
    #5 ViT-Adapter-L with 200 queries.  -> {name: {"value", "unit", "steps", "ms_per_step", "workload"}}"""
    from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
    out = {}

    def timed(m, vids, warm, streamed):
        def run(vs):
            if streamed:
                for _ in m.stream(vs):
                    pass
            else:
                for v in vs:
                    m([v])
        run(vids[:warm])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(vids)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def videos_of(m, frames_list):
        vids = [{"image": f, "height": 720, "width": 1280} for f in frames_list]
        m.allow_input_threshold = True
        if args.task == "vps":
            for v in vids:
                v["object_mask_threshold"] = calibrate_threshold(m, [v], args.candidates * m.num_queries // 100)
        return vids

    # ---- config #2: DVIS++ online R50, T = 5 windows (the segmenter of a window replays from a hipGraph)
    m2 = build_dvis_plus_r50("online", task=args.task, object_mask_threshold=0.0, num_queries=100).to(device)
    wins = [clips[i % len(clips)][5 * (i // len(clips)):5 * (i // len(clips)) + 5] for i in range(8)]
    vids = videos_of(m2, wins)
    dt = timed(m2, vids, 2, False)
    out["online_T5"] = {"value": round(5 * len(vids) / dt, 3), "unit": "frames/s", "steps": len(vids), "ms_per_step": round(dt / len(vids) * 1e3, 2),
                        "workload": "BASELINE config #2: DVIS++ online R50, T=5 720p windows, 100 queries, one forward() per window"}
    del m2, vids, wins
    # ---- config #4's clip length on one GPU: T = 64 through the headline model
    long_clips = [synthetic_clip(64, device, seed=4321 + i) for i in range(2)]
    vids = videos_of(model, long_clips) * 2
    dt = timed(model, vids[:3], 1, True)
    out["offline_T64"] = {"value": round(64 * 3 / dt, 3), "unit": "frames/s", "steps": 3, "ms_per_step": round(dt / 3 * 1e3, 2),
                          "workload": "BASELINE config #4's clip (DVIS++ offline R50, T=64 720p, refiner on) unsharded on one GPU, stream()"}
    del vids, long_clips
    torch.cuda.empty_cache()
    # ---- config #5: ViT-Adapter-L backbone, 200 queries, T = 30
    m5 = build_dvis_plus_r50("offline", task=args.task, object_mask_threshold=0.0, backbone="vitl", num_queries=200).to(device)
    vids = videos_of(m5, clips[:3])
    dt = timed(m5, vids, 1, True)
    out["vitl_200q_T30"] = {"value": round(30 * len(vids) / dt, 3), "unit": "frames/s", "steps": len(vids),
                            "ms_per_step": round(dt / len(vids) * 1e3, 2),
                            "workload": "BASELINE config #5: DVIS++ offline DINOv2 ViT-L / ViT-Adapter, T=30 720p, 200 queries, on one GPU, stream()"}
    del m5, vids
    torch.cuda.empty_cache()
    return out


BB_NAME = {"r50": "R50", "vitl": "ViT-Adapter-L", "vitb": "ViT-Adapter-B"}


def launch_command(n, argv, n_devices, port):
    """`python bench.py --gpus N` without a launcher: the command + environment that re-runs this script as N ranks of one
    node (what detectron2's launch(main, num_gpus) does for the reference, train_net_video.py:322-329).  With fewer than N
    GPUs visible (development boxes) the ranks share device 0 and talk through gloo — a functional check of the sharded
    path, flagged in the JSON line, never a scaling number."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if n_devices < n:
        env.update(DVIS_BENCH_ONE_DEVICE="1", DVIS_DIST_BACKEND="gloo")
    return cmd, env


def self_launch(n):
    import socket
    import subprocess
    with socket.socket() as sk:                    # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd, env = launch_command(n, sys.argv[1:], torch.cuda.device_count(), port)
    print("bench.py: launching", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--task", default="vps")
    ap.add_argument("--mode", default="offline", choices=["offline", "online"],
                    help="offline = BASELINE headline config (T=30, refiner on); online = config #2 (use --frames 5)")
    ap.add_argument("--candidates", type=int, default=20, help="queries sent to the panoptic stage (see main)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed stage breakdown and the 100-candidate pass")
    ap.add_argument("--no-configs", action="store_true", help="skip the short passes of BASELINE configs #2 / #4-length / #5")
    ap.add_argument("--backbone", default="r50", choices=["r50", "vitl", "vitb"],
                    help="r50 = the headline config; vitl = BASELINE config #5 (ViT-Adapter-L, use --queries 200)")
    ap.add_argument("--queries", type=int, default=100)
    ap.add_argument("--segmenter-chunk", type=int, default=0, help="frames per segmenter call (0 = all local frames)")
    ap.add_argument("--clip-stream", type=int, default=1, choices=[0, 1],
                    help="offline mode: 1 (default) = the K timed clips go through model.stream(): the segmenter of clip "
                         "i+1 is enqueued before the tracker / refiner / post-processing of clip i, which run on a second "
                         "stream (all K clips complete inside the timed region); 0 = one forward() per clip, back to back")
    ap.add_argument("--rounds", type=int, default=0,
                    help="offline mode: spans handed to the tracker while the segmenter runs the next span (0 = default)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))           # one rank per GPU under torch.distributed.run; rank 0 prints the line
    # a hung collective must not hang the box: dump every thread's Python stack and exit after this many seconds
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("DVIS_BENCH_WATCHDOG", "1500")), exit=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist_on = world > 1 or os.environ.get("DVIS_FORCE_COLLECTIVES") == "1"   # dev aid: RCCL calls on a single rank
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    # DVIS_BENCH_ONE_DEVICE=1 + DVIS_DIST_BACKEND=gloo: development aid to exercise the sharded pipeline with several
    # ranks on a single-GPU box (all ranks on cuda:0, collectives through gloo).  Never used for reported numbers.
    one_device = os.environ.get("DVIS_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("DVIS_DIST_BACKEND", "nccl")               # "nccl" = RCCL over xGMI
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)
        else:
            torch.distributed.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run or let bench.py do it)"

    if os.environ.get("DVIS_MIOPEN_FIND", "0") == "1":
        torch.backends.cudnn.benchmark = True      # MIOpen exhaustive find per conv shape (first call of a shape is slow)
    from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
    # Random-init class logits are near-uniform (max prob ~ 1/125), so the reference's 0.8 score threshold would
    # keep no query and post-processing would be skipped; the threshold is calibrated below so that --candidates
    # (20) queries reach the panoptic stage (representative work).  Everything else follows
    # VIPSeg/DVIS_Plus_Offline_R50.yaml.
    model = build_dvis_plus_r50(args.mode, task=args.task, object_mask_threshold=0.0, backbone=args.backbone,
                                num_queries=args.queries, segmenter_chunk=args.segmenter_chunk).to(device)
    model.allow_input_threshold = True     # each synthetic clip carries its own calibrated score threshold (see below)
    if args.rounds:
        model.pipeline_rounds = args.rounds
    T = args.frames
    # step i runs on its own clip (seed 1234 + i); all of them resident in HBM before timing
    clips = [synthetic_clip(T, device, seed=1234 + i) for i in range(max(1, args.steps))]
    videos = [{"image": c, "height": 720, "width": 1280} for c in clips]
    streamed = bool(args.clip_stream and args.mode == "offline")

    # calibration (untimed, per clip): the score threshold that sends args.candidates queries of THAT clip to the
    # panoptic stage — random-init class scores are near-uniform, so a threshold does not transfer between clips (one
    # calibrated on clip 0 lets anything between 0 and 99 queries of the other clips through)
    if args.task == "vps":
        for v in videos:
            v["object_mask_threshold"] = calibrate_threshold(model, [v], args.candidates)

    def run_pass(vids, latencies=None):
        """All of `vids` through the model.  latencies: list that receives (start event, end event) per clip."""
        outs = []
        if streamed:
            starts = []

            def feed():
                for v in vids:
                    if latencies is not None:
                        e = torch.cuda.Event(enable_timing=True)
                        e.record()
                        starts.append(e)
                    yield v
            for i, out in enumerate(model.stream(feed())):
                if latencies is not None:
                    latencies.append((starts[i], out["ready_event"]))
                outs.append(out)
        else:
            for v in vids:
                if mark:
                    torch.empty(64, device=device).uniform_()
                if latencies is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                outs.append(model([v]))
                if latencies is not None:
                    e1.record()
                    latencies.append((e0, e1))
        return outs

    # Warm-up must see every convolution shape of the timed region: MIOpen searches its solvers (naive kernels
    # included, seconds per shape) the first time a shape appears.  One clip per step on a single GPU; with several
    # ranks stream() batches this rank's frames of a whole round of `world` clips into one segmenter call, so the
    # shapes are those of a full round and of the last, partial round (K mod world clips, same rotation of the ragged
    # split as in the timed pass) — warm up with exactly that sequence.
    mark = os.environ.get("DVIS_BENCH_MARK") == "1"     # tools/steady_stats.py: marker kernel at each timed step start

    def timed(owner_rounds):
        """W warm-up clips (round structure of the timed pass), then EXACTLY K clips between barrier + synchronize on both
        sides; max over ranks.  -> (seconds, outputs, latencies, MSDA timer, warm-up clips run)."""
        model.owner_rounds = owner_rounds
        model.stream_timing = False
        warm = warmup_clip_count(args.warmup, args.steps, world, owner_rounds, streamed, model.tracker_batch)
        run_pass([videos[i % len(videos)] for i in range(warm)])
        torch.cuda.synchronize()
        if dist_on:
            torch.distributed.barrier()
        model.stream_timing = True
        tm, lt = MsdaTimer(), []
        tm.conv, tm.ffn = ConvTimer(), FfnTimer()
        tm.mask0, tm.mask2 = LibTimer("dvis_mask_logits", _acct_mask_logits), LibTimer("dvis_attn_mask_pooled", _acct_mask_pooled)
        tm.attn = LibTimer("dvis_attention_forward_k", _acct_attention)
        tm.bneck = LibTimer("dvis_bneck_x3", _acct_bneck)
        t0 = time.perf_counter()
        with tm, tm.conv, tm.ffn, tm.mask0, tm.mask2, tm.attn, tm.bneck:
            res_ = run_pass(videos[:args.steps], lt)
        torch.cuda.synchronize()
        if dist_on:
            torch.distributed.barrier()
        secs = time.perf_counter() - t0
        if dist_on:
            tt = torch.tensor([secs], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            secs = tt.item()
        return secs, res_, lt, tm, warm

    # N > 1: the HEADLINE is north_star's split — frames sharded, ONE all-gather of the per-frame queries per clip, tracker
    # + refiner replicated on every rank (stream(): one clip per round, their sequential kernels overlap the next clip's
    # segmenter) — and a second timed pass measures stream()'s tracker-owner rounds (clip j of a round of `world` clips is
    # tracked by rank j alone; one more all-gather per round) as an extra key.
    owner_default = model.owner_rounds
    if world > 1 and streamed and "DVIS_TRACKER_BATCH" not in os.environ:
        # replicated tracker: several clips per round advance through ONE tracker pass (same results per clip) and a rank's
        # frames of those clips are one segmenter batch — the replicated recurrence is otherwise the critical path.  Per rank,
        # measured on one GPU with the collective replaced by a copy (tools/rank_emulation.py, profiles/r03_rank_emulation*):
        # 4 ranks 0.65 / 0.72 / 0.77 of linear at 1 / 2 / 4 clips per round, 8 ranks 0.47 / 0.56 / 0.61; 2 ranks 0.86 / 0.92 / 0.92
        model.tracker_batch = 4      # (final commit: 2 ranks 0.81 / 0.90 / 0.92, 4 ranks 0.61 / 0.71 / 0.77, 8 ranks 0.44 / 0.55 / 0.62)
    dt, outs, lat, timer, warm_clips = timed(owner_rounds=False if world > 1 else owner_default)
    owner_line = None
    if world > 1 and streamed and owner_default:
        dt_o, outs_o, lat_o, _, warm_o = timed(owner_rounds=True)
        owner_line = {"value": round(args.frames * args.steps / dt_o, 3), "unit": "frames/s",
                      "ms_per_step": round(dt_o / args.steps * 1e3, 2), "warmup_clips_run": warm_o,
                      "collectives_per_round": f"{world} query all-gathers + 1 result all-gather (+ VPS area all-reduces)",
                      "note": "stream() in rounds of `world` clips, clip j's tracker + refiner on rank j only"}
        model.owner_rounds = False
    out = outs[-1]
    ncands = [float(o.get("num_candidates") or 0) for o in outs]
    dist_info = None
    if dist_on:
        import socket
        mine = {"rank": rank, "device": f"cuda:{dev_index}", "name": torch.cuda.get_device_name(dev_index),
                "pci_bus": getattr(torch.cuda.get_device_properties(dev_index), "pci_bus_id", None),
                "host": socket.gethostname()}
        per_rank = [None] * torch.distributed.get_world_size()
        torch.distributed.all_gather_object(per_rank, mine)
        if not one_device:
            # a real multi-GPU run: one rank per GPU over RCCL, or it is not the measurement the line claims to be
            bus = [r["pci_bus"] for r in per_rank]
            ok = torch.distributed.get_backend() == "nccl" and len({(r["host"], r["device"]) for r in per_rank}) == len(per_rank) \
                and (None in bus or len({(r["host"], b) for r, b in zip(per_rank, bus)}) == len(per_rank))
            if not ok:
                sys.stderr.write(f"bench.py: N > 1 needs backend nccl (RCCL) and one distinct GPU per rank; got backend "
                                 f"{torch.distributed.get_backend()}, ranks {per_rank}\n")
                sys.exit(3)
        dist_info = {"world_size": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend(),
                     "ranks": per_rank, "ranks_share_one_gpu": bool(one_device),
                     "split": "frames sharded contiguously (fixed split), one all-gather of per-frame queries "
                              "per clip, tracker + refiner replicated",
                     "tracker_batch": model.tracker_batch}

    # second, short timed pass: every non-void query goes to the panoptic stage (the upper end of the work that the
    # candidate count controls).  Same protocol, up to 4 clips; single GPU only.
    cand100 = None
    if args.task == "vps" and world == 1 and not dist_on and args.candidates < args.queries and not args.no_extra:
        n2 = min(4, args.steps)
        all_q = [dict(v, object_mask_threshold=0.0) for v in videos[:n2]]
        run_pass(all_q[:1])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        o2 = run_pass(all_q)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        cand100 = {"value": round(T * n2 / dt2, 3), "unit": "frames/s", "steps": n2,
                   "panoptic_candidates": [int(o.get("num_candidates") or 0) for o in o2]}

    # third, short timed pass: the encoder's GEMMs on the exact-fp32 matrix instructions / library GEMMs (DVIS_X3=0) instead of
    # the split-f16 kernels of csrc/gemm_x3.hip — the same protocol on up to 6 clips
    exact = None
    from dvis_plus_amd import functions as _Fn
    if world == 1 and not dist_on and not args.no_extra and _Fn.X3:
        n3 = min(6, args.steps)
        _Fn.X3 = False
        try:
            run_pass(videos[:2])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_pass(videos[:n3])
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t1
        finally:
            _Fn.X3 = True
        exact = {"value": round(T * n3 / dt3, 3), "unit": "frames/s", "steps": n3, "ms_per_step": round(dt3 / n3 * 1e3, 2),
                 "note": "DVIS_X3=0: every GEMM / convolution on v_mfma_f32_*_f32 or the fp32 library GEMM"}

    # fourth, short timed pass: the same clips one forward() at a time, back to back (no overlap of clip i's tracker / refiner /
    # post-processing with clip i + 1's segmenter): BASELINE.md section 3's "frames/s = T / clip latency" reading of the metric
    cbc = None
    if world == 1 and not dist_on and not args.no_extra and streamed:
        n4 = min(6, args.steps)
        streamed = False
        try:
            run_pass(videos[:1])
            torch.cuda.synchronize()
            lt4 = []
            t1 = time.perf_counter()
            run_pass(videos[:n4], lt4)
            torch.cuda.synchronize()
            dt4 = time.perf_counter() - t1
        finally:
            streamed = True
        l4 = sorted(e0.elapsed_time(e1) for e0, e1 in lt4)
        cbc = {"value": round(T * n4 / dt4, 3), "unit": "frames/s", "steps": n4, "ms_per_step": round(dt4 / n4 * 1e3, 2),
               "latency_ms_p50": round(l4[len(l4) // 2], 2), "note": "--clip-stream 0: one forward() per clip, nothing overlapped"}

    timed_clips = args.steps               # clips the kernel timers saw
    if rank == 0 and not timer.events:
        timed_clips = 2
        # online mode replays the window's segmenter from a hipGraph: its launches do not pass the Python front-ends the timers
        # wrap.  One more (untimed) window launched eagerly, under the same timers, prices the kernels.
        prev = os.environ.get("DVIS_SEGMENTER_GRAPH")
        os.environ["DVIS_SEGMENTER_GRAPH"] = "0"
        try:
            with timer, timer.conv, timer.ffn, timer.mask0, timer.mask2, timer.attn, timer.bneck:
                run_pass(videos[:2])
        finally:
            os.environ.pop("DVIS_SEGMENTER_GRAPH") if prev is None else os.environ.__setitem__("DVIS_SEGMENTER_GRAPH", prev)
    if rank == 0:
        fps = T * args.steps / dt
        sec, nfr, nlaunch = timer.summary()
        achieved = MSDA_BYTES_PER_FRAME_LAYER * nfr / sec / 1e9
        traffic, traffic_src = None, None     # HBM bytes per launch from the committed rocprofv3 PMC passes (bench.py cannot read PMCs itself)
        for name in ("r06b_msda_traffic.json", "r06_msda_traffic.json", "r05_msda_traffic.json", "r03_msda_traffic.json", "r02_msda_traffic.json", "r01_msda_traffic.json"):
            tj = os.path.join(ROOT, "profiles", name)
            if os.path.exists(tj):
                t = json.load(open(tj))
                traffic = round(t["hbm_bytes_per_launch"] * nfr / t["frames_per_launch"])
                traffic_src = f"profiles/{name} (rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE)"
                break
        csec, cflops, claunch = timer.conv.summary()
        conv_roof = None
        if claunch and timer.conv.x3:
            # nine taps, 3 f16 matrix-core products per fp32 product
            tf16 = 3.0 * cflops / csec / 1e12
            conv_roof = {"bound": "mfma", "kernel": "conv3x3_x3 (3x3 convolutions from 128 channels on: FPN output conv + R50 conv2, nine-tap "
                                                    "split-f16 kernel)",
                         "achieved": round(tf16, 1), "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf16 / MFMA_F16_PEAK_TF, 4),
                         "fp32_equivalent_tflops": round(cflops / csec / 1e12, 1), "launches_timed": claunch,
                         "ms_per_clip": round(csec / timed_clips * 1e3, 2), "traffic": None,
                         "note": "f16 matrix-core flops issued = 3 x 2*9*N*C*K*OH*OW per launch, summed over the timed launches / their "
                                 "summed HIP-event durations (the exact-fp32 Winograd kernel it replaces ran these layers at 216 - 264 "
                                 "TF direct-equivalent)"}
        elif claunch:
            # the kernel's own arithmetic: F(2x2, 3x3) does 4 multiply-adds per output where the direct form does 9
            tf = cflops / 2.25 / csec / 1e12
            conv_roof = {"bound": "mfma", "kernel": "winograd_f2x3 (3x3 convolutions: FPN output conv + R50 conv2)",
                         "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 4),
                         "direct_equivalent_tflops": round(cflops / csec / 1e12, 1), "launches_timed": claunch,
                         "ms_per_clip": round(csec / timed_clips * 1e3, 2), "traffic": None,
                         "note": "flops = 2 * 4 * N*C*K*H*W per launch (Winograd multiplies), summed over the timed launches / "
                                 "their summed HIP-event durations"}
        ksec, kflops, klaunch = timer.conv.summary_kernel()
        convk_roof = None
        if klaunch:
            tf16 = 3.0 * kflops / ksec / 1e12
            convk_roof = {"bound": "mfma", "kernel": "conv1x1_x3_kernel, every launch (1x1 and nine-tap 3x3 forms: R50, pixel-decoder "
                                                     "projections, FPN output conv) - the kernel with the largest share of a clip's time",
                          "achieved": round(tf16, 1), "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf16 / MFMA_F16_PEAK_TF, 4),
                          "fp32_equivalent_tflops": round(kflops / ksec / 1e12, 1), "launches_timed": klaunch,
                          "ms_per_clip": round(ksec / timed_clips * 1e3, 2), "traffic": None,
                          "note": "f16 matrix-core flops issued = 3 x 2*taps*N*C*K*OH*OW per launch, summed over the timed launches / their "
                                  "summed HIP-event durations; includes the HBM-bound layers (64 / 128 input channels at the large maps)"}
        fsec, fflops, flaunch = timer.ffn.summary()
        ffn_roof = None
        if flaunch:
            # the kernel issues 3 f16 matrix-core products per fp32 product (a_lo w_hi + a_hi w_lo + a_hi w_hi)
            tf16 = 3.0 * fflops / fsec / 1e12
            ffn_roof = {"bound": "mfma", "kernel": "x3_ffn (encoder FFN: linear1 + ReLU + linear2 + residual + LayerNorm, one kernel)",
                        "achieved": round(tf16, 1), "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf16 / MFMA_F16_PEAK_TF, 4),
                        "fp32_equivalent_tflops": round(fflops / fsec / 1e12, 1), "launches_timed": flaunch,
                        "ms_per_clip": round(fsec / timed_clips * 1e3, 2), "traffic": None,
                        "note": "f16 matrix-core flops issued = 3 x the fp32 GEMM flops 2*M*H*(K+N) per launch, summed over the "
                                "timed launches / their summed HIP-event durations; fp32_equivalent = the fp32 GEMM flops / time "
                                "(the fp32 matrix peak is 157.3)"}
        x3_traffic = load_x3_traffic()
        for roof, key in ((convk_roof, "conv1x1_x3_kernel"), (conv_roof, "conv1x1_x3_kernel"), (ffn_roof, "x3_ffn_kernel")):
            if roof is not None:
                # the split-f16 kernels issue 3 matrix-core products per algorithmic one: alg_frac prices the ALGORITHMIC flops
                #  against the same dense f16 peak
                roof["alg_frac"] = round(roof["fp32_equivalent_tflops"] / MFMA_F16_PEAK_TF, 4) if "fp32_equivalent_tflops" in roof else None
                t = x3_traffic.get(key)
                if t is not None:
                    roof["traffic"] = t["hbm_bytes_per_launch"]
                    roof["traffic_source"] = t["source"]
        mask_roof = attn_roof = None
        m0s, m0, m0n = timer.mask0.summary()
        m2s, m2, m2n = timer.mask2.summary()
        if m0n or m2n:
            def entry(secs, tot, n, what):
                gbs, tf = tot["bytes"] / secs / 1e9, tot["flops"] / secs / 1e12
                return {"form": what, "achieved": round(gbs, 1), "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "tflops_f32": round(tf, 1),
                        "mfma_f32_frac": round(tf / MFMA_F32_PEAK_TF, 4), "launches_timed": n, "us_per_launch": round(secs / n * 1e6, 1),
                        "alg_bytes_per_launch": int(tot["bytes"] / n), "flops_per_launch": int(tot["flops"] / n),
                        "ms_per_clip": round(secs / timed_clips * 1e3, 3)}
            forms = {}
            if m0n:
                forms["mask_logits"] = entry(m0s, m0, m0n, "einsum(bqc,bchw->bqhw) at stride 4 for the queries post-processing keeps (a8 / a11: 2 Q' C HW "
                                                           "flops, 4 (Q' C + C HW + Q' HW) B per frame)")
            if m2n:
                forms["attn_mask_pooled"] = entry(m2s, m2, m2n, "the decoder's per-layer attention masks: contraction on the level's pooled map + threshold "
                                                                "(2 Q C hw flops; fp32 operands in, one byte per (query, pixel) out)")
            lead = forms.get("mask_logits") or forms["attn_mask_pooled"]
            # the contraction is HBM-bound at fp32 (AI = 36 flop / B at Q = 100; exact-fp32 MFMA is the arithmetic)
            mask_roof = {"bound": "hbm", "kernel": "mask_gemm_kernel (exact-fp32 MFMA mask-logit contraction, csrc/mask_gemm.hip)",
                         "achieved": lead["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": lead["frac"], "traffic": None,
                         "mfma_f32_frac": lead["mfma_f32_frac"], "mfma_f32_peak_tflops": MFMA_F32_PEAK_TF, "forms": forms,
                         "note": "algorithmic bytes / summed HIP-event durations of the timed launches; mfma_f32_frac = the same launches' "
                                 "2 Q C HW flops against the dense fp32 matrix peak (both roofs shown: at Q' = 20 kept queries the "
                                 "stride-4 feature map read dominates, at Q = 100 the fp32 matrix pipe does)"}
        a_s, a_t, a_n = timer.attn.summary(lambda a: a["masked"] and a["long"])
        if a_n:
            tf = a_t["flops"] / a_s / 1e12
            s_s, s_t, s_n = timer.attn.summary(lambda a: not a["long"])
            attn_roof = {"bound": "mfma", "kernel": "attn_keysplit_kernel (masked cross-attention of the decoder, exact fp32 MFMA, keys split over "
                                                    "workgroups) + attn_combine_kernel", "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TF,
                         "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 4), "traffic": None, "launches_timed": a_n,
                         "us_per_launch": round(a_s / a_n * 1e6, 1), "ms_per_clip": round(a_s / timed_clips * 1e3, 3),
                         "alg_bytes_per_launch": int(a_t["bytes"] / a_n), "hbm_frac_of_alg_bytes": round(a_t["bytes"] / a_s / 1e9 / HBM_PEAK_GBS, 4),
                         "short_key_calls": {"launches_timed": s_n, "us_per_launch": round(s_s / max(1, s_n) * 1e6, 1),
                                             "note": "self-attention over the queries / tracker / refiner (Lk <= 128): latency-bound, not priced"},
                         "note": "flops = 4 Q HW_l C per frame and layer (QK^T + PV, SURVEY.md section 8d), summed over the timed masked "
                                 "launches (levels 920 / 3680 / 14720 keys) / their summed HIP-event durations (both launches of a call)"}
        bneck_roof = None
        b_s, b_t, b_n = timer.bneck.summary()
        if b_n:
            gbs, tf16 = b_t["bytes"] / b_s / 1e9, 3.0 * b_t["flops"] / b_s / 1e12
            bneck_roof = {"bound": "hbm", "kernel": "bneck_chain_kernel (csrc/bneck_x3.hip: a res2 bottleneck per launch - conv2 3x3 -> conv3 + "
                                                    "shortcut -> the next block's conv1, 64-channel maps as pre-split operand images)",
                          "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                          "launches_timed": b_n, "us_per_launch": round(b_s / b_n * 1e6, 1), "ms_per_clip": round(b_s / timed_clips * 1e3, 3),
                          "alg_bytes_per_launch": int(b_t["bytes"] / b_n), "mfma_f16_issued_tflops": round(tf16, 1),
                          "mfma_f16_frac": round(tf16 / MFMA_F16_PEAK_TF, 4),
                          "note": "algorithmic bytes (operand image in, shortcut in, block output out, next operand image out) / summed "
                                  "HIP-event durations of the timed launches; the three layers it replaces moved 7.2 GB per 30-frame block "
                                  "as separate launches, the chain 4.6 GB"}
        for roof, key in ((mask_roof, "mask_gemm_kernel"), (attn_roof, "attn_keysplit_kernel"), (bneck_roof, "bneck_chain_kernel")):
            t = x3_traffic.get(key)
            if roof is not None and t is not None:
                roof["traffic"] = t["hbm_bytes_per_launch"]      # (mean over every launch of the kernel family in the profiled bench run)
                roof["traffic_source"] = t["source"]
        ms = sorted(e0.elapsed_time(e1) for e0, e1 in lat)
        pct = lambda q: round(ms[min(len(ms) - 1, int(q * len(ms)))], 2) if ms else None
        res = {
            "metric": f"frames/sec DVIS++ {BB_NAME[args.backbone]} {args.mode}, 720p T={T} synthetic", "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE_NOTE if _Fn.X3 else "f32", "data": "synthetic",
            "config": {"workload": f"DVIS++ {args.mode} {BB_NAME[args.backbone]}, T={T} 720p synthetic clip (padded 736x1280), "
                                   f"{args.queries} queries, "
                                   f"temporal refiner {'on' if args.mode == 'offline' else 'off'}, task={args.task}, "
                                   f"frames sharded {world}-way",
                       "clips": f"{args.steps} distinct clips, seed 1234 + i",
                       "panoptic_candidates": [int(min(ncands)), int(max(ncands))] if ncands else None,
                       "segments": len(out.get("segments_infos", [])),
                       "tracker_spans": len(model.clip_shard.round_plan(T, getattr(model, "pipeline_rounds", 1))[0]),
                       "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                       "clip_stream": streamed, "warmup_clips_run": warm_clips,
                       "tracker_owner_rounds": False if world > 1 else None},
            "latency_fps": round(T / (pct(0.5) * 1e-3), 1) if ms else None,     # T / p50 clip latency (BASELINE.md section 3's definition)
            "latency_ms": {"p10": pct(0.1), "p50": pct(0.5), "p90": pct(0.9), "n": len(ms),
                           "note": "clip handed to the model -> its outputs complete (HIP events); under clip streaming a "
                                   "clip waits one segmenter pass of the previous clip"},
            "roofline": {"bound": "hbm", "kernel": "msda_fwd (fused MSDeformAttn forward)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "alg_bytes_per_launch": int(MSDA_BYTES_PER_FRAME_LAYER * nfr),
                         "us_per_launch": round(sec * 1e6, 1), "frames_per_launch": nfr, "launches_timed": nlaunch,
                         "alg_bytes_per_frame_layer": MSDA_BYTES_PER_FRAME_LAYER,
                         # the same launches against the roof the kernel is actually under: the vector-L1 return path
                         "l1_path": {"bound": "vector L1 -> VGPR return path (64 B / clk / CU)",
                                     "gathered_bytes_per_frame_layer": MSDA_GATHERED_BYTES_PER_FRAME_LAYER,
                                     "achieved": round(MSDA_GATHERED_BYTES_PER_FRAME_LAYER * nfr / sec / 1e9, 1),
                                     "peak": round(L1_PATH_PEAK_GBS, 1), "unit": "GB/s",
                                     "frac": round(MSDA_GATHERED_BYTES_PER_FRAME_LAYER * nfr / sec / 1e9 / L1_PATH_PEAK_GBS, 4)}},
        }
        if dist_info is not None:
            res["dist"] = dist_info
        if conv_roof is not None:
            res["roofline_conv3x3"] = conv_roof
        if convk_roof is not None:
            res["roofline_conv_x3"] = convk_roof
        if ffn_roof is not None:
            res["roofline_ffn"] = ffn_roof
        if bneck_roof is not None:
            res["roofline_bneck"] = bneck_roof
        if mask_roof is not None:
            res["roofline_mask_gemm"] = mask_roof
        if attn_roof is not None:
            res["roofline_attn"] = attn_roof
        if exact is not None:
            res["exact_f32"] = exact
        if cbc is not None:
            res["clip_by_clip"] = cbc
        if owner_line is not None:
            res["owner_rounds"] = owner_line
        if cand100 is not None:
            res["candidates_100"] = cand100
        if world == 1 and not dist_on and not args.no_extra:
            trk_roof = tracker_roofline(model)
            if trk_roof is not None:
                res["roofline_tracker"] = trk_roof
        if world == 1 and not dist_on and args.mode == "offline" and not args.no_extra:
            res["stages_ms"] = stage_breakdown(model, videos[0], args.task)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(model, clips[0], thr=videos[0].get("object_mask_threshold", 0.8))
        if world == 1 and not dist_on and not args.no_extra and not args.no_configs and args.mode == "offline" \
                and args.backbone == "r50" and T == 30 and len(clips) >= 2:
            res["configs"] = other_configurations(model, clips, device, args)
        print(json.dumps(res))
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
