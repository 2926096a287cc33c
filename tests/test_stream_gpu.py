"""GPU: DVIS_Plus_offline.stream() is structurally hang-free and its phase B is deterministic.

Rounds 1-2 overlapped phase B (all-gather, tracker, refiner, masks, post-processing) of clip i with phase A of clip
i + 1 on two HIP streams and met a deadlock: two hipBLASLt stream-K GEMMs in flight on two streams can spin on each other
forever.  Since round 3 phase B issues no library GEMM / convolution at all — every projection of
ReferringTracker_noiser (dvis_Plus/tracker.py:187-357) and TemporalRefiner (dvis_Plus/refiner.py:91-158) runs on
dvis_gemm_nt, which never waits for another workgroup — so the caps (T <= 32 on the second stream, 4 GiB per segmenter
call) are gone.  Asserted here:
  * no aten matmul / convolution is dispatched while phase B runs (a library GEMM cannot sneak back in);
  * phase B is bit-reproducible: same per-frame queries in -> torch.equal refined embeddings, logits, panoptic maps,
    hipGraph replay and eager launch alike;
  * a soak of 200 (DVIS_SOAK_CLIPS) streamed T = 64 clips (one 64-frame segmenter call each: 4.7 GiB FFN activation, the shape that
    stalled about once in ten runs before) finishes inside a hard timeout, in a child process.
"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LIBRARY_OPS = ("mm", "addmm", "bmm", "baddbmm", "matmul", "linear", "_addmm_activation", "convolution",
               "_convolution", "conv1d", "conv2d", "miopen_convolution", "cudnn_convolution", "addmv", "mv", "dot",
               "einsum", "tensordot")


def _model(task="vps", **kw):
    from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
    return build_dvis_plus_r50("offline", task=task, object_mask_threshold=0.0, **kw)


def _clip(T, seed, h=360, w=640):
    g = torch.Generator().manual_seed(seed)
    return {"image": torch.randint(0, 256, (T, 3, h, w), generator=g, dtype=torch.uint8).to(DEV), "height": h, "width": w}


@pytest.mark.parametrize("task,kw", [("vps", {}), ("vis", {}), ("vss", {}), ("vss", {"num_classes": 150})])
def test_phase_b_dispatches_no_library_gemm_or_convolution(task, kw):
    """vss with 150 classes: past the one-pass kernel's 128-class limit — the class contraction must still be the own GEMM."""
    from torch.utils._python_dispatch import TorchDispatchMode
    seen, offenders = [], []

    class Watch(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = func.overloadpacket.__name__
            seen.append(name)
            if name in LIBRARY_OPS:
                offenders.append(name)
            return func(*args, **(kwargs or {}))

    m = _model(task=task, **kw).to(DEV)
    m.object_mask_threshold = 0.008                 # some queries reach the panoptic stage
    inner = m._track_round

    def watched(sts):
        with Watch():
            return inner(sts)
    m._track_round = watched
    for use_graphs in (False, True):                # eager launches, then hipGraph capture (its warm-ups run the same code)
        m.tracker.use_graphs = m.refiner.use_graphs = use_graphs
        seen.clear()
        outs = list(m.stream([_clip(4, 1), _clip(5, 2)]))
        torch.cuda.synchronize()
        assert len(outs) == 2 and outs[1]["pred_masks"].shape[1 if task == "vis" else 0] == 5
        assert seen, "the dispatch watch saw no op: phase B did not run under it"
        assert not offenders, f"phase B dispatched library GEMM / convolution ops: {sorted(set(offenders))}"


def test_phase_b_is_bit_reproducible():
    """Same per-frame queries in -> the same bits out, run after run, graph replay vs eager launch, forward() vs
    stream()'s side stream."""
    m = _model().to(DEV)
    m.object_mask_threshold = 0.008
    video = _clip(7, 3)
    with torch.no_grad():
        st = m._segment_phase(video)

        def phase_b():
            m.debug_stages = {}
            out = m._track_phase(dict(st))
            emb = m.debug_stages["mask_fn"](None).clone()
            return (out["pred_masks"].clone(), out["segments_infos"], m.debug_stages["cls"].clone(),
                    m.debug_stages["aux"].clone(), emb)
        ref = phase_b()
        assert ref[0].any(), "degenerate test: empty panoptic map"
        for use_graphs in (True, True, False, False):
            m.tracker.use_graphs = m.refiner.use_graphs = use_graphs
            got = phase_b()
            assert got[1] == ref[1]
            for a, b, what in zip(got, ref, ("panoptic map", None, "refiner class logits", "tracker class logits",
                                             "mask logits of all queries")):
                if what is not None:
                    assert torch.equal(a, b), f"phase B not reproducible (graphs={use_graphs}): {what} differ"
        # the side stream of stream() next to another clip's segmenter: same bits again
        m.tracker.use_graphs = m.refiner.use_graphs = True
        seg = m._segment_round

        def replay_first(videos, shift=0, **kw):
            sts = seg(videos, shift, **kw)
            if videos and videos[0] is video:
                for k in ("embds", "embds_nn", "logits", "mf"):
                    sts[0][k] = st[k]                      # the SAME phase-A tensors as above
            return sts
        m._segment_round = replay_first
        outs = list(m.stream([video, _clip(7, 4), _clip(7, 5)]))
        torch.cuda.synchronize()
        assert torch.equal(outs[0]["pred_masks"], ref[0]) and outs[0]["segments_infos"] == ref[1]
        # ... and with phase B on its own host thread (opt-in schedule): the same bits, clips in order, errors re-raised
        m.stream_thread = True
        # (only clip 0 has replayed phase-A tensors: at this small size the library kernels of phase A are not reproducible
        # from call to call, so the other clips are checked for ORDER — their lengths differ — not for bits)
        clips = [video, _clip(6, 4), _clip(5, 5), _clip(4, 6), _clip(3, 7)]
        outs_t = list(m.stream(clips))
        torch.cuda.synchronize()
        assert len(outs_t) == 5 and torch.equal(outs_t[0]["pred_masks"], ref[0]) and outs_t[0]["segments_infos"] == ref[1]
        assert [o["pred_masks"].shape[0] for o in outs_t] == [7, 6, 5, 4, 3]

        def boom(sts):
            raise RuntimeError("phase B failed")
        good, m._track_round = m._track_round, boom
        with pytest.raises(RuntimeError, match="phase B failed"):
            list(m.stream(clips))
        m._track_round = good
        assert len(list(m.stream(clips[:2]))) == 2      # the model is usable afterwards


SOAK = r"""
import sys, time, torch
sys.path.insert(0, %r)
import bench
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50, segmenter_frames_per_call
assert segmenter_frames_per_call(64, 736, 1280) == 64
dev = torch.device("cuda:0")
m = build_dvis_plus_r50("offline", task="vps", object_mask_threshold=0.0).to(dev)
clips = [{"image": bench.synthetic_clip(64, dev, seed=90 + i), "height": 720, "width": 1280} for i in range(3)]
m.object_mask_threshold = bench.calibrate_threshold(m, clips[:1], 20)
n, t0 = int(sys.argv[1]), time.time()
done = 0
for out in m.stream(clips[i %% 3] for i in range(n)):
    done += 1
    assert out["pred_masks"].shape == (64, 720, 1280)
    if done %% 50 == 0:
        torch.cuda.synchronize()
        print(f"soak: {done} clips, {64 * done / (time.time() - t0):.1f} frames/s", flush=True)
torch.cuda.synchronize()
print(f"SOAK OK {done} clips of 64 frames in {time.time() - t0:.1f} s = {64 * done / (time.time() - t0):.1f} frames/s", flush=True)
"""


def test_soak_streamed_t64_clips_finish():
    n = int(os.environ.get("DVIS_SOAK_CLIPS", "200"))      # (round 5 cut it to 120 for the suite's time limit; round 6 got the time back from the oracle's host threads)
    r = subprocess.run([sys.executable, "-c", SOAK % ROOT, str(n)], cwd=ROOT, capture_output=True, text=True,
                       timeout=int(os.environ.get("DVIS_SOAK_TIMEOUT", "420")))
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-12:])
    assert r.returncode == 0 and f"SOAK OK {n} clips" in r.stdout, tail
    print(tail)


def test_tracker_batch_is_bit_identical_on_the_gpu():
    """tracker_batch = 2: two clips' tracker recurrences advance together in one pass (batch 2 through every GEMM, attention
    and add+LayerNorm of the recurrence, hipGraph-captured).  The GEMM tile configuration is pinned to one clip's rows and
    the attention kernel works per (batch, head), so — from the same per-frame queries — every clip's outputs equal the
    one-by-one run BIT FOR BIT.  (Phase A is replayed from stored tensors: at this small test size the library picks
    GEMM kernels that are not reproducible run to run; at 720p / T = 30 they are, tools/determinism_probe.py.)"""
    m = _model().to(DEV)
    m.object_mask_threshold = 0.008
    clips = [_clip(7, 20 + i) for i in range(5)] + [_clip(6, 30)]
    with torch.no_grad():
        stored = {id(c): m._segment_phase(c) for c in clips}
        want = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in m._track_phase(dict(stored[id(c)])).items()}
                for c in clips]
        m._segment_round = lambda videos, shift=0, **kw: [dict(stored[id(v)]) for v in videos]
        calls = []
        fwd = m.tracker.forward
        m.tracker.forward = lambda fe, *a, **k: (calls.append(fe.shape[0]), fwd(fe, *a, **k))[1]
        m.tracker_batch = 2
        got = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()} for o in m.stream(clips)]
    torch.cuda.synchronize()
    assert calls == [2, 2, 1, 1]                        # (7, 7) (7, 7) together; (7, 6) differ in length: one by one
    for i, (g, w) in enumerate(zip(got, want)):
        assert g["segments_infos"] == w["segments_infos"] and g["pred_ids"] == w["pred_ids"], i
        assert torch.equal(g["pred_masks"], w["pred_masks"]), f"clip {i}: panoptic map differs from the unbatched run"
    assert any(w["segments_infos"] for w in want), "degenerate test: no segment anywhere"



def test_online_segmenter_graph_replay_equals_eager(monkeypatch):
    """DVIS_Plus_online replays the segmenter of a small window from a hipGraph (config #2: ~450 launches per 5-frame window,
    17 % of the wall was the device waiting for the host).  Same bits as the eager launches, window after window (`keep`
    continues the video: meta_architecture.py:629-632, 793), through tracker calls in between and replay after replay — the
    regression test of round 5's withdrawn graph (a hipMemsetAsync node that fills with garbage from its second replay on, csrc/dvis_common.h)."""
    from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
    m = build_dvis_plus_r50("online", task="vps", object_mask_threshold=0.008).to(DEV)
    windows = [_clip(5, 11), _clip(5, 12), _clip(3, 13)]

    def run():
        outs = []
        for i, w in enumerate(windows):
            o = m([dict(w, keep=i > 0)])
            outs.append((o["pred_masks"].clone(), o["segments_infos"], o["pred_ids"]))
        return outs
    monkeypatch.setenv("DVIS_SEGMENTER_GRAPH", "0")
    eager = run()
    assert m._seg_graph is None or not m._seg_graph._cache
    monkeypatch.setenv("DVIS_SEGMENTER_GRAPH", "8")
    first, replay, replay2 = run(), run(), run()                 # capture, then pure replays (tracker calls in between)
    assert len(m._seg_graph._cache) == 2          # two window shapes (5 and 3 frames)
    assert any(e[1] for e in eager), "degenerate test: no segment"
    for a, b, c, d in zip(eager, first, replay, replay2):
        assert torch.equal(a[0], b[0]) and torch.equal(a[0], c[0]) and torch.equal(a[0], d[0])
        assert a[1] == b[1] == c[1] == d[1] and a[2] == b[2] == c[2] == d[2]


def test_captured_attention_mask_counts_replay(monkeypatch):
    """The op-level form of the same regression: attn_mask / attn_mask_pooled inside a hipGraph give the eager call's mask AND
    counts on every replay (the counts are accumulated with atomics onto a buffer the launch itself zeroes — with a kernel)."""
    from dvis_plus_amd import functions as Fn
    from dvis_plus_amd.graphs import GraphRunner
    torch.manual_seed(0)
    emb = torch.randn(3, 100, 256, device=DEV)
    feat = torch.randn(3, 256, 48, 80, device=DEV)
    with torch.no_grad():
        pooled = Fn.center_pool3(feat)
        want_a, want_p = Fn.attn_mask(emb, feat, (12, 20)), Fn.attn_mask_pooled(emb, pooled[1])
        g = GraphRunner(lambda e, f, p: (*Fn.attn_mask(e, f, (12, 20)), *Fn.attn_mask_pooled(e, p)))
        for i in range(4):
            got = [t.clone() for t in g("k", emb, feat, pooled[1])]
            torch.empty(1 << 20, device=DEV).normal_()              # (other work between the replays)
            assert torch.equal(got[0], want_a[0]) and torch.equal(got[1], want_a[1]), f"attn_mask, replay {i}"
            assert torch.equal(got[2], want_p[0]) and torch.equal(got[3], want_p[1]), f"attn_mask_pooled, replay {i}"
    assert int(want_a[1].min()) >= 0 and int(want_p[1].max()) <= 12 * 20
