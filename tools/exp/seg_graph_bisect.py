"""Round 6: root-causing "a captured segmenter hipGraph returns wrong DECODER outputs after any tracker call" (round 5, commit
88fe0a7).  One process, many experiments, everything printed:

  A  capture segment() with every op front-end's output recorded (clones inside the capture -> graph-owned buffers), replay ==
     eager before the tracker runs;
  B  run the tracker (eager, no graph), replay again: first recorded op whose output differs, its max |d|, whether the outputs of the
     ops BEFORE it are intact;
  C  which PART of a tracker call breaks it: the cost GEMM, the host solver + its D2H / H2D copies, the recurrence, the heads —
     each alone on a freshly verified graph;
  D  which eager call HEALS a broken graph: synchronize, an eager encode, an eager decode, single op calls;
  E  the graph's node list (hipGraphDebugDotPrint through torch's debug_dump) — memcpy nodes with host pointers would show here.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd import functions as Fn  # noqa: E402
from dvis_plus_amd.graphs import GraphRunner  # noqa: E402
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
m = build_dvis_plus_r50("online", task="vps", object_mask_threshold=0.008).to(dev)
g = torch.Generator().manual_seed(11)
frames = torch.randint(0, 256, (5, 3, 360, 640), generator=g, dtype=torch.uint8).to(dev)

REC = None          # list receiving (name, tensor) while recording
NAMES = ["attention", "attn_mask_pooled", "attn_mask", "center_pool3", "add_layer_norm", "linear", "x3_linear", "gemm_nt", "mask_logits",
         "msda_fused_forward", "x3_ffn_ln", "x3_linear_ln", "maps_to_tokens", "tokens_to_map", "conv1x1", "upsample_add"]
ORIG = {n: getattr(Fn, n) for n in NAMES}


def _wrap(name):
    fn = ORIG[name]

    def w(*a, **k):
        out = fn(*a, **k)
        if REC is not None:
            outs = out if isinstance(out, (tuple, list)) else (out,)
            for j, o in enumerate(outs):
                if torch.is_tensor(o) and o.is_floating_point() or (torch.is_tensor(o) and o.dtype in (torch.uint8, torch.int32)):
                    REC.append((f"{len(REC):03d} {name}[{j}] {tuple(o.shape)}", o.clone()))
        return out
    return w


for n in NAMES:
    setattr(Fn, n, _wrap(n))


def segment_recorded(images):
    global REC
    REC = []
    out = tuple(m.segment(images))
    rec, REC = REC, None
    return out + tuple(t for _, t in rec), [n for n, _ in rec]


with torch.no_grad():
    images, _ = m.preprocess(frames)
    eager, names = segment_recorded(images)
    eager = [t.clone() for t in eager]
    eager2, _ = segment_recorded(images)
    print("eager vs eager (all recorded tensors equal):", all(torch.equal(a, b) for a, b in zip(eager, eager2)), "n recorded", len(names))
    names_box = {}

    def fn(im):
        out, nm = segment_recorded(im)
        names_box["n"] = nm
        return out
    runner = GraphRunner(fn, max_entries=16)
    NOUT = 4

    def replay(key="g"):
        out = runner((key,), images)
        torch.cuda.synchronize()
        return out

    def compare(tag, key="g", verbose=True):
        out = replay(key)
        bad = [(i, float((a.float() - b.float()).abs().max())) for i, (a, b) in enumerate(zip(eager, out)) if not torch.equal(a, b)]
        if not bad:
            print(f"   [{tag}] graph '{key}' == eager on all {len(out)} tensors")
            return True
        first = bad[0][0]
        nm = (["embds", "embds_nn", "logits", "mask_features"] + names)[first]
        print(f"   [{tag}] graph '{key}': {len(bad)} of {len(out)} tensors differ; FIRST = #{first} '{nm}' max|d| {bad[0][1]:.3e}; "
              f"final outputs differ: {[i for i, _ in bad if i < NOUT]}")
        if verbose:
            for i, d in bad[:6]:
                nm = (["embds", "embds_nn", "logits", "mask_features"] + names)[i]
                a, b = eager[i].float(), out[i].float()
                nz = int((a != b).sum())
                print(f"        #{i} {nm}: max|d| {d:.3e}, {nz} of {a.numel()} elements differ, graph has NaN: {bool(torch.isnan(b).any())}")
        return False

    # ---- E first: debug dump of a graph
    try:
        gdbg = torch.cuda.CUDAGraph()
        gdbg.enable_debug_mode()
        s_in = images.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m.segment(s_in)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(gdbg, capture_error_mode="thread_local"):
            dbg_out = m.segment(s_in)
        os.makedirs("gpurun_out", exist_ok=True)
        gdbg.debug_dump("gpurun_out/seg_graph.dot")
        txt = open("gpurun_out/seg_graph.dot").read()
        import re
        kinds = {}
        for mm in re.finditer(r'label="([^"]*)"', txt):
            lab = mm.group(1)
            k = "MEMCPY" if "emcpy" in lab or "MEMCPY" in lab else "MEMSET" if "emset" in lab or "MEMSET" in lab else "KERNEL/other"
            kinds[k] = kinds.get(k, 0) + 1
        print("E: graph dot dump:", len(txt), "bytes; node label kinds:", kinds)
        for mm in re.finditer(r'label="([^"]*emcpy[^"]*)"', txt):
            print("      memcpy node:", mm.group(1)[:300].replace("\n", " | "))
        for mm in list(re.finditer(r'label="([^"]*emset[^"]*)"', txt))[:3]:
            print("      memset node:", mm.group(1)[:300].replace("\n", " | "))
        os.system("head -c 3000 gpurun_out/seg_graph.dot > gpurun_out/seg_graph_head.dot; rm -f gpurun_out/seg_graph.dot")
    except Exception as e:  # noqa: BLE001
        print("E: debug dump failed:", repr(e)[:300])

    print("---- A: capture, replay before any tracker call")
    compare("A first replay")
    compare("A second replay")

    to_bctq = lambda z: z.permute(2, 0, 1).unsqueeze(0)
    e_embds, e_nn, e_logits, e_mf = eager[:4]
    trk = m.tracker

    def tracker_call(graphs):
        trk.use_graphs = graphs
        out = trk(to_bctq(e_embds), None, resume=False, frame_embeds_no_norm=to_bctq(e_nn), need_masks=False)
        trk.use_graphs = True
        torch.cuda.synchronize()
        return out

    print("---- C: pieces of a tracker call, each on a verified graph")
    from dvis_plus_amd import tracker as TR

    def piece_cost():
        fe = to_bctq(e_embds).permute(2, 3, 0, 1)
        with Fn.gemm_sizes_as(batch=fe.shape[0]):
            c = TR.cosine_costs(fe[:, :, 0, :], fe[0, :, 0, :])
        torch.cuda.synchronize()
        return c

    def piece_host(c):
        idx = TR.match_chains(torch.stack([c]))
        return torch.from_numpy(idx).to(dev)

    def piece_bigcpu():
        xs = [torch.randn(1 << 20) for _ in range(8)]        # churn the host allocator
        ys = [x.to(dev) for x in xs]
        torch.cuda.synchronize()
        return len(ys)

    def piece_kv():
        fe_nn = to_bctq(e_nn).permute(2, 3, 0, 1).contiguous()
        W, b = trk._kv_weights()
        with Fn.gemm_sizes_as(rows=fe_nn.shape[0] * fe_nn.shape[1]):
            return ORIG["linear"](fe_nn, W, b, own=True)

    def piece_attn64():
        q = torch.randn(100, 1, 512, device=dev)
        return ORIG["attention"](q, q, q, 8, short=True)

    def piece_attn48():
        q = torch.randn(100, 1, 3072, device=dev)
        return ORIG["attention"](q, q, q, 48, short=True)

    def piece_gemm_ln():
        a = torch.randn(100, 1, 512, device=dev)
        rp = trk.ref_proj.layers
        return Fn.gemm_ln(a, rp[0].weight, rp[0].bias, relu=True)

    def piece_gemm_ln_norm():
        a = torch.randn(100, 1, 512, device=dev)
        sa = trk.transformer_self_attention_layers[0]
        ff = trk.transformer_ffn_layers[0]
        cr = trk.transformer_cross_attention_layers[0]
        return Fn.gemm_ln(a, sa.self_attn.in_proj_weight, sa.self_attn.in_proj_bias, norm1=ff.norm, add=a, norm2=cr.norm)

    def piece_stacked():
        Wo, bo = trk._o_weights()
        att = torch.randn(100, 6 * 512, device=dev)
        return Fn.gemm_nt_stacked(att, Wo, bo)

    def piece_layernorm512():
        return trk.decoder_norm(torch.randn(5, 100, 1, 512, device=dev))

    def piece_add_ln512():
        x = torch.randn(100, 1, 512, device=dev)
        return ORIG["add_layer_norm"](x, x, trk.decoder_norm)

    def piece_gather():
        fe_nn = to_bctq(e_nn).permute(2, 3, 0, 1).contiguous()
        idx = torch.zeros(fe_nn.shape[:3], dtype=torch.long, device=dev)
        return torch.gather(fe_nn, 1, idx[..., None].expand(*fe_nn.shape))

    pieces = [("cosine cost GEMM (bmm_nt)", piece_cost), ("host allocator churn + H2D", piece_bigcpu), ("kv GEMM own=True", piece_kv),
              ("attention short d=64 8 heads", piece_attn64), ("attention short d=64 48 heads", piece_attn48), ("gemm_ln plain", piece_gemm_ln),
              ("gemm_ln norms", piece_gemm_ln_norm), ("gemm_nt_stacked", piece_stacked), ("nn.LayerNorm 512", piece_layernorm512),
              ("add_layer_norm 512", piece_add_ln512), ("torch.gather", piece_gather)]
    c_keep = None
    for tag, f in pieces:
        r = f()
        torch.cuda.synchronize()
        if tag.startswith("cosine"):
            c_keep = r
        ok = compare(f"after {tag}", verbose=False)
        if not ok:
            m.segment(images)
            torch.cuda.synchronize()
            compare("   ... and after an eager segment()", verbose=False)
    piece_host(c_keep)
    torch.cuda.synchronize()
    ok = compare("after host solver + D2H + H2D", verbose=False)
    if not ok:
        m.segment(images)
        compare("   ... and after an eager segment()", verbose=False)

    print("---- B: whole tracker call (eager, graphs off), then replay")
    tracker_call(False)
    okB = compare("B after eager tracker")
    if okB:
        tracker_call(True)
        okB = compare("B after tracker with its hipGraph (capture + replay)")
    if okB:
        tracker_call(True)
        okB = compare("B after tracker hipGraph replay only")
    if okB:
        proj = trk.project_mask_features(e_mf)
        torch.cuda.synchronize()
        okB = compare("B after project_mask_features")
    if okB:
        v = {"image": frames, "height": 360, "width": 640}
        m([v])
        torch.cuda.synchronize()
        okB = compare("B after a whole forward()")

    if not okB:
        print("---- D: what heals it?")
        torch.cuda.synchronize()
        compare("D after synchronize only", verbose=False)
        torch.cuda.empty_cache()
        compare("D after empty_cache", verbose=False)
        # single eager op calls at the decoder's shapes
        pred = m.sem_seg_head.predictor
        q = torch.randn(100, 5, 256, device=dev)
        ORIG["attention"](q, q, q, 8)
        torch.cuda.synchronize()
        if compare("D after eager attention short d=32", verbose=False):
            print("   HEALED by: attention short d=32")
        else:
            ORIG["linear"](q, pred.class_embed.weight, pred.class_embed.bias)
            torch.cuda.synchronize()
            if compare("D after eager own gemm (class_embed)", verbose=False):
                print("   HEALED by: own gemm")
            else:
                ORIG["add_layer_norm"](q, q, pred.decoder_norm)
                torch.cuda.synchronize()
                if compare("D after eager add_layer_norm 256", verbose=False):
                    print("   HEALED by: add_layer_norm")
                else:
                    pred.decoder_norm(q)
                    torch.cuda.synchronize()
                    if compare("D after eager nn.LayerNorm 256", verbose=False):
                        print("   HEALED by: nn.LayerNorm")
                    else:
                        ms, mf = m.encode(images)
                        torch.cuda.synchronize()
                        if compare("D after eager encode()", verbose=False):
                            print("   HEALED by: encode")
                        else:
                            m.decode(ms, mf)
                            torch.cuda.synchronize()
                            if compare("D after eager decode()", verbose=False):
                                print("   HEALED by: decode")
                            else:
                                print("   nothing healed it")
print("done")
