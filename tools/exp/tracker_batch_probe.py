"""Which op of the tracker recurrence gives a clip different bits when two clips advance together?  (dev probe)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd import functions as Fn
from dvis_plus_amd.tracker import ReferringTracker_noiser
dev = "cuda:0"
torch.manual_seed(0)
Q, C, T = 100, 512, 4
x1 = torch.randn(Q, 1, C, device=dev); x2 = torch.randn(Q, 1, C, device=dev)
xb = torch.cat([x1, x2], 1).contiguous()
W = torch.randn(2048, C, device=dev) * 0.05; b = torch.randn(2048, device=dev)
with torch.no_grad(), Fn.gemm_sizes_as(rows=Q):
    a = Fn.linear(x1, W, b, own=True); bb = Fn.linear(xb, W, b, own=True)
    print("gemm   ", torch.equal(a[:, 0], bb[:, 0]), float((a[:, 0] - bb[:, 0]).abs().max()))
    q1 = torch.randn(Q, 1, C, device=dev); qb = torch.cat([q1, torch.randn(Q, 1, C, device=dev)], 1).contiguous()
    k1 = torch.randn(Q, 1, C, device=dev); kb = torch.cat([k1, torch.randn(Q, 1, C, device=dev)], 1).contiguous()
    o1 = Fn.attention(q1, k1, k1, 8); ob = Fn.attention(qb, kb, kb, 8)
    print("attn   ", torch.equal(o1[:, 0], ob[:, 0]), float((o1[:, 0] - ob[:, 0]).abs().max()))
    ln = torch.nn.LayerNorm(C).to(dev)
    l1 = Fn.add_layer_norm(x1.contiguous(), q1, ln); lb = Fn.add_layer_norm(xb, qb, ln)
    print("add_ln ", torch.equal(l1[:, 0], lb[:, 0]))
    print("torchLN", torch.equal(ln(x1)[:, 0], ln(xb)[:, 0]))
    trk = ReferringTracker_noiser(hidden_channel=C, feedforward_channel=2048, num_head=8, decoder_layer_num=2, mask_dim=256, class_num=10).to(dev).eval()
    fe = torch.randn(2, C, T, Q, device=dev)
    for g in (False, True):
        trk.use_graphs = g
        o_b = trk(fe, None, frame_embeds_no_norm=fe * 1.3, need_masks=False)
        o_0 = trk(fe[:1], None, frame_embeds_no_norm=fe[:1] * 1.3, need_masks=False)
        o_1 = trk(fe[1:], None, frame_embeds_no_norm=fe[1:] * 1.3, need_masks=False)
        for key in ("pred_embds", "pred_logits", "pred_references"):
            print(f"tracker graphs={g} {key}: clip0 {torch.equal(o_b[key][:1], o_0[key])} clip1 {torch.equal(o_b[key][1:], o_1[key])} "
                  f"maxdiff {float((o_b[key][:1] - o_0[key]).abs().max()):.2e}")
