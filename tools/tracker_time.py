"""Time the referring tracker's recurrence alone (hipGraph replay) at the production sizes, fused chain vs layer by layer.
    python tools/tracker_time.py [T] [B]
Prints ms per clip, us per frame, and the 101 MB / frame weight-streaming fraction of SURVEY.md section 8(d)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvis_plus_amd.tracker import ReferringTracker_noiser  # noqa: E402

DEV = "cuda:0"


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    torch.manual_seed(0)
    trk = ReferringTracker_noiser(hidden_channel=512, feedforward_channel=2048, num_head=8, decoder_layer_num=6,
                                  mask_dim=256, class_num=124).eval().to(DEV)
    g = torch.Generator().manual_seed(1)
    fe_nn = torch.randn(B, 512, T, 100, generator=g).to(DEV)
    fe = torch.nn.functional.layer_norm(fe_nn.permute(0, 2, 3, 1), (512,)).permute(0, 3, 1, 2).contiguous()
    weights_mb = sum(p.numel() for n, p in trk.named_parameters()
                     if not n.startswith(("mask_embed", "mask_feature_proj", "class_embed"))) * 4 / 1e6
    for fused in ((True,) if os.environ.get("DVIS_TT_ONLY_FUSED") == "1" else (True, False)):
        trk.fused_chain = fused
        with torch.no_grad():
            for _ in range(3):
                trk(fe, None, resume=False, frame_embeds_no_norm=fe_nn, need_masks=False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                trk(fe, None, resume=False, frame_embeds_no_norm=fe_nn, need_masks=False)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        # the recurrence alone: replay of the captured graph (no host assignment, no heads)
        ent = next(iter(trk._graph._cache.values())) if trk._graph._cache else None
        if ent is not None:
            torch.cuda.synchronize()
            e0.record()
            for _ in range(n):
                ent[0].replay()
            e1.record()
            torch.cuda.synchronize()
            gms = e0.elapsed_time(e1) / n
            print(f"   graph replay alone: {gms:.3f} ms per pass, {gms * 1e3 / T:.1f} us per frame -> "
                  f"{weights_mb * 1e6 * T / (gms * 1e-3) / 8e12:.4f} of 8 TB/s")
        trk._graph.clear()
        print(f"fused={fused}  T={T} B={B}: {ms:.3f} ms per pass, {ms * 1e3 / T:.1f} us per frame; weight streaming "
              f"{weights_mb:.1f} MB/frame -> {weights_mb * 1e6 * T / (ms * 1e-3) / 1e9:.1f} GB/s = "
              f"{weights_mb * 1e6 * T / (ms * 1e-3) / 8e12:.4f} of 8 TB/s", flush=True)


if __name__ == "__main__":
    main()
