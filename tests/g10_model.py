"""Test infrastructure: the PRODUCT meta-architecture at the sizes of the g10_window_loop golden (reference sub-modules'
state_dict loaded strictly), for the CPU host-logic test and the GPU parity test of the a12 composition."""
import torch

from conftest import Golden
from toy_backbone import ToyBackbone


def build(mode, task, device="cpu"):
    from dvis_plus_amd.meta_architecture import DVIS_Plus_offline, DVIS_Plus_online, MaskFormerHead
    from dvis_plus_amd.pixel_decoder import MSDeformAttnPixelDecoder
    from dvis_plus_amd.refiner import TemporalRefiner
    from dvis_plus_amd.tracker import ReferringTracker_noiser
    from dvis_plus_amd.transformer_decoder import VideoMultiScaleMaskedTransformerDecoder_dvisPlus
    g = Golden("g10_window_loop")
    cfg, sd = g.meta["cfg"], dict(g.sd)
    K, Q, HID, MD = cfg["K"], cfg["Q"], cfg["hidden"], cfg["mask_dim"]
    bb = ToyBackbone()
    pd = MSDeformAttnPixelDecoder(bb.output_shape(), transformer_dropout=0.0, transformer_nheads=cfg["nheads"],
                                  transformer_dim_feedforward=cfg["enc_ffn"], transformer_enc_layers=cfg["enc_layers"],
                                  conv_dim=HID, mask_dim=MD, norm="GN", transformer_in_features=["res3", "res4", "res5"],
                                  common_stride=4)
    pred = VideoMultiScaleMaskedTransformerDecoder_dvisPlus(
        HID, True, num_classes=K, hidden_dim=HID, num_queries=Q, nheads=cfg["nheads"], dim_feedforward=cfg["dec_ffn"],
        dec_layers=cfg["dec_layers"], pre_norm=False, mask_dim=MD, enforce_input_project=False, num_frames=cfg["window"],
        num_reid_head_layers=3, reid_hidden_dim=HID)
    head = MaskFormerHead(num_classes=K, pixel_decoder=pd, transformer_predictor=pred)
    trk = ReferringTracker_noiser(hidden_channel=2 * HID, feedforward_channel=cfg["trk_ffn"], num_head=cfg["trk_heads"],
                                  decoder_layer_num=cfg["tracker_layers"], noise_mode="wa", mask_dim=MD, class_num=K)
    kw = dict(backbone=bb, sem_seg_head=head, num_queries=Q, object_mask_threshold=cfg["object_mask_threshold"],
              overlap_threshold=cfg["overlap_threshold"], n_things=cfg["n_things"], tracker=trk, task=task,
              max_num=cfg["max_num"], window_size=cfg["window"], window_inference=True)
    if mode == "offline":
        ref = TemporalRefiner(hidden_channel=2 * HID, feedforward_channel=cfg["trk_ffn"], num_head=cfg["trk_heads"],
                              decoder_layer_num=cfg["refiner_layers"], mask_dim=MD, class_num=K, windows=cfg["window"])
        m = DVIS_Plus_offline(refiner=ref, **kw)
    else:
        m = DVIS_Plus_online(**kw)
        sd = {k: v for k, v in sd.items() if not k.startswith("refiner.")}
    mean, std = sd.pop("pixel_mean"), sd.pop("pixel_std")
    m.load_state_dict(sd, strict=True)                      # the checkpoint surface: the reference's key names
    assert torch.equal(m.pixel_mean, mean) and torch.equal(m.pixel_std, std)
    frames = g.ins["frames"]
    return m.eval().to(device), g, cfg, frames


def video(frames, cfg, lo=0, hi=None, keep=None, device="cpu"):
    v = {"image": [f.to(device) for f in frames[lo:hi]], "height": cfg["out_hw"][0], "width": cfg["out_hw"][1]}
    if keep is not None:
        v["keep"] = keep
    return v
