"""Generate the config / checkpoint-layout fixtures of tests/test_d2_protocol.py (build container only).

    cd tests/golden && python gen_cfg_fixtures.py

1. cfg_*.json — the reference's own yaml files (configs/dvis_Plus/VIPSeg/*.yaml) with their ``_BASE_`` chains resolved:
   key/value DATA, exactly what ``cfg.merge_from_file`` would overlay on the defaults.
2. ref_state_shapes_dvis_plus_r50.json — ``{state_dict key: shape}`` of the REFERENCE modules (imported from
   /root/reference through _ref_import's third-party stubs) built at the sizes those yamls + the reference's
   add_*_config defaults select: the checkpoint layout a strict ``load_state_dict`` must accept.
"""
import json
import os
import sys

import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_import as R  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
CFG_DIR = os.path.join(R.REF, "configs", "dvis_Plus", "VIPSeg")
YAMLS = {
    "cfg_DVIS_Plus_Offline_R50": "DVIS_Plus_Offline_R50.yaml",
    "cfg_DVIS_Plus_Online_R50": "DVIS_Plus_Online_R50.yaml",
    "cfg_MinVIS_R50": "MinVIS_R50.yaml",
    "cfg_DVIS_Plus_Offline_VitAdapterL": "vit_adapter/DVIS_Plus_Offline_VitAdapterL.yaml",
}


def resolve(path):
    with open(path) as f:
        cfg = yaml.safe_load(f) or {}
    base = cfg.pop("_BASE_", None)
    if base is None:
        return cfg
    merged = resolve(os.path.join(os.path.dirname(path), base))

    def merge(a, b):
        for k, v in b.items():
            if isinstance(v, dict) and isinstance(a.get(k), dict):
                merge(a[k], v)
            else:
                a[k] = v
        return a
    return merge(merged, cfg)


def state_shapes():
    import torch  # noqa: F401
    pdm = R.ref("mask2former.modeling.pixel_decoder.msdeformattn")
    dm = R.ref("dvis_Plus.video_mask2former_transformer_decoder")
    tm, rm = R.ref("dvis_Plus.tracker"), R.ref("dvis_Plus.refiner")
    SS = sys.modules["detectron2.layers"].ShapeSpec
    inp = {k: SS(channels=c, stride=s) for k, c, s in (("res2", 256, 4), ("res3", 512, 8), ("res4", 1024, 16), ("res5", 2048, 32))}
    # sizes: configs/dvis_Plus/VIPSeg/*_R50.yaml + mask2former/config.py:6-123 + dvis_Plus/config.py:12-78 defaults
    pd = pdm.MSDeformAttnPixelDecoder(inp, transformer_dropout=0.0, transformer_nheads=8, transformer_dim_feedforward=1024,
                                      transformer_enc_layers=6, conv_dim=256, mask_dim=256, norm="GN",
                                      transformer_in_features=["res3", "res4", "res5"], common_stride=4)
    dec = dm.VideoMultiScaleMaskedTransformerDecoder_dvisPlus(
        256, True, num_classes=124, hidden_dim=256, num_queries=100, nheads=8, dim_feedforward=2048, dec_layers=9,
        pre_norm=False, mask_dim=256, enforce_input_project=False, num_frames=2, num_reid_head_layers=3, reid_hidden_dim=256)
    trk = tm.ReferringTracker_noiser(hidden_channel=512, feedforward_channel=2048, num_head=8, decoder_layer_num=6,
                                     noise_mode="wa", noise_ratio=0.8, mask_dim=256, class_num=124)
    rfn = rm.TemporalRefiner(hidden_channel=512, feedforward_channel=2048, num_head=8, decoder_layer_num=6, mask_dim=256,
                             class_num=124, windows=3)
    out = {}
    for prefix, mod in (("sem_seg_head.pixel_decoder.", pd), ("sem_seg_head.predictor.", dec), ("tracker.", trk),
                        ("refiner.", rfn)):
        for k, v in mod.state_dict().items():
            out[prefix + k] = list(v.shape)
    return out


if __name__ == "__main__":
    for name, rel in YAMLS.items():
        path = os.path.join(CFG_DIR, rel)
        if not os.path.exists(path):
            print("missing", path)
            continue
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(resolve(path), f, indent=1, sort_keys=True)
        print("wrote", name)
    with open(os.path.join(OUT, "ref_state_shapes_dvis_plus_r50.json"), "w") as f:
        json.dump(state_shapes(), f, indent=0, sort_keys=True)
    print("wrote ref_state_shapes_dvis_plus_r50")
