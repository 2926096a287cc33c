"""CPU: yaml + _BASE_ config loading and registry-driven model construction with the reference's key names."""
import torch

from dvis_plus_amd.config import CfgNode, build_model, get_default_cfg


def test_base_inheritance_overrides_and_build(tmp_path):
    (tmp_path / "Base.yaml").write_text(
        "MODEL:\n  SEM_SEG_HEAD:\n    NUM_CLASSES: 11\n    TRANSFORMER_ENC_LAYERS: 1\n"
        "  MASK_FORMER:\n    DEC_LAYERS: 3\n    NUM_OBJECT_QUERIES: 9\n    TEST:\n      TASK: vps\n")
    (tmp_path / "Online.yaml").write_text(
        "_BASE_: Base.yaml\nMODEL:\n  META_ARCHITECTURE: DVIS_Plus_online\n  TRACKER:\n    DECODER_LAYERS: 2\n")
    (tmp_path / "Offline.yaml").write_text(
        "_BASE_: Online.yaml\nMODEL:\n  META_ARCHITECTURE: DVIS_Plus_offline\n  REFINER:\n    DECODER_LAYERS: 1\n")
    cfg = get_default_cfg()
    cfg.merge_from_file(str(tmp_path / "Offline.yaml"))
    cfg.merge_from_list(["MODEL.MASK_FORMER.TEST.MAX_NUM", "7", "MODEL.MASK_FORMER.DIM_FEEDFORWARD", "128"])
    assert cfg.MODEL.META_ARCHITECTURE == "DVIS_Plus_offline" and cfg.MODEL.TRACKER.DECODER_LAYERS == 2
    assert cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES == 11 and cfg.MODEL.MASK_FORMER.HIDDEN_DIM == 256   # default kept
    m = build_model(cfg, n_things=4)
    assert type(m).__name__ == "DVIS_Plus_offline" and m.task == "vps" and m.max_num == 7
    assert len(m.tracker.transformer_ffn_layers) == 2 and len(m.refiner.transformer_ffn_layers) == 1
    assert m.sem_seg_head.predictor.num_layers == 2 and m.sem_seg_head.predictor.num_queries == 9
    assert m.tracker.decoder_norm.weight.shape[0] == 512                      # REID branch doubles the width
    keys = m.state_dict().keys()
    assert "sem_seg_head.pixel_decoder.transformer.encoder.layers.0.self_attn.sampling_offsets.weight" in keys
    assert "refiner.conv_short_aggregate_layers.0.2.weight" in keys and "backbone.res2.0.conv1.norm.running_var" in keys
    cfg.MODEL.META_ARCHITECTURE = "DVIS_Plus_online"
    assert type(build_model(cfg)).__name__ == "DVIS_Plus_online"
    assert isinstance(cfg.MODEL, CfgNode) and isinstance(torch.tensor(cfg.MODEL.PIXEL_MEAN), torch.Tensor)
