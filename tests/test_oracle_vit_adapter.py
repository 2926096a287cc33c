"""CPU: the ViT-Adapter oracle (oracle/vit_adapter_torch.py) against the golden captured from the imported reference
(tests/golden/g8_vit_adapter.npz: DinoVisionTransformer + DinoV2ViTAdapter, tiny width, head dim 32)."""
import torch

from conftest import Golden
from oracle import vit_adapter_torch as V

TOL = dict(rtol=1e-5, atol=5e-6)


def test_vit_adapter_oracle_matches_reference_outputs():
    g = Golden("g8_vit_adapter")
    cfg = g.meta["cfg"]
    stages = {}
    with torch.no_grad():
        f = V.vit_adapter_forward(g.sd, g.ins["x"], heads=cfg["heads"], deform_heads=cfg["deform_heads"],
                                  interaction_indexes=cfg["interaction_indexes"], n_points=cfg["n_points"],
                                  stages=stages)
    torch.testing.assert_close(stages["tokens"], g.outs["tokens"], **TOL)
    torch.testing.assert_close(stages["block0"], g.outs["block0"], **TOL)
    for got, k in zip(f, ("f1", "f2", "f3", "f4")):
        torch.testing.assert_close(got, g.outs[k], **TOL)
