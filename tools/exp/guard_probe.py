"""Does an R50 forward at the small test sizes leave the range-guard word set?  (development aid)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd import functions as Fn
from dvis_plus_amd.backbone import build_resnet50
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = build_resnet50().to(dev).eval()
with torch.no_grad():
    for hw in ((96, 160), (64, 96), (128, 256), (736, 1280)):
        x = torch.rand(2, 3, *hw, device=dev) * 255 - 120
        for name in ("stem", "res2", "res3", "res4", "res5"):
            pass
        out = m(x)
        mx = {k: (float(v.abs().max()), bool(torch.isfinite(v).all())) for k, v in out.items()}
        try:
            Fn.X3_GUARD.check_now(dev, m)
            print(hw, "guard clean", mx)
        except Fn.X3RangeError as e:
            print(hw, "GUARD:", str(e)[:160], mx)
        # block by block
        y = m.stem(x.float())
        for sname in m.stage_names:
            for bi, blk in enumerate(getattr(m, sname)):
                y = blk(y)
                try:
                    Fn.X3_GUARD.check_now(dev, m)
                except Fn.X3RangeError as e:
                    print("   ", sname, bi, "max|x|", float(y.abs().max()), str(e)[28:120])
