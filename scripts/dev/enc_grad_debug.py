import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image_structured, synth_state_dict
from focoos_amd.train_nn import HybridEncoder, ResNetVd
from focoos_amd import _lib as L
from oracle import detr_oracle as O
from tests.helpers import rel_l2
DEV = "cuda:0"
cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
sd = synth_state_dict(cfg, 12)
pre = "pixel_decoder.backbone."
net = ResNetVd(50).to(DEV)
net.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
enc = HybridEncoder(L.load()).to(DEV)
enc.load_state_dict({k[len("pixel_decoder."):]: v for k, v in sd.items() if k.startswith("pixel_decoder.") and not k.startswith(pre)}, strict=True)
HH, WW = int(sys.argv[1]), int(sys.argv[2])
imgs = [synth_image_structured(60 + i, HH, WW) for i in range(2)]
x_u8 = torch.from_numpy(np.stack(imgs)).to(DEV)
g = torch.Generator().manual_seed(4)
proj = [torch.randn(256, generator=g) for _ in range(3)]
f = net(x_u8)
for k in f: f[k].retain_grad()
# expose encoder internals by re-implementing forward with retain_grad
lib = enc.lib
pj = [p(ff) for p, ff in zip(enc.input_proj, [f["res3"], f["res4"], f["res5"]])]
for t in pj: t.retain_grad()
outs = enc([f["res3"], f["res4"], f["res5"]])
loss = sum((o.float() * p.to(DEV)).sum() for o, p in zip(outs, proj)) * 1e-2
loss.backward()
ref_sd = {k: v for k, v in sd.items()}
mean = torch.tensor(cfg["pixel_mean"]).view(-1, 1, 1); std = torch.tensor(cfg["pixel_std"]).view(-1, 1, 1)
xi = (O.get_torch_batch(imgs, None) - mean) / std
xi.requires_grad_(True)
feats = O.resnet_vd(ref_sd, pre[:-1], xi, O.RESNET_BLOCKS[50])
fr = feats
for v in fr.values(): v.retain_grad()
col = {}
ref_outs = O.hybrid_encoder(ref_sd, [fr["res3"], fr["res4"], fr["res5"]], cfg, col)
(sum((o * p.view(1, -1, 1, 1)).sum() for o, p in zip(ref_outs, proj)) * 1e-2).backward()
for k in ("res3", "res4", "res5"):
    print(k, "grad rel-L2", rel_l2(f[k].grad.float().cpu().permute(0, 3, 1, 2), fr[k].grad), "ref |grad| mean", float(fr[k].grad.abs().mean()))
