import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_state_dict, synth_image_structured
from focoos_amd.train_nn import HybridEncoder, ResNetVd, _AddFn
from focoos_amd import _lib as L
from oracle import detr_oracle as O
from tests.helpers import rel_l2
DEV = "cuda:0"
cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
sd = synth_state_dict(cfg, 12)
pre = "pixel_decoder.backbone."
enc = HybridEncoder(L.load()).to(DEV)
enc.load_state_dict({k[len("pixel_decoder."):]: v for k, v in sd.items() if k.startswith("pixel_decoder.") and not k.startswith(pre)}, strict=True)
net = ResNetVd(50).to(DEV)
net.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
imgs = [synth_image_structured(60 + i, 128, 160) for i in range(2)]
with torch.no_grad():
    f5 = net(torch.from_numpy(np.stack(imgs)).to(DEV))["res5"]
    src0 = enc.input_proj[2](f5).reshape(2, 20, 256).cpu()
print("src std", float(src0.float().std()), "absmax", float(src0.float().abs().max()))
g = torch.Generator().manual_seed(1)
dy = torch.randn(2, 20, 256, generator=g).bfloat16()
lay = enc.encoder[0].layers[0]
lib = enc.lib
T, R = {}, {}
def keep(d, n, t): t.retain_grad(); d[n] = t; return t
x = keep(T, "src", src0.to(DEV).requires_grad_(True))
posd = enc._pos_for(4, 5, x.device)
qk = keep(T, "qk", _AddFn.apply(x, posd, lib))
sa = keep(T, "sa", lay.self_attn(qk, qk, x, residual=x))
n1 = keep(T, "n1", lay.norm1(sa))
l1 = keep(T, "l1", lay.linear1(n1))
l2 = keep(T, "l2", lay.linear2(l1, residual=n1))
n2 = keep(T, "n2", lay.norm2(l2))
n2.backward(dy.to(DEV))
P = "pixel_decoder.encoder.0.layers.0"
xr = keep(R, "src", src0.float().requires_grad_(True))
pos = O.position_embedding_sine(4, 5, 128)
qkr = keep(R, "qk", xr + pos)
a = O.mha(sd, f"{P}.self_attn", qkr, qkr, xr, 8)
sar = keep(R, "sa", xr + a)
n1r = keep(R, "n1", O.layer_norm(sd, f"{P}.norm1", sar))
l1r = keep(R, "l1", F.gelu(O.linear(sd, f"{P}.linear1", n1r)))
l2r = keep(R, "l2", n1r + O.linear(sd, f"{P}.linear2", l1r))
n2r = keep(R, "n2", O.layer_norm(sd, f"{P}.norm2", l2r))
n2r.backward(dy.float())
for k in T:
    print(f"{k:4s} fwd {rel_l2(T[k].detach().float().cpu(), R[k].detach()):.4f} grad {rel_l2(T[k].grad.float().cpu(), R[k].grad):.4f}  |ref grad| {float(R[k].grad.abs().mean()):.4g} |ref val| {float(R[k].detach().abs().mean()):.4g}")
