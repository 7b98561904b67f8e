import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_state_dict
from focoos_amd.train_nn import HybridEncoder, _ResizeFn
from focoos_amd import _lib as L
from oracle import detr_oracle as O
from tests.helpers import rel_l2
DEV = "cuda:0"
cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
sd = synth_state_dict(cfg, 12)
pre = "pixel_decoder.backbone."
enc = HybridEncoder(L.load()).to(DEV)
enc.load_state_dict({k[len("pixel_decoder."):]: v for k, v in sd.items() if k.startswith("pixel_decoder.") and not k.startswith(pre)}, strict=True)
lib = enc.lib
g = torch.Generator().manual_seed(0)
B, h, w = 2, 4, 5
if len(sys.argv) > 1:  # real backbone features
    from focoos_amd.train_nn import ResNetVd
    from focoos_amd.synth import synth_image_structured
    net = ResNetVd(50).to(DEV)
    net.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
    imgs = [synth_image_structured(60 + i, 128, 160) for i in range(2)]
    with torch.no_grad():
        ff = net(torch.from_numpy(np.stack(imgs)).to(DEV))
    f3, f4, f5 = ff["res3"].cpu(), ff["res4"].cpu(), ff["res5"].cpu()
    print("feature std", float(f3.float().std()), float(f4.float().std()), float(f5.float().std()))
else:
    f3 = torch.randn(B, 4 * h, 4 * w, 512, generator=g).clamp_min(0).bfloat16()
    f4 = torch.randn(B, 2 * h, 2 * w, 1024, generator=g).clamp_min(0).bfloat16()
    f5 = torch.randn(B, h, w, 2048, generator=g).clamp_min(0).bfloat16()
T = {}
def keep(name, t):
    t.retain_grad(); T[name] = t; return t
# ---- HIP graph (copy of HybridEncoder.forward with taps)
x3, x4, x5 = (keep(n, t.to(DEV).requires_grad_(True)) for n, t in (("f3", f3), ("f4", f4), ("f5", f5)))
proj = [keep(f"proj{i}", p(ff)) for i, (p, ff) in enumerate(zip(enc.input_proj, [x3, x4, x5]))]
src = proj[2].reshape(B, h * w, 256)
pos = enc._pos_for(h, w, src.device)
src = keep("aifi", enc.encoder[0].layers[0](src, pos))
proj2 = src.reshape(B, h, w, 256)
inner = [proj2]
for idx in (2, 1):
    high = keep(f"lat{2-idx}", enc.lateral_convs[2 - idx](inner[0]))
    inner[0] = high
    low = proj[idx - 1]
    up = keep(f"up{2-idx}", _ResizeFn.apply(high, low.shape[1], low.shape[2], lib))
    inner.insert(0, keep(f"fpn{2-idx}", enc.fpn_blocks[2 - idx](torch.cat([up, low], dim=-1))))
outs = [inner[0]]
for idx in range(2):
    nxt = inner[idx + 1]
    down = keep(f"down{idx}", enc.downsample_convs[idx](_ResizeFn.apply(outs[-1], nxt.shape[1], nxt.shape[2], lib)))
    outs.append(keep(f"pan{idx}", enc.pan_blocks[idx](torch.cat([down, nxt], dim=-1))))
pj = [torch.randn(256, generator=g) for _ in range(3)]
(sum((o.float() * p.to(DEV)).sum() for o, p in zip(outs[::-1], pj)) * 1e-2).backward()
# ---- oracle with the same taps
R = {}
def keepr(name, t):
    t.retain_grad(); R[name] = t; return t
P = "pixel_decoder"
r3, r4, r5 = (keepr(n, t.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)) for n, t in (("f3", f3), ("f4", f4), ("f5", f5)))
rproj = [keepr(f"proj{i}", O.conv_bn(sd, f"{P}.input_proj.{i}", ff, 1, None, conv="0", norm="1")) for i, ff in enumerate([r3, r4, r5])]
rsrc = rproj[2].flatten(2).permute(0, 2, 1)
rsrc = keepr("aifi", O.encoder_layer(sd, f"{P}.encoder.0.layers.0", rsrc, O.position_embedding_sine(h, w, 128), 8))
rproj2 = rsrc.permute(0, 2, 1).reshape(B, 256, h, w)
rinner = [rproj2]
for idx in (2, 1):
    high = keepr(f"lat{2-idx}", O.conv_bn(sd, f"{P}.lateral_convs.{2 - idx}", rinner[0], 1, "silu"))
    rinner[0] = high
    low = rproj[idx - 1]
    up = keepr(f"up{2-idx}", F.interpolate(high, size=low.shape[-2:], mode="bilinear"))
    rinner.insert(0, keepr(f"fpn{2-idx}", O.csp_rep_layer(sd, f"{P}.fpn_blocks.{2 - idx}", torch.cat([up, low], 1))))
routs = [rinner[0]]
for idx in range(2):
    down = F.interpolate(routs[-1], size=rinner[idx + 1].shape[-2:], mode="bilinear")
    down = keepr(f"down{idx}", O.conv_bn(sd, f"{P}.downsample_convs.{idx}", down, 1, "silu"))
    routs.append(keepr(f"pan{idx}", O.csp_rep_layer(sd, f"{P}.pan_blocks.{idx}", torch.cat([down, rinner[idx + 1]], 1))))
(sum((o * p.view(1, -1, 1, 1)).sum() for o, p in zip(routs[::-1], pj)) * 1e-2).backward()
def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2) if t.dim() == 4 else t.float().cpu()
for k in T:
    a, b = nchw(T[k].detach()), R[k].detach()
    ga, gb = nchw(T[k].grad), R[k].grad
    print(f"{k:8s} fwd {rel_l2(a, b):.4f}  grad {rel_l2(ga, gb):.4f}")
