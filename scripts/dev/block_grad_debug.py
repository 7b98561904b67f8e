import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_state_dict
from focoos_amd.train_nn import HybridEncoder
from focoos_amd import _lib as L
from oracle import detr_oracle as O
from tests.helpers import rel_l2
DEV = "cuda:0"
cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
sd = synth_state_dict(cfg, 12)
pre = "pixel_decoder.backbone."
enc = HybridEncoder(L.load()).to(DEV)
enc.load_state_dict({k[len("pixel_decoder."):]: v for k, v in sd.items() if k.startswith("pixel_decoder.") and not k.startswith(pre)}, strict=True)
g = torch.Generator().manual_seed(0)
def cmp(name, mod_fn, ref_fn, shape):
    x = torch.randn(*shape, generator=g).bfloat16()
    xd = x.to(DEV).requires_grad_(True)
    y = mod_fn(xd)
    dy = torch.randn(*y.shape, generator=g).bfloat16()
    y.backward(dy.to(DEV))
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True) if len(shape) == 4 else x.float().requires_grad_(True)
    yr = ref_fn(xr)
    yr.backward(dy.float().permute(0, 3, 1, 2) if len(shape) == 4 else dy.float())
    yo = y.detach().float().cpu().permute(0, 3, 1, 2) if len(shape) == 4 else y.detach().float().cpu()
    go = xd.grad.float().cpu().permute(0, 3, 1, 2) if len(shape) == 4 else xd.grad.float().cpu()
    print(f"{name:28s} out rel-L2 {rel_l2(yo, yr.detach()):.4f}  dx rel-L2 {rel_l2(go, xr.grad):.4f}")
P = "pixel_decoder"
cmp("input_proj.2 (1x1 2048->256)", lambda x: enc.input_proj[2](x), lambda x: O.conv_bn(sd, f"{P}.input_proj.2", x, 1, None, conv="0", norm="1"), (2, 4, 5, 2048))
cmp("lateral_convs.0 (silu)", lambda x: enc.lateral_convs[0](x), lambda x: O.conv_bn(sd, f"{P}.lateral_convs.0", x, 1, "silu"), (2, 8, 10, 256))
cmp("downsample_convs.0 (3x3 silu)", lambda x: enc.downsample_convs[0](x), lambda x: O.conv_bn(sd, f"{P}.downsample_convs.0", x, 1, "silu"), (2, 8, 10, 256))
cmp("repvgg", lambda x: enc.fpn_blocks[0].bottlenecks[0](x), lambda x: O.rep_vgg_block(sd, f"{P}.fpn_blocks.0.bottlenecks.0", x), (2, 8, 10, 256))
cmp("csp fpn_blocks.0", lambda x: enc.fpn_blocks[0](x), lambda x: O.csp_rep_layer(sd, f"{P}.fpn_blocks.0", x), (2, 8, 10, 512))
pos = O.position_embedding_sine(4, 5, 128)
posd = pos[0].to(DEV).bfloat16()
cmp("aifi layer (L=20)", lambda x: enc.encoder[0].layers[0](x, posd), lambda x: O.encoder_layer(sd, f"{P}.encoder.0.layers.0", x, pos, 8), (2, 20, 256))
pos = O.position_embedding_sine(10, 10, 128); posd = pos[0].to(DEV).bfloat16()
cmp("aifi layer (L=100)", lambda x: enc.encoder[0].layers[0](x, posd), lambda x: O.encoder_layer(sd, f"{P}.encoder.0.layers.0", x, pos, 8), (2, 100, 256))
