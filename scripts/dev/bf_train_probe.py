"""Stage-by-stage comparison of the trainable BiSeNetFormer graph against the training oracle (teacher-forced attention masks)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image_structured, synth_state_dict
from focoos_amd.train_bf import BisenetFormerTrainable
from oracle import detr_oracle as O, train_oracle as T
from tests.helpers import rel_l2
norm = sys.argv[1] if len(sys.argv) > 1 else "FrozenBN"
DEV = "cuda:0"
cfg = dict(ModelRegistry.get_model_info("bisenetformer-l-ade")["config"], criterion_num_points=2048)
sd = synth_state_dict(cfg, 31, family="bisenetformer")
nimg = 4 if norm == "BN" else 2
imgs = [synth_image_structured(60 + i, 192, 256) for i in range(nimg)]
x = O.get_torch_batch(imgs, None)
col = {}
O.BN_TRAINING[0] = norm != "FrozenBN"
with torch.no_grad():
    sdc = {k: v.clone() for k, v in sd.items()}
    outs = T.bf_train_outputs(sdc, cfg, x, collect=col)
O.BN_TRAINING[0] = False
model = BisenetFormerTrainable(cfg, norm=norm).to(DEV)
model.load_state_dict(sd, strict=True); model.train()
x_u8 = torch.from_numpy(np.stack(imgs)).to(DEV)
nchw = lambda t: t.float().cpu().permute(0, 3, 1, 2)
with torch.no_grad():
    bbm = model.pixel_decoder.backbone
    xx = bbm.features[0](x_u8)
    import torch.nn.functional as Fn
    from oracle import bf_oracle as BF
    O.BN_TRAINING[0] = norm != "FrozenBN"
    mean = torch.tensor([123.675, 116.28, 103.53]).view(-1, 1, 1); std = torch.tensor([58.395, 57.12, 57.375]).view(-1, 1, 1)
    sd2 = {k: v.clone() for k, v in sd.items()}
    xo = BF.conv_x(sd2, "pixel_decoder.backbone.features.0", (x - mean) / std, 2)
    print("features.0", rel_l2(nchw(xx), xo))
    xx1 = bbm.features[1](xx); xo1 = BF.conv_x(sd2, "pixel_decoder.backbone.features.1", xo, 2)
    print("features.1", rel_l2(nchw(xx1), xo1))
    blk = bbm.features[2]
    o1 = blk.conv_list[0](xx1); oo1 = BF.conv_x(sd2, "pixel_decoder.backbone.features.2.conv_list.0", xo1)
    print("f2.conv0", rel_l2(nchw(o1), oo1))
    av = blk.avd_layer(o1)
    w = sd2["pixel_decoder.backbone.features.2.avd_layer.0.weight"]
    ao = BF._bn(sd2, "pixel_decoder.backbone.features.2.avd_layer.1", Fn.conv2d(oo1, w, None, stride=2, padding=1, groups=w.shape[0]))
    print("f2.avd", rel_l2(nchw(av), ao))
    # teacher-forced per block: the oracle's block input through the engine block vs the oracle's block output
    cur = xo1
    idx = 2
    for i, n in enumerate((4, 5, 3)):
        for j in range(n):
            nxt = BF.cat_bottleneck(sd2, f"pixel_decoder.backbone.features.{idx}", cur, 2 if j == 0 else 1)
            eng = bbm.features[idx](cur.permute(0, 2, 3, 1).contiguous().to(DEV).bfloat16())
            ch = nxt.shape[1]
            parts = [0, ch // 2, ch // 2 + ch // 4, ch // 2 + ch // 4 + ch // 8, ch]
            pe = [round(rel_l2(nchw(eng)[:, parts[k]:parts[k + 1]], nxt[:, parts[k]:parts[k + 1]]), 4) for k in range(4)]
            print(f"block {idx} (stride {2 if j == 0 else 1}, {tuple(nxt.shape)}): rel_l2 {rel_l2(nchw(eng), nxt):.4f} branches {pe} min|var| chk", flush=True)
            cur = nxt
            idx += 1
    O.BN_TRAINING[0] = False
    f = model.pixel_decoder.backbone(x_u8)
    for k in ("res2", "res3", "res4", "res5"):
        print(k, rel_l2(nchw(f[k]), col[k]) if k in col else "n/a")
    mf, msf = model.pixel_decoder.decode(f)
    print("cp32", rel_l2(nchw(msf[0]), col["cp32"]), "cp16", rel_l2(nchw(msf[1]), col["cp16"]), "cp8", rel_l2(nchw(msf[2]), col["cp8"]))
    print("ffm", "mask_features", rel_l2(nchw(mf), col["mask_features"]))
    out = model.head.predictor(msf[:-1], mf, col["attn_masks"])
    sets = [outs] if False else None
    allo = outs["aux_outputs"] + [{"pred_logits": outs["pred_logits"], "pred_masks": outs["pred_masks"]}]
    alle = out["aux_outputs"] + [{"pred_logits": out["pred_logits"], "pred_masks": out["pred_masks"]}]
    for i, (a, b) in enumerate(zip(alle, allo)):
        print("head", i, "cls", rel_l2(a["pred_logits"].float().cpu(), b["pred_logits"]), "masks", rel_l2(a["pred_masks"].float().cpu(), b["pred_masks"]))
        if i == len(alle) - 1:
            pq = [(round(rel_l2(a["pred_masks"][0, q].float().cpu(), b["pred_masks"][0, q]), 3)) for q in range(0, 100, 7)]
            print("  per-query (img 0):", pq)
