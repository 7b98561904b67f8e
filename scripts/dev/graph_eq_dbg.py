import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from focoos_amd.ports import DETRTargets
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image_structured, synth_state_dict
from focoos_amd.train_detr import FAIDetrTrainable, TrainStep
from oracle import train_oracle as T
DEV="cuda:0"
cfg = ModelRegistry.get_model_info("fai-detr-l-coco")["config"]
sd = synth_state_dict(cfg, 8)
B,(ih,iw)=4,(160,192)
runs={}
for graphs in (True, False):
    model = FAIDetrTrainable(cfg, norm="FrozenBN").to(DEV); model.load_state_dict(sd, strict=True)
    ts = TrainStep(model, lr=float(os.environ.get("LR","1e-4")), weight_decay=1e-4, graphs=graphs)
    out=[]
    for it in range(5):
        imgs = torch.from_numpy(np.stack([synth_image_structured(300 + it * B + i, ih, iw) for i in range(B)])).to(DEV)
        labels, boxes = T.synth_targets(60 + it, B, 80, counts=tuple(1 + (2 * i + 3 * it) % 7 for i in range(B)))
        targets = [DETRTargets(labels=l.to(DEV), boxes=b.to(DEV)) for l, b in zip(labels, boxes)]
        losses = ts.step(imgs, targets); torch.cuda.synchronize()
        out.append((torch.stack([losses[k].detach().float() for k in sorted(losses)]).cpu(), ts.opt.flat_g.clone(), ts.opt.flat_p.clone(), model.last_outputs["pred_logits"].detach().float().clone()))
    runs[graphs]=out
for it,((l1,g1,p1,o1),(l0,g0,p0,o0)) in enumerate(zip(runs[True],runs[False])):
    print(it, "loss rel", float((l1-l0).abs().max()/l0.abs().max()), "grad rel", float((g1-g0).norm()/g0.norm()), "param maxabs", float((p1-p0).abs().max()), "logits rel", float((o1-o0).norm()/o0.norm()))
