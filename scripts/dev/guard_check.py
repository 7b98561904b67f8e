import sys, os
os.environ["FX_GUARD"] = "4096"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from focoos_amd.model import FAIDetr
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image_structured as sis
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
model = FAIDetr(cfg, device="cuda:0", seed=0)
eng = model.engine
for B in (16, 32, 2):
    imgs = torch.from_numpy(np.stack([sis(100 + i) for i in range(B)])).to("cuda:0")
    pl = eng.plan(B, 640, 640, False, 1)
    with torch.cuda.stream(eng.stream):
        pl.input.copy_(imgs); pl.sizes.copy_(torch.tensor([[640, 640]] * B, dtype=torch.int32))
        pl.run(eng.stream.cuda_stream, 0.3, None, False)
    eng.stream.synchronize()
    print("B", B, "corrupted guards:", pl.check_guards())
