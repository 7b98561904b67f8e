import numpy as np, torch, time, sys
sys.path.insert(0, ".")
from focoos_amd.model import FAIDetr
from focoos_amd.registry import ModelRegistry
from focoos_amd.synth import synth_image_structured as sis
cfg = ModelRegistry.get_model_info("fai-detr-l-obj365")["config"]
eng = FAIDetr(cfg, device="cuda:0", seed=0).engine
B = 32
imgs = torch.from_numpy(np.stack([sis(100 + i) for i in range(B)])).to("cuda:0")
pl = eng.plan(B, 640, 640, False, 2)
st = eng.stream
with torch.cuda.stream(st):
    pl.input.copy_(imgs); pl.sizes.copy_(torch.tensor([[640, 640]] * B, dtype=torch.int32))
    for p in pl.parts: p._launch(p.ops, st.cuda_stream, 0.3)
st.synchronize()
ref = (pl.probs.clone(), pl.boxes.clone(), pl.det_count.clone())
bad = 0; N = 3000; t0 = time.time()
for i in range(N):
    with torch.cuda.stream(st):
        pl.run(st.cuda_stream, 0.3, None, True)
    st.synchronize()
    bad += not (torch.equal(ref[0], pl.probs) and torch.equal(ref[1], pl.boxes) and torch.equal(ref[2], pl.det_count))
print(f"LONGRUN: {bad} of {N} concurrent replays differ from the serial result ({time.time() - t0:.1f} s)")
