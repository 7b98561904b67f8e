import ctypes as C, sys
import torch, torch.nn.functional as F
sys.path.insert(0, ".")
from focoos_amd import _lib
from focoos_amd._lib import check
lib = _lib.load()
DEV = "cuda:0"
g = torch.Generator().manual_seed(11)
B, H, W = 2, 38, 50
img = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
Wt = torch.randn(32, 3, 3, 3, generator=g) * 0.2
bias = torch.randn(32, generator=g) * 0.1
mean, std = torch.tensor([123.675, 116.28, 103.53]), torch.tensor([58.395, 57.12, 57.375])
Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
y = torch.empty(B, Ho, Wo, 32, dtype=torch.bfloat16, device=DEV)
wd, bd, md, sd_ = Wt.permute(2, 3, 1, 0).contiguous().to(DEV), bias.to(DEV), mean.to(DEV), (1.0 / std).to(DEV)
xin = img.to(DEV)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
check(lib.fx_stem_conv3x3s2(xin.data_ptr(), 0, wd.data_ptr(), bd.data_ptr(), md.data_ptr(), sd_.data_ptr(), y.data_ptr(), B, H, W, 32, st))
torch.cuda.synchronize()
xn = (img.float().permute(0, 3, 1, 2) - mean.view(-1, 1, 1)) / std.view(-1, 1, 1)
ref = F.relu(F.conv2d(xn, Wt, bias, stride=2, padding=1)).permute(0, 2, 3, 1)
err = (y.float().cpu() - ref).abs()
print("max rel err", float(err.max() / ref.abs().max()), "mean abs err", float(err.mean()), "ref mean", float(ref.abs().mean()))
print("err by row (ho):", [round(float(err[:, i].max()), 3) for i in range(Ho)])
print("err by col (wo):", [round(float(err[:, :, i].max()), 3) for i in range(Wo)])
print("err by channel:", [round(float(err[..., c].max()), 3) for c in range(32)])
# taps contribution test: one-hot weights
for kh, kw, c in ((0, 0, 0), (0, 2, 2), (1, 1, 1), (2, 0, 1), (2, 2, 2), (1, 2, 2)):
    W1 = torch.zeros(32, 3, 3, 3); W1[:, c, kh, kw] = 1.0
    wd1 = W1.permute(2, 3, 1, 0).contiguous().to(DEV)
    zb = torch.zeros(32, device=DEV)
    check(lib.fx_stem_conv3x3s2(xin.data_ptr(), 0, wd1.data_ptr(), zb.data_ptr(), md.data_ptr(), sd_.data_ptr(), y.data_ptr(), B, H, W, 32, st))
    torch.cuda.synchronize()
    r1 = F.relu(F.conv2d(xn, W1, None, stride=2, padding=1)).permute(0, 2, 3, 1)
    e1 = (y.float().cpu() - r1).abs()
    print(f"tap kh={kh} kw={kw} c={c}: max err {float(e1.max()):.3f} (ref max {float(r1.max()):.2f}); interior err {float(e1[:, 1:-1, 1:-1].max()):.3f}")
