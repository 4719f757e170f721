import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import torch
from stylish_tts_amd import lib as L
lib = L.load()
def run(shape):
    B, Ci, Co, K, d, T = shape
    g = torch.Generator().manual_seed(sum(shape))
    x, w, b = torch.randn(B, Ci, T, generator=g), torch.randn(Co, Ci, K, generator=g) / (Ci * K) ** 0.5, torch.randn(Co, generator=g)
    gy = torch.randn(B, Co, T, generator=g)
    rnd = lambda t: t.bfloat16().double()
    pad = (K - 1) * d // 2
    xr, wr, gr = rnd(x).requires_grad_(True), rnd(w).requires_grad_(True), rnd(gy)
    (torch.nn.functional.conv1d(xr, wr, None, padding=pad, dilation=d) * gr).sum().backward()
    ref_dw = wr.grad.float()
    xd, wd, gd = x.cuda(), w.cuda(), gy.cuda()
    need = C.c_size_t()
    L.check(lib.sty_conv1d_bwd_workspace_bytes(B, Ci, Co, K, T, C.byref(need)))
    ws2 = torch.empty(need.value, dtype=torch.uint8, device="cuda")
    dw, db, dx = torch.empty(Co, Ci, K, device="cuda"), torch.empty(Co, device="cuda"), torch.empty(B, Ci, T, device="cuda")
    L.check(lib.sty_conv1d_bwd(B, Ci, Co, K, d, T, L.ptr(xd), L.ptr(wd), L.ptr(gd), L.ptr(dw), L.ptr(db), L.ptr(dx), L.ptr(ws2), ws2.numel(), 1, None))
    torch.cuda.synchronize()
    e = (dw.cpu() - ref_dw).abs() / ref_dw.abs().max()
    print(shape, "max rel", e.max().item())
    bad = e > 1e-3
    print(" bad frac", bad.float().mean().item(), "per co block32", [round(bad[i:i+32].float().mean().item(),2) for i in range(0,Co,32)],
          "per ci block32", [round(bad[:, i:i+32].float().mean().item(),2) for i in range(0,Ci,32)], "per k", [round(bad[:,:,k].float().mean().item(),2) for k in range(K)])
    ratio = (dw.cpu() / ref_dw)[bad]
    if bad.any(): print(" ratio sample", ratio[:8].tolist())
for s in [(3,128,130,1,1,64),(2,512,64,3,1,37),(2,96,200,3,1,300),(4,64,64,1,1,256),(1,64,64,1,1,128), (2,64,64,3,1,128)]:
    run(s)
