import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import torch
from stylish_tts_amd import lib as L
lib = L.load()
def run(x, gy, K=1, d=1):
    B, Ci, T = x.shape; Co = gy.shape[1]
    w = torch.zeros(Co, Ci, K)
    xd, wd, gd = x.cuda(), w.cuda(), gy.cuda()
    need = C.c_size_t()
    L.check(lib.sty_conv1d_bwd_workspace_bytes(B, Ci, Co, K, T, C.byref(need)))
    ws2 = torch.empty(need.value, dtype=torch.uint8, device="cuda")
    dw, db, dx = torch.empty(Co, Ci, K, device="cuda"), torch.empty(Co, device="cuda"), torch.empty(B, Ci, T, device="cuda")
    L.check(lib.sty_conv1d_bwd(B, Ci, Co, K, d, T, L.ptr(xd), L.ptr(wd), L.ptr(gd), L.ptr(dw), L.ptr(db), L.ptr(dx), L.ptr(ws2), ws2.numel(), 1, None))
    torch.cuda.synchronize()
    return dw.cpu(), db.cpu()
T=128
x = torch.arange(64).float()[None,:,None].expand(1,64,T).contiguous()
g = torch.ones(1,64,T)
dw,db = run(x,g)
print("x=ci, g=1: dw[0,:8]", dw[0,:8,0].tolist(), "dw[:8,1]", dw[:8,1,0].tolist(), "db", db[:4].tolist())
x = torch.ones(1,64,T); g = torch.arange(64).float()[None,:,None].expand(1,64,T).contiguous()
dw,db = run(x,g)
print("x=1, g=co: dw[:8,0]", dw[:8,0,0].tolist(), "dw[1,:8]", dw[1,:8,0].tolist(), "db", db[:4].tolist())
x = torch.zeros(1,64,T); x[0,:,5]=1; g = torch.zeros(1,64,T); g[0,:,5]=1
dw,db = run(x,g)
print("delta t=5 both: dw[:4,:4]", dw[:4,:4,0].tolist())
g = torch.zeros(1,64,T); g[0,:,6]=1
dw,db = run(x,g)
print("x delta 5, g delta 6: dw[:2,:4]", dw[:2,:4,0].tolist())
x = torch.arange(T).float()[None,None,:].expand(1,64,T).contiguous()
for j in (0,1,5,7,8,9,16,40,127):
    g = torch.zeros(1,64,T); g[0,:,j]=1
    dw,db = run(x,g)
    print("x=t ramp, g delta at", j, "-> dw[0,0] =", dw[0,0,0].item(), " db", db[0].item())
g = torch.arange(T).float()[None,None,:].expand(1,64,T).contiguous()
for j in (0,5,9):
    x = torch.zeros(1,64,T); x[0,:,j]=1
    dw,db = run(x,g)
    print("g=t ramp, x delta at", j, "-> dw[0,0] =", dw[0,0,0].item())
