"""Debug aid: where does convp16 disagree with float64 on a given shape?  python tools/convp16_debug.py B Ci Co K d T"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from stylish_tts_amd import lib as L
os.environ["STY_CONVP16_MIN_TILES"] = "1"
lib = L.load()
B, Ci, Co, K, d, T = [int(x) for x in sys.argv[1:7]]
g = torch.Generator().manual_seed(1)
x, w, b = torch.randn(B, Ci, T, generator=g), torch.randn(Co, Ci, K, generator=g) / (Ci * K) ** 0.5, torch.randn(Co, generator=g)
rnd = lambda t: t.bfloat16().double()
ref = torch.nn.functional.conv1d(rnd(x), rnd(w), b.double(), padding=(K - 1) * d // 2, dilation=d).float()
xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
y = torch.full((B, Co, T), 777.0, device="cuda")
need = C.c_size_t(); L.check(lib.sty_conv1d_workspace_bytes(Co, Ci, K, C.byref(need)))
ws = torch.empty(need.value, dtype=torch.uint8, device="cuda")
lib.sty_prof_enable(1)
L.check(lib.sty_conv1d_fwd(B, Ci, Co, K, d, T, L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(y), L.ptr(ws), ws.numel(), 1, None))
torch.cuda.synchronize(); lib.sty_prof_enable(0)
print([r["name"] for r in L.prof_report(64)])
e = (y.cpu() - ref).abs()
print("max err", e.max().item(), "untouched", (y == 777.0).sum().item())
bad = (e > 1e-3)
print("bad fraction", bad.float().mean().item())
print("bad per batch", bad.float().mean((1, 2)).tolist())
print("bad per cout (first 40)", [round(v, 2) for v in bad.float().mean((0, 2)).tolist()[:40]])
tb = bad.float().mean((0, 1))
print("bad per column block of 16:", [round(tb[i:i + 16].mean().item(), 2) for i in range(0, T, 16)])
