"""Phase timing of convq_kernel<3, 3, 0> by switching its parts off (STY_CQ_DBG, compile-time variants; results are wrong in every
mode but 0):  python tools/convq_phases.py [reps]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from stylish_tts_amd import lib as L
lib = L.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
SHAPES = [(32, 480, 160, 3, 10440), (32, 240, 80, 3, 41680), (32, 1152, 384, 3, 660)]
MODES = [(0, "full"), (3, "no global loads (offsets out of range)"), (4, "no loads, no LDS writes"), (8, "no fragment reads / MFMAs"),
         (16, "no epilogue"), (24, "staging only"), (32, "fragment reads, no MFMAs"), (48, "fragment reads, no MFMAs, no epilogue"),
         (7, "no loads, no commits (same as 4)")]
for B, Ci, Co, K, T in SHAPES:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Ci, T, generator=g).cuda()
    w = (torch.randn(Co, Ci, K, generator=g) / (Ci * K) ** 0.5).cuda()
    y = torch.empty(B, Co, T, device="cuda")
    need = C.c_size_t(); L.check(lib.sty_conv1d_workspace_bytes(Co, Ci, K, C.byref(need)))
    ws = torch.empty(need.value + B * (Ci + Co) * T * 2 + 1024, dtype=torch.uint8, device="cuda")
    flops = 2.0 * B * Ci * Co * K * T
    print(f"B{B} ci{Ci} co{Co} k{K} T{T}  ({flops / 1e9:.1f} GF; MFMA-bound at 96-cout tiles: "
          f"{-(-Co // 96) * 96 * 2.0 * B * Ci * K * T / 2.5e15 * 1e6:.1f} us)", flush=True)
    for m, what in MODES:
        os.environ["STY_CQ_DBG"] = str(m)
        for _ in range(2):
            L.check(lib.sty_conv1d_fwd(B, Ci, Co, K, 1, T, L.ptr(x), L.ptr(w), None, L.ptr(y), L.ptr(ws), ws.numel(), 2, None))
        torch.cuda.synchronize()
        L.prof_report(64)
        lib.sty_prof_enable(1)
        for _ in range(reps):
            L.check(lib.sty_conv1d_fwd(B, Ci, Co, K, 1, T, L.ptr(x), L.ptr(w), None, L.ptr(y), L.ptr(ws), ws.numel(), 2, None))
        torch.cuda.synchronize()
        lib.sty_prof_enable(0)
        for r in L.prof_report(64):
            if r["name"].startswith("convq"):
                print(f"   dbg {m:2d} {what:48s} {r['ms'] * 1e3 / max(1, r['launches']):8.1f} us", flush=True)
    os.environ.pop("STY_CQ_DBG", None)
