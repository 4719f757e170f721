"""Microbenchmark of the bf16 persistent conv (convp16.hip) through sty_conv1d_fwd on 1-D stand-ins of the c3 layers:
    python tools/convp16_bench.py [reps]
prints the library's own per-kernel timing (HIP events around the launch) and the achieved TFLOP/s."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from stylish_tts_amd import lib as L
lib = L.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
BF = int(os.environ.get("BF16", "1"))  # 0: the fp32 twin (convpf.hip)
SHAPES = [  # B, Ci, Co, K, T
    (32, 80, 80, 3, 41600 // 8),    # style encoder block 1 (1/8 of the image rows)
    (32, 160, 160, 3, 10400 // 2),  # block 2
    (32, 320, 320, 3, 2600),        # block 3
    (32, 384, 384, 3, 650),         # block 4
    (32, 512, 512, 3, 520),         # decoder k3
    (32, 1024, 512, 1, 1040),       # decoder 1x1
    (32, 256, 1024, 1, 520),        # conformer ff
]
sel = os.environ.get('SHAPES')
for B, Ci, Co, K, T in [SHAPES[int(i)] for i in sel.split(',')] if sel else SHAPES:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Ci, T, generator=g).cuda()
    w = (torch.randn(Co, Ci, K, generator=g) / (Ci * K) ** 0.5).cuda()
    b = torch.randn(Co, generator=g).cuda()
    y = torch.empty(B, Co, T, device="cuda")
    need = C.c_size_t(); L.check(lib.sty_conv1d_workspace_bytes(Co, Ci, K, C.byref(need)))
    ws = torch.empty(need.value, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        L.check(lib.sty_conv1d_fwd(B, Ci, Co, K, 1, T, L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(ws), ws.numel(), BF, None))
    torch.cuda.synchronize()
    lib.sty_prof_enable(1)
    for _ in range(reps):
        L.check(lib.sty_conv1d_fwd(B, Ci, Co, K, 1, T, L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(ws), ws.numel(), BF, None))
    torch.cuda.synchronize()
    lib.sty_prof_enable(0)
    flops = 2.0 * B * Ci * Co * K * T
    for r in L.prof_report(64):
        if "conv" in r["name"] and "pack" not in r["name"]:
            us = r["ms"] * 1e3 / max(1, r["launches"])
            print(f"B{B} ci{Ci} co{Co} k{K} T{T}: {r['name']:34s} {us:9.1f} us  {flops / us / 1e6:7.1f} TF  ({flops / 1e9:.1f} GF)")
