"""A/B microbenchmark of the twin-operand conv (convq.hip, round 6) against convp16_kernel<.., X16> through sty_conv1d_fwd with
compute_bf16 = 2 on 1-D stand-ins of the style encoder's c3 layers (same Cin x taps, Cout and column count as the flat 3x3 convs):
    python tools/convq_bench.py [reps]
prints the library's own per-kernel timing (HIP events around the launch), TFLOP/s and whether the two outputs are bit-equal."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from stylish_tts_amd import lib as L
lib = L.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
SHAPES = [  # B, Ci, Co, K, T   (Ci = 3 x the image channels: the (kh, ci) reduction axis of a flat 3x3)
    (32, 240, 80, 3, 41680),   # block 1 convs
    (32, 480, 160, 3, 10440),  # block 2
    (32, 240, 160, 3, 10440),  # block 2, first conv (80 -> 160)
    (32, 480, 80, 3, 10440),   # its input gradient
    (32, 960, 320, 3, 2620),   # block 3 (T % 8 != 0)
    (32, 1152, 384, 3, 660),   # block 4
    (32, 80, 160, 1, 10440),   # 1x1 shortcut
    (32, 512, 512, 3, 520),    # decoder k3 shape
]
sel = os.environ.get("SHAPES")
for B, Ci, Co, K, T in [SHAPES[int(i)] for i in sel.split(",")] if sel else SHAPES:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Ci, T, generator=g).cuda()
    w = (torch.randn(Co, Ci, K, generator=g) / (Ci * K) ** 0.5).cuda()
    b = torch.randn(Co, generator=g).cuda()
    need = C.c_size_t(); L.check(lib.sty_conv1d_workspace_bytes(Co, Ci, K, C.byref(need)))
    ws = torch.empty(need.value + B * (Ci + Co) * T * 2 + 1024, dtype=torch.uint8, device="cuda")
    flops = 2.0 * B * Ci * Co * K * T
    outs = {}
    for which in ("convp16", "convq"):
        if which == "convp16":
            os.environ["STY_NO_CONVQ"] = "1"
        else:
            os.environ.pop("STY_NO_CONVQ", None)
        y = torch.empty(B, Co, T, device="cuda")
        for _ in range(2):
            L.check(lib.sty_conv1d_fwd(B, Ci, Co, K, 1, T, L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(ws), ws.numel(), 2, None))
        torch.cuda.synchronize()
        L.prof_report(64)
        lib.sty_prof_enable(1)
        for _ in range(reps):
            L.check(lib.sty_conv1d_fwd(B, Ci, Co, K, 1, T, L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(ws), ws.numel(), 2, None))
        torch.cuda.synchronize()
        lib.sty_prof_enable(0)
        outs[which] = y.clone()
        for r in L.prof_report(64):
            if r["name"].startswith(("convp16", "convq", "convk1", "conv1d_mfma")):
                us = r["ms"] * 1e3 / max(1, r["launches"])
                print(f"B{B} ci{Ci} co{Co} k{K} T{T}: {r['name']:28s} {us:9.1f} us  {flops / us / 1e6:7.1f} TF", flush=True)
    print(f"    bit-equal: {torch.equal(outs['convp16'], outs['convq'])}", flush=True)
