"""Kernel-tuning aid: time the weight-gradient kernels in isolation through sty_conv1d_bwd (dx not requested).

    [SWEEP_BF16=1] [STY_LIB_VARIANT=name] python tools/wgrad_sweep.py      # on the GPU box
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [  # (B, Cin, Cout, K, dil, T): c2 / c3 shapes of the style encoder, decoder, vocoder
    (16, 240, 80, 3, 1, 12880), (16, 480, 160, 3, 1, 3240), (16, 960, 320, 3, 1, 820), (16, 1920, 384, 5, 1, 210),
    (32, 240, 80, 3, 1, 41680), (32, 960, 320, 3, 1, 2620), (32, 1152, 384, 3, 1, 660),
    (16, 32, 32, 11, 1, 12000), (32, 32, 32, 11, 1, 39000), (32, 32, 32, 21, 1, 39000),
    (16, 256, 1024, 1, 1, 160), (32, 256, 1024, 1, 1, 520), (32, 32, 128, 1, 1, 39000), (32, 128, 32, 1, 1, 39000),
]


def main():
    import torch
    from stylish_tts_amd import lib as L
    lib = L.load()
    dev = torch.device("cuda")
    bf = int(os.environ.get("SWEEP_BF16", "0"))
    for (B, Ci, Co, K, d, T) in SHAPES:
        x = torch.randn(B, Ci, T, device=dev)
        w = torch.randn(Co, Ci, K, device=dev) * 0.05
        gy = torch.randn(B, Co, T, device=dev)
        dw = torch.empty(Co, Ci, K, device=dev)
        db = torch.empty(Co, device=dev)
        need = C.c_size_t()
        L.check(lib.sty_conv1d_bwd_workspace_bytes(B, Ci, Co, K, T, C.byref(need)))
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        args = (B, Ci, Co, K, d, T, L.ptr(x), L.ptr(w), L.ptr(gy), L.ptr(dw), L.ptr(db) if K <= 12 else None, None,
                L.ptr(ws), ws.numel(), bf, st)
        for _ in range(3):
            L.check(lib.sty_conv1d_bwd(*args))
        torch.cuda.synchronize()
        lib.sty_prof_enable(1)
        for _ in range(10):
            L.check(lib.sty_conv1d_bwd(*args))
        torch.cuda.synchronize()
        lib.sty_prof_enable(0)
        for r in L.prof_report(64):
            if "wgrad" in r["name"]:
                us = 1e3 * r["ms"] / r["launches"]
                print(f"{r['name'][:44]:44s} B{B} ci{Ci} co{Co} k{K} T{T}: {us:8.1f} us "
                      f"{r['flops'] / r['launches'] / us / 1e6:6.1f} TF {r['bytes'] / r['launches'] / us / 1e3:6.0f} GB/s")


if __name__ == "__main__":
    main()
