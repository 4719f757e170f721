"""Kernel-tuning aid: time sty_conv1d_fwd over a list of shapes x forced tile configurations (STY_CONV_CFG).

    python tools/conv_sweep.py            # on the GPU box; prints us / TFLOP/s per (shape, config)
Each configuration runs in a fresh process (the forced configuration is read once per process)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [  # (B, Cin, Cout, K, dil, T)  -- c2 shapes of the training step (forward and input-gradient forms)
    (16, 1024, 256, 1, 1, 160), (16, 256, 1024, 1, 1, 160), (16, 128, 128, 1, 1, 37), (16, 512, 128, 3, 1, 37),
    (16, 128, 512, 3, 1, 37), (16, 1920, 384, 5, 1, 210), (16, 384, 1920, 5, 1, 210), (16, 1152, 384, 3, 1, 210),
    (16, 960, 320, 3, 1, 820), (16, 320, 960, 3, 1, 820), (16, 480, 160, 3, 1, 3240), (16, 240, 80, 3, 1, 12880),
    (16, 32, 128, 1, 1, 12000), (16, 128, 32, 1, 1, 12000), (16, 32, 32, 11, 1, 12000), (16, 32, 32, 21, 1, 12000),
    (16, 128, 128, 3, 1, 160), (16, 195, 128, 3, 1, 160), (1, 2048, 2050, 1, 1, 1504),
]


def run_one():
    import torch
    from stylish_tts_amd import lib as L
    lib = L.load()
    dev = torch.device("cuda")
    out = []
    for (B, Ci, Co, K, d, T) in SHAPES:
        x = torch.randn(B, Ci, T, device=dev)
        w = torch.randn(Co, Ci, K, device=dev) * 0.05
        b = torch.randn(Co, device=dev)
        y = torch.empty(B, Co, T, device=dev)
        need = C.c_size_t()
        L.check(lib.sty_conv1d_workspace_bytes(Co, Ci, K, C.byref(need)))
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        args = (B, Ci, Co, K, d, T, L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(ws), ws.numel(),
                int(os.environ.get("SWEEP_BF16", "0")), st)
        for _ in range(3):
            L.check(lib.sty_conv1d_fwd(*args))
        lib.sty_prof_enable(1)
        for _ in range(20):
            L.check(lib.sty_conv1d_fwd(*args))
        lib.sty_prof_enable(0)
        rows = [r for r in L.prof_report(64) if r["name"].startswith("conv1d_mfma")]
        r = rows[0]
        us = 1e3 * r["ms"] / r["launches"]
        out.append(f"{r['name']:28s} B{B} ci{Ci} co{Co} k{K} T{T}: {us:8.1f} us {r['flops'] / r['launches'] / us / 1e6:6.1f} TF")
    print("\n".join(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        run_one()
    else:
        for cfg in os.environ.get("SWEEP_CFGS", ",0,1,2,3,4,5,6").split(","):
            env = dict(os.environ)
            if cfg:
                env["STY_CONV_CFG"] = cfg
            print(f"--- STY_CONV_CFG={cfg or 'auto'}")
            sys.stdout.flush()
            subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env)
