"""Kernel-tuning aid: conv32p_kernel (persistent 32 -> 32 channel conv) vs conv1d_mfma_kernel on the 75T-rate shapes
of configs c5 / c3, fp32 and bf16 operands, through the unit entry point sty_conv1d_fwd.

    python tools/conv32p_bench.py            # on the GPU box; prints us, TFLOP/s, algorithmic GB/s per variant
STY_P_DBG bits (measurement only, results are wrong): 1 = one tap, 2 = no staging after the first tile, 4 = no epilogue.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(8, 32, 32, 11, 1, 60000), (8, 32, 32, 11, 5, 60000), (8, 32, 32, 21, 1, 60000), (32, 32, 32, 11, 3, 39000),
          (32, 32, 32, 1, 1, 39000)]


def main():
    import torch
    from stylish_tts_amd import lib as L
    lib = L.load()
    dev = torch.device("cuda")
    only = [int(a) for a in sys.argv[1:2]]          # optional: shape index
    modes = (int(sys.argv[2]),) if len(sys.argv) > 2 else (0, 1)   # optional: 0 = fp32, 1 = bf16
    plain = len(sys.argv) > 3                       # optional third argument: only the plain conv32p variant (PMC runs)
    for (B, Ci, Co, K, d, T) in ([SHAPES[i] for i in only] if only else SHAPES):
        x = torch.randn(B, Ci, T, device=dev)
        w = torch.randn(Co, Ci, K, device=dev) * 0.05
        b = torch.randn(Co, device=dev)
        y = torch.empty(B, Co, T, device=dev)
        need = C.c_size_t()
        L.check(lib.sty_conv1d_workspace_bytes(Co, Ci, K, C.byref(need)))
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for bf in modes:
            variants = (("old kernel", dict(STY_NO_CONV32P_CALL="1")), ("conv32p", {}),
                        ("conv32p bf16 source", dict(STY_P_FORCE_H="1")), ("conv32p bf16 output", dict(STY_P_FORCE_H="2")),
                        ("conv32p bf16 source+output", dict(STY_P_FORCE_H="3")),
                        ("conv32p bf16 src+out, no staging", dict(STY_P_FORCE_H="3", STY_P_DBG="2")),
                        ("conv32p bf16 src+out, no epilogue", dict(STY_P_FORCE_H="3", STY_P_DBG="4")),
                        ("conv32p dbg=1 (1 tap)", dict(STY_P_DBG="1")),
                             ("conv32p dbg=2 (no staging)", dict(STY_P_DBG="2")), ("conv32p dbg=4 (no epilogue)", dict(STY_P_DBG="4")),
                             ("conv32p dbg=6 (MFMA only)", dict(STY_P_DBG="6")))
            for tag, env in (variants[1:2] if plain else variants):
                if bf == 0 and "STY_P_FORCE_H" in env:
                    continue
                for k in ("STY_P_DBG", "STY_CONV32P_MIN_TILES", "STY_P_FORCE_H"):
                    os.environ.pop(k, None)
                if "STY_NO_CONV32P_CALL" in env:
                    os.environ["STY_CONV32P_MIN_TILES"] = "1000000000"
                else:
                    os.environ.update(env)
                args = (B, Ci, Co, K, d, T, L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(ws), ws.numel(), bf, st)
                for _ in range(3):
                    L.check(lib.sty_conv1d_fwd(*args))
                torch.cuda.synchronize()
                L.prof_report(256)
                lib.sty_prof_enable(1)
                for _ in range(20):
                    L.check(lib.sty_conv1d_fwd(*args))
                torch.cuda.synchronize()
                lib.sty_prof_enable(0)
                rows = [r for r in L.prof_report(256) if r["name"].startswith("conv")]
                r = max(rows, key=lambda r: r["ms"])
                us = 1e3 * r["ms"] / r["launches"]
                print(f"B{B} k{K} d{d} T{T} {'bf16' if bf else 'fp32'} {tag:28s} {r['name']:36s} {us:8.1f} us "
                      f"{r['flops'] / r['launches'] / us / 1e6:6.1f} TF {r['bytes'] / r['launches'] / us / 1e3:7.0f} GB/s")
                sys.stdout.flush()


if __name__ == "__main__":
    main()
