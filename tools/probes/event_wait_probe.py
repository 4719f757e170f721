"""Does a stream that waits on an event recorded in the MIDDLE of another stream's work start when that point is reached,
or only when the other stream's later work has drained?  A: K1, record E, K2 (many small kernels); B: wait E, K3."""
import os, sys, time
import torch

def spin(x, n):
    for _ in range(n):
        x = x * 1.0001 + 0.1
    return x

dev = torch.device("cuda:0")
a = torch.randn(64 << 20, device=dev)
b = torch.randn(64 << 20, device=dev)
small = torch.randn(1024, device=dev)
A = torch.cuda.current_stream()
B = torch.cuda.Stream()
for flavour in ("torch_event", "torch_event_timing"):
    for rep in range(3):
        torch.cuda.synchronize()
        t = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        t[0].record(A)
        x = spin(a, 20)           # K1: ~ms of big kernels on A
        t[1].record(A)
        E = torch.cuda.Event(enable_timing=(flavour == "torch_event_timing"))
        E.record(A)
        for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):     # K2: a chain of tiny kernels on A
            small = small * 1.0001
        t[2].record(A)
        with torch.cuda.stream(B):
            B.wait_event(E)
            t[3].record(B)
            y = spin(b, 5)
            t[4].record(B)
        torch.cuda.synchronize()
        print(f"{flavour}: K1 ends {t[0].elapsed_time(t[1]):.2f} ms, K2 ends {t[0].elapsed_time(t[2]):.2f} ms, "
              f"B starts {t[0].elapsed_time(t[3]):.2f} ms, B ends {t[0].elapsed_time(t[4]):.2f} ms")
