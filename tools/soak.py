"""Stability check: N training steps of the c2 workload on fresh random batches, losses must stay finite and the mel loss
must go down; every 50 steps the gradients of a repeated reference batch are compared with the single-stream result.

    python tools/soak.py [steps]      # on the GPU box
"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from stylish_tts_amd import lib as L  # noqa: E402
from stylish_tts_amd.acoustic import AcousticTrainer  # noqa: E402


def grads(tr):
    return torch.cat([p.grad.detach().flatten() for m in (tr.sp, tr.se) for p in m.parameters() if p.grad is not None])


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    lib = L.load()
    model, se, _ = bench.build_model(dev)
    w = bench.WORKLOADS["c2"]
    tr = AcousticTrainer(model, se, lr=1e-4, train_mode=False)  # eval-mode graph: deterministic, so runs are comparable
    ref_inp = bench.make_inputs(w, 4242, dev)
    hist, worst = [], 0.0
    for i in range(steps):
        inp = bench.make_inputs(w, 5000 + i, dev)
        losses = tr.train_batch(audio_gt=inp["audio_gt"], texts=inp["texts"], text_lengths=inp["text_lengths"],
                                pitch=inp["pitch"], durations=inp["durations"], seed=i)
        if i % 10 == 0:
            hist.append(losses.cpu().tolist())
            assert all(map(lambda v: v == v and abs(v) < 1e6, hist[-1])), hist[-1]
        if i % 50 == 49:
            # same batch from the same parameters: multi-stream gradients vs single-stream gradients
            g2 = []
            for single in (0, 1):
                lib.sty_set_single_stream(single)
                tr.single_stream = bool(single)
                state = [p.detach().clone() for m in (tr.sp, tr.se) for p in m.parameters()]
                tr.train_batch(audio_gt=ref_inp["audio_gt"], texts=ref_inp["texts"], text_lengths=ref_inp["text_lengths"],
                               pitch=ref_inp["pitch"], durations=ref_inp["durations"], seed=7)
                torch.cuda.synchronize()
                g2.append(grads(tr).clone())
                with torch.no_grad():  # undo the optimizer step
                    for p, s in zip([p for m in (tr.sp, tr.se) for p in m.parameters()], state):
                        p.copy_(s)
            lib.sty_set_single_stream(0)
            tr.single_stream = False
            rel = ((g2[0] - g2[1]).norm() / g2[1].norm()).item()
            worst = max(worst, rel)
            print(f"step {i + 1}: losses {hist[-1]}  multi-stream vs single-stream gradients: relative L2 {rel:.3e}")
    print("first", hist[0], "last", hist[-1], "worst stream discrepancy", worst)
    assert hist[-1][0] < hist[0][0] and worst < 1e-3


if __name__ == "__main__":
    main()
