import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench
from stylish_tts_amd.acoustic import AcousticTrainer
dev = torch.device("cuda:0")
model, style_enc, P = bench.build_model(dev)
w = bench.WORKLOADS[os.environ.get("WL", "c2")]
inp = bench.make_inputs(w, 1000, dev)
tr = AcousticTrainer(model, style_enc, lr=1e-4, compute=w.get("compute", "fp32"))
def step(i):
    return tr.train_batch(audio_gt=inp["audio_gt"], texts=inp["texts"], text_lengths=inp["text_lengths"], pitch=inp["pitch"], durations=inp["durations"], seed=i)
for i in range(3): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter(); cpu = []
for i in range(10):
    a = time.perf_counter(); step(3 + i); cpu.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("cpu issue per step ms", [round(1e3 * c, 1) for c in cpu])
print("issue total", round(1e3 * (t1 - t0), 1), "ms; after sync", round(1e3 * (t2 - t0), 1), "ms; per step", round(1e2 * (t2 - t0), 2))
