"""Host and GPU time of the c3 step's backward tail: when does the host issue the style encoder's backward, when does the
GPU reach d_style, when does the style-encoder stream actually start?  (wrappers around AcousticTrainer's calls; no
product code is changed)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
from stylish_tts_amd import lib as L
lib = L.load()
w = bench.WORKLOADS["c3"]
model, style_enc, P = bench.build_model(dev)
from stylish_tts_amd.acoustic import AcousticTrainer
inp = bench.make_inputs(w, 1000, dev)
tr = AcousticTrainer(model, style_enc, lr=1e-4, compute=w.get("compute", "fp32"), seed=0)
log = []
sp_bwd, se_bwd, wait_ds = tr.sp.backward, tr.se.backward, tr.sp.wait_d_style
main = torch.cuda.current_stream(dev)

def ev(stream):
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream)
    return e

def sp_backward(*a, **k):
    rec = {"h_sp0": time.perf_counter(), "g_sp0": ev(main)}
    log.append(rec)
    r = sp_bwd(*a, **k)
    rec["h_sp1"] = time.perf_counter()
    rec["g_sp1"] = ev(main)
    return r

def wait_d_style(side):
    rec = log[-1]
    rec["h_wait"] = time.perf_counter()
    wait_ds(side)
    rec["g_side_unblocked"] = ev(side)

def se_backward(d):
    rec = log[-1]
    rec["h_se0"] = time.perf_counter()
    rec["g_se0"] = ev(torch.cuda.current_stream(dev))
    r = se_bwd(d)
    rec["h_se1"] = time.perf_counter()
    rec["g_se1"] = ev(torch.cuda.current_stream(dev))
    return r

tr.sp.backward, tr.se.backward, tr.sp.wait_d_style = sp_backward, se_backward, wait_d_style
N = 8
t_host = []
for i in range(N):
    t_host.append(time.perf_counter())
    tr.train_batch(audio_gt=inp["audio_gt"], texts=inp["texts"], text_lengths=inp["text_lengths"], pitch=inp["pitch"],
                   durations=inp["durations"], seed=i)
torch.cuda.synchronize()
t_end = time.perf_counter()
for i in range(3, N):
    r = log[i]
    g0 = r["g_sp0"]
    print(f"step {i}: host: step start {1e3 * (t_host[i] - r['h_sp0']):8.2f}  sp.backward 0 .. {1e3 * (r['h_sp1'] - r['h_sp0']):6.2f} ms, "
          f"se.backward {1e3 * (r['h_se0'] - r['h_sp0']):6.2f} .. {1e3 * (r['h_se1'] - r['h_sp0']):6.2f} ms  |  "
          f"GPU: main after sp.backward {g0.elapsed_time(r['g_sp1']):6.2f} ms, side unblocked {g0.elapsed_time(r['g_side_unblocked']):6.2f}, "
          f"se bwd {g0.elapsed_time(r['g_se0']):6.2f} .. {g0.elapsed_time(r['g_se1']):6.2f} ms")
print(f"host loop {1e3 * (t_host[-1] - t_host[0]) / (N - 1):.2f} ms/step issue; total {1e3 * (t_end - t_host[0]) / N:.2f} ms/step")
