"""How far the bf16-mode gradients of the c3 predictor move between DISPATCH configurations of the same kernels (DESIGN.md section 7
item 10): the backward under d<audio, R>, R = sign(fp32 HIP audio) / N, with convp16_kernel taking launches from 256 / 32 / 48 tiles;
max-abs distance of each configuration's gradients from the fp32 HIP run.  Run on the GPU box: python tools/cmp_dispatch_bf16.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import stylish_tts_amd as S
from test_full_size import _inputs, _models, C3_SP_KEYS
from oracle import frontend
DEV = "cuda"
w, inp = _inputs("c3", 4242)
_, _, P, _ = _models()
g = torch.Generator().manual_seed(9)
B = w["B"]
style = torch.randn(B, 64, generator=g)
energy = torch.randn(B, w["T"], generator=g)
ali = frontend.duration_to_alignment(inp["durations"])
voiced = (inp["pitch"] > 20).float()
dev = lambda t: t.to(DEV)
res = {}
cfgs = [("fp32", False, {}), ("p256", True, {"STY_CONVP16_MIN_TILES": "256"}), ("p32", True, {"STY_CONVP16_MIN_TILES": "32"}),
        ("p48", True, {"STY_CONVP16_MIN_TILES": "48"})]
for tag, bf, envs in cfgs:
    for k in ("STY_CONVP16_MIN_TILES",):
        os.environ.pop(k, None)
    os.environ.update(envs)
    m = S.SpeechPredictor()
    m.load_state_dict({k: v.detach() for k, v in P.items()}, strict=False)
    m = m.to(DEV).enable_training().set_train_opts(compute_bf16=bf)
    a = m.forward_train(dev(inp["texts"]), dev(inp["text_lengths"]), dev(ali), dev(inp["pitch"]), dev(energy), dev(voiced),
                        dev(style), dev(inp["pitch"]), noise=dev(inp["noise"]))
    if "R" not in res:
        res["R"] = torch.sign(a.detach()) / (a[0].numel() * B)
    d_style, _ = m.backward(res["R"], want_energy=False)
    torch.cuda.synchronize()
    named = dict(m.named_parameters())
    res[tag] = dict(audio=a.cpu(), d_style=d_style.cpu(), **{k: named[k].grad.cpu().clone() for k in C3_SP_KEYS if k in named})
    del m
def dist(x, y):
    return ((x - y).abs().max() / y.abs().max()).item()
keys = ["audio", "d_style", "generator.basegen.phase_convnext.3.pwconv1.weight", "generator.basegen.amp_convnext.2.pwconv1.weight",
        "generator.amp_conformer.layers.0.ff1.fn.fn.net.0.weight", "text_encoder.proj_m.weight"]
print("max-abs distance to the fp32 HIP run / its scale:", [k[-24:] for k in keys])
for tag, _, _ in cfgs[1:]:
    print("  %-6s" % tag, " ".join("%.3e" % dist(res[tag][k].double(), res["fp32"][k].double()) for k in keys))
