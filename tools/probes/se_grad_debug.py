"""Where do the style encoder's activation gradients differ from the oracle's?  (debug aid for tests/test_hip_parity.py::
test_mel_style_encoder_block_taps_forward_and_gradient)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import stylish_tts_amd as S  # noqa: E402
from oracle import style_encoder as ose  # noqa: E402
from oracle.manifest import style_encoder_manifest  # noqa: E402
from oracle.weights import fill_state_dict  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 80
P = fill_state_dict(style_encoder_manifest(), 0)
m = S.MelStyleEncoder()
m.load_state_dict(P, strict=False)
m = m.cuda().enable_training()
g = torch.Generator().manual_seed(W)
x = torch.randn(2, 1, 80, W, generator=g) * 0.8 - 0.3
cot = torch.randn(2, 64, generator=g)
out = m.forward_train(x.cuda())
m.backward(cot.cuda())
grads = [m.tap(i, grad=True).cpu() for i in range(6)]
want = {}
Pr = {k: v.clone() for k, v in P.items()}
Pr["shared.0.bias"].requires_grad_(True)
ref = ose.mel_style_encoder(Pr, "", x, want)
names = [f"se.block{i}" for i in range(5)] + ["se.head"]
for k in names:
    want[k].retain_grad()
(ref * cot).sum().backward()
for i, k in enumerate(names[:5]):
    r = want[k].grad
    d = (grads[i] - r).abs()
    scale = r.abs().max().item()
    bad = (d > 1e-4 * scale).nonzero()
    print(f"{k}: shape {tuple(r.shape)} max err {d.max().item() / scale:.2e}, {bad.shape[0]} bad elements")
    if bad.shape[0]:
        print("   b:", sorted(set(bad[:, 0].tolist())), " c: n=", len(set(bad[:, 1].tolist())),
              " h:", sorted(set(bad[:, 2].tolist()))[:20], " w:", sorted(set(bad[:, 3].tolist()))[:20])
        for j in bad[:6].tolist():
            print("   ", j, grads[i][tuple(j)].item(), r[tuple(j)].item())
