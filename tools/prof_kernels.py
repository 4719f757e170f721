"""Run the 75T-rate building blocks at config-c5 size a few times (for rocprofv3 --pmc / --kernel-trace)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stylish_tts_amd as S
from stylish_tts_amd import lib as L
from oracle.manifest import speech_predictor_manifest
from oracle.weights import fill_state_dict

B, T = 8, 60000
m = S.SpeechPredictor()
m.load_state_dict(fill_state_dict(speech_predictor_manifest(), 0), strict=False)
m = m.cuda()
m._ensure(torch.device("cuda:0"))
lib = L.load()
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 32, T, generator=g).cuda()
style = torch.randn(B, 64, generator=g).cuda()
y = torch.empty_like(x)
ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    L.check(lib.sty_resblock_fwd(m._handle, b"generator.basegen.amp_prior_block", B, T, L.ptr(x), L.ptr(style), L.ptr(y),
                                 L.ptr(ws), ws.numel(), st))
    L.check(lib.sty_convnext_fwd(m._handle, b"generator.basegen.phase_convnext.0", B, 32, T, L.ptr(x), L.ptr(style),
                                 L.ptr(y), L.ptr(ws), ws.numel(), st))
torch.cuda.synchronize()
print("ok")
