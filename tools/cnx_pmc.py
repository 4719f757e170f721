"""One ConvNeXt32 block and one resblock of the vocoder, training graph, bf16 mode, at c3's 75T-rate size (B = 8 utterances of
T = 39 000 = a quarter of the batch), a few times: the workload of tools/cnx_pmc.sh (rocprofv3 --pmc / --kernel-trace)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stylish_tts_amd as S
from stylish_tts_amd.manifest import speech_predictor_manifest
from stylish_tts_amd.synthetic_weights import fill_state_dict

B, T = 8, 39000
m = S.SpeechPredictor()
m.load_state_dict(fill_state_dict(speech_predictor_manifest(), 0), strict=False)
m = m.cuda().enable_training()
m._ensure(torch.device("cuda:0"))
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 32, T, generator=g).cuda()
style = torch.randn(B, 64, generator=g).cuda()
gy = torch.randn(B, 32, T, generator=g).cuda()
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    m.block_forward_backward("convnext", "generator.basegen.phase_convnext.3", x, style, gy, compute_bf16=True)
    m.block_forward_backward("resblock", "generator.basegen.amp_prior_block", x, style, gy, compute_bf16=True)
torch.cuda.synchronize()
print("ok")
