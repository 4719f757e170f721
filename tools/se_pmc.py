"""The style encoder's training forward + backward at c3's size (B = 32 utterances, 80-band mel of T = 520 frames... the encoder
sees the 2048-point mel: [B, 1, 80, 131]) in the bf16 mode, a few times: workload of tools/cnx_pmc.sh se (SQ counters of
convp16_kernel / wgradb16_kernel / the down-sampling kernels)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stylish_tts_amd as S
from stylish_tts_amd.manifest import style_encoder_manifest
from stylish_tts_amd.synthetic_weights import fill_state_dict
import bench

w = bench.WORKLOADS["c3"]
inp = bench.make_inputs(w, 5, "cuda")
from stylish_tts_amd.acoustic import TO_STYLE_MEL  # noqa: E402
from stylish_tts_amd.frontend import calculate_mel  # noqa: E402
se = S.MelStyleEncoder()
se.load_state_dict(fill_state_dict(style_encoder_manifest(), 0))
se = se.cuda().enable_training().set_train_opts(sn_power_iter=True, compute_bf16=True)
mel = calculate_mel(inp["audio_gt"], TO_STYLE_MEL, -4.0, 4.0)[0].unsqueeze(1)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    out = se.forward_train(mel)
    se.backward(torch.ones_like(out) / out.numel())
torch.cuda.synchronize()
print("ok")
