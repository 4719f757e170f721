"""Worker of test_fft_front_end_equals_the_gemm_front_end: computes the two mel front ends and the acoustic losses with their
seed gradient on one seeded input and saves them.  The parent runs it twice -- default (LDS FFT) and with STY_DFT_GEMM=1 (the
folded-DFT GEMMs) -- and compares the files.  usage: frontend_ab_worker.py OUT.pt B N"""
import sys

import torch


def main():
    out, B, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    from stylish_tts_amd.frontend import MelSpec, calculate_mel
    from stylish_tts_amd.losses import acoustic_loss
    g = torch.Generator().manual_seed(11)
    t = torch.arange(N) / 24000.0
    gt = torch.stack([0.3 * torch.sin(2 * torch.pi * (110.0 * (b + 1)) * t) + 0.05 * torch.randn(N, generator=g)
                      for b in range(B)])
    pred = 0.8 * gt + 0.05 * torch.randn(B, N, generator=g)
    gt, pred = gt.cuda(), pred.cuda()
    res = {}
    for name, spec in (("mel512", MelSpec(512, 512, 300)), ("mel2048", MelSpec(2048, 1200, 300))):
        mel, _, energy = calculate_mel(gt, spec, -4.0, 4.0, want_energy=True)
        res[name], res[name + "_energy"] = mel.cpu(), energy.cpu()
    losses, d = acoustic_loss(gt, pred)
    torch.cuda.synchronize()
    res["losses"], res["d_pred"] = losses.cpu(), d.cpu()
    torch.save(res, out)


if __name__ == "__main__":
    main()
