"""Worker of test_stem_weight_gradient_streaming_kernel_equals_the_tiled_kernel: the style encoder's training-graph forward +
backward on one seeded mel batch, in the fp32 or the bf16-operand mode; saves the stem's weight / bias gradient and the one of
the conv behind it.  The parent runs it with and without STY_NO_STEM_WGRAD=1 (the switch is read once per process) and compares
the files.  usage: stem_wgrad_ab_worker.py OUT.pt B T bf16(0|1)"""
import sys

import torch


def main():
    out, B, T, bf16 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import stylish_tts_amd as S
    from oracle.manifest import style_encoder_manifest
    from oracle.weights import fill_state_dict
    m = S.MelStyleEncoder()
    m.load_state_dict(fill_state_dict(style_encoder_manifest(), 0), strict=False)
    m = m.to("cuda").enable_training()
    if bf16:
        m.set_train_opts(compute_bf16=True)
    g = torch.Generator().manual_seed(T + 7 * B)
    x = torch.randn(B, 1, 80, T, generator=g) * 0.8 - 0.3
    cot = torch.randn(B, 64, generator=g)
    m.forward_train(x.cuda())
    m.backward(cot.cuda())
    torch.cuda.synchronize()
    named = dict(m.named_parameters())
    keys = [k for k in named if named[k].grad is not None]
    stem = "shared.0.weight_orig"  # mel_style_encoder.py: spectral_norm(Conv2d(1, dim_in, 3, 1, 1)), the one conv of a one-channel image
    assert stem in keys and named[stem].shape[1] == 1 and tuple(named[stem].shape[2:]) == (3, 3), named[stem].shape
    res = {"stem_key": stem}
    for k in keys:
        res[k] = named[k].grad.detach().cpu().clone()
    res["stem_w"] = res[stem]
    res["stem_b"] = res["shared.0.bias"]
    torch.save(res, out)


if __name__ == "__main__":
    main()
