"""How well-conditioned are the parameter gradients of one train_acoustic step at c3's shape (T = 520, L = 100)?

For a chunk of B utterances of the c3 inputs: the fp32 oracle's autograd gradients, the float64 oracle's, and the HIP
path's (fp32), pairwise, for the keys test_c3_train_step_full_size_vs_oracle gates.  If the fp32 oracle sits as far from
float64 as the HIP path does, the distance is the graph's fp32 conditioning and not a kernel.  Run on the GPU box:
  python tools/probes/c3_grad_conditioning.py [B]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main(B):
    from oracle import losses as ol, speech_predictor as osp
    from stylish_tts_amd.acoustic import AcousticTrainer
    from tests import test_full_size as F
    w, inp = F._inputs("c3", 2024)
    inp = {k: v[:B].contiguous() for k, v in inp.items()}
    sp, se, P, Pse = F._models()
    sp_keys = ["generator.basegen.amp_output_conv.weight", "generator.basegen.phase_convnext.3.pwconv1.weight",
               "generator.basegen.amp_convnext.2.pwconv1.weight", "generator.basegen.amp_prior_block.convs2.1.bias",
               "generator.basegen.amp_prior_block.convs1.1.parametrizations.weight.original1",
               "generator.amp_conformer.layers.0.ff1.fn.fn.net.0.weight",
               "decoder.decode.0.norm1.fc.weight", "decoder.decode.1.conv1.parametrizations.weight.original1",
               "text_encoder.encoder.ffn_layers.3.conv_1.weight", "text_encoder.encoder.ffn_layers.7.conv_2.weight",
               "text_encoder.encoder.attn_layers.0.conv_q.weight", "text_encoder.proj_m.weight", "text_encoder.emb.weight"]
    sp_keys = [k for k in sp_keys if k in P and P[k].is_floating_point()]
    se_keys = ["shared.0.weight_orig", "shared.2.conv1.weight_orig", "shared.4.conv2.weight_orig", "unshared.weight"]

    def oracle(dtype):
        Pd = {k: (v.to(dtype) if v.is_floating_point() else v).clone() for k, v in P.items()}
        Ps = {k: (v.to(dtype) if v.is_floating_point() else v).clone() for k, v in Pse.items()}
        for k in sp_keys:
            Pd[k].requires_grad_(True)
        for k in se_keys:
            Ps[k].requires_grad_(True)
        c = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in inp.items()}
        want = {}
        t0 = time.perf_counter()
        ref = osp.acoustic_forward(Pd, Ps, c["audio_gt"], c["texts"], c["text_lengths"], c["pitch"], c["durations"],
                                   c["noise"], want)
        mel, mph, tot = ol.acoustic_losses(c["audio_gt"], ref.squeeze(1))
        tot.backward()
        print(f"oracle {dtype}: {time.perf_counter() - t0:.1f} s, mel {mel.item():.6f} multi_phase {mph.item():.6f}", flush=True)
        g = {"sp." + k: Pd[k].grad.double() for k in sp_keys}
        g.update({"se." + k: Ps[k].grad.double() for k in se_keys})
        return g, want["prior"].float(), ref.detach()

    g32, prior, _ = oracle(torch.float32)
    g64, _, _ = oracle(torch.float64)
    d = lambda t: t.to("cuda:0")
    kw = dict(audio_gt=d(inp["audio_gt"]), texts=d(inp["texts"]), text_lengths=d(inp["text_lengths"]), pitch=d(inp["pitch"]),
              durations=d(inp["durations"]), noise=d(inp["noise"]), prior_override=d(prior))
    runs = {}
    for tag, single in (("hip", 0), ("hip-single-stream", 1)):
        from stylish_tts_amd import lib as L
        L.load().sty_set_single_stream(single)
        spx, sex, _, _ = F._models()
        tr = AcousticTrainer(spx, sex, lr=0.0, train_mode=False)
        tr.single_stream = bool(single)
        losses = tr.train_batch(**kw)
        torch.cuda.synchronize()
        print(f"{tag}: losses {losses.tolist()}")
        nsp, nse = dict(tr.sp.named_parameters()), dict(tr.se.named_parameters())
        g = {"sp." + k: nsp[k].grad.detach().cpu().double() for k in sp_keys}
        g.update({"se." + k: nse[k].grad.detach().cpu().double() for k in se_keys})
        runs[tag] = g
        L.load().sty_set_single_stream(0)

    def cmp(a, b):
        e = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
        cos = torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
        return f"{e:9.2e} {1 - cos:9.2e}"

    print(f"\nB = {B}: rel max err / (1 - cosine)")
    print(f"{'key':58s} {'oracle32 vs 64':>20s} {'hip vs 64':>20s} {'hip vs oracle32':>20s} {'hip vs hip-1stream':>20s}")
    for k in g64:
        print(f"{k[-58:]:58s} {cmp(g32[k], g64[k]):>20s} {cmp(runs['hip'][k], g64[k]):>20s} "
              f"{cmp(runs['hip'][k], g32[k]):>20s} {cmp(runs['hip'][k], runs['hip-single-stream'][k]):>20s}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
