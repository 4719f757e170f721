"""Golden fixtures of the second and third training stages by RUNNING THE REFERENCE'S OWN STAGE FUNCTIONS (build
container only):

    python tools/gen_golden_stages.py

`train_duration` (train/stage_type.py:495-556) and `train_textual` (stage_type.py:415-450, through `AcousticStep`,
stage_type.py:61-262 with use_predicted_pe=True) are called as the reference's trainer calls them, on a stand-in `train`
object that carries what the two functions read: the reference's own model classes filled by the key-named generator of
oracle/weights.py, the reference's DurationProcessor, DurationLoss, GeneratorLoss (with its PitchDiscriminators, weights
from tests/golden/pdisc_small), MultiResolutionSTFTLoss, LossLog and the loss weights of config/config.yml.  What is
NOT the reference's: torchaudio is absent from this image (and from uv.lock), so the three objects that wrap it --
`train.to_mel`, `train.to_style_mel` and the MelScale inside MultiSpectrogram -- are the restatement in oracle/frontend.py
(the same caveat as row A1 of SURVEY.md 8: parity unpinned at the torchaudio boundary).  Everything downstream of the mel
(style encoders, predictors, alignment, every loss term, LossLog.backwards_loss and the autograd backward) is reference code.

Writes tests/golden/stages_small.safetensors: the logged loss values and the gradients the reference's backward left on a
list of parameters of the trained models.  Inputs are regenerated from seeds by the test (tests/cases.py).
The SpeechPredictor runs in eval mode here although StageType puts it in eval_models anyway (stage.py:451-468); the
trained models run in .eval() too (dropouts / DropPath off): the training-mode dropouts are pinned separately
(tools/gen_golden_dropout.py).  Only data is written.
"""
import os
import sys
import types

import torch
from safetensors.torch import load_file, save_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import  # noqa: E402

OUT = os.environ.get("STY_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))
G = os.path.join(ROOT, "tests", "golden")

DP_KEYS = ["cross_attention.conv_q.weight", "cross_post.0.parametrizations.weight.original1", "conv_next.1.pwconv1.weight",
           "duration_proj.linear_layer.weight", "query_norm.fc.weight", "text_encoder.proj_m.weight"]
DSE_KEYS = ["shared.0.weight_orig", "shared.2.conv1.weight_orig", "unshared.weight"]
PEP_KEYS = ["prosody_encoder.attn_layers.0.conv_q.weight", "prosody_encoder.proj_layers.1.weight",
            "F0.0.conv1.parametrizations.weight.original1", "N.3.conv2.parametrizations.weight.original1",
            "F0_proj.weight", "N_proj.weight", "text_encoder.proj_m.weight"]
PSE_KEYS = ["preconv.parametrizations.weight.original1", "shared.2.conv1.weight_orig", "unshared.weight"]


def test_audio(B, n, seed):
    """the audio of tests/test_hip_parity.py::_test_audio (kept in step with it by the test that reads this fixture)"""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n) / 24000.0
    f0 = 110.0 + 60.0 * torch.rand(B, 1, generator=g)
    x = sum(torch.sin(2 * torch.pi * f0 * (h + 1) * t) / (h + 1) for h in range(6)) * 0.2
    return x + 0.01 * torch.randn(B, n, generator=g)


def sub(t, n=4096):
    """large gradients are stored as a strided sample of n elements + their L2 norm (fixtures stay small)"""
    f = t.detach().flatten()
    return f if f.numel() <= n else f[::f.numel() // n][:n].clone()


class _Mel:
    """stands in for torchaudio.transforms.MelSpectrogram (absent here): power mel through oracle/frontend.py"""

    def __init__(self, n_fft, win, hop):
        self.a = (n_fft, win, hop)

    def __call__(self, audio):
        from oracle import frontend
        return frontend.mel_spectrogram(audio, *self.a)


def main():
    mc = ref_import.model_config()
    for name in ("soundfile", "librosa", "librosa.filters", "tqdm", "tensorboard", "torch.utils.tensorboard",
                 "torch.utils.tensorboard.writer", "k2"):
        if name not in sys.modules:
            ref_import._stub(name, tqdm=lambda x, *a, **k: x, mel=None, SummaryWriter=object)
    # loss_log.py / stage_type.py import train_context for type annotations only
    ref_import._stub("stylish_tts.train.train_context", TrainContext=object)
    from oracle import frontend
    import torchaudio

    class MelScale(torch.nn.Module):  # multi_spectrogram.py:32 -- torchaudio.transforms.MelScale(n_mels, sr, n_stft=...)
        def __init__(self, n_mels=128, sample_rate=24000, f_min=0.0, f_max=None, n_stft=201, norm=None, mel_scale="htk"):
            super().__init__()
            self.register_buffer("fb", frontend.mel_filterbank(n_stft, n_mels, sample_rate))

        def forward(self, spec):
            return torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)

    torchaudio.transforms.MelScale = MelScale
    from stylish_tts.lib.config_loader import load_config_yaml
    from stylish_tts.train import stage_type as ST
    from stylish_tts.train.losses import DurationLoss, GeneratorLoss, MultiResolutionSTFTLoss
    from stylish_tts.train.models.duration_predictor import DurationPredictor
    from stylish_tts.train.models.mel_style_encoder import MelStyleEncoder, PitchStyleEncoder
    from stylish_tts.train.models.pitch_discriminator import PitchDiscriminator
    from stylish_tts.train.models.pitch_energy_predictor import PitchEnergyPredictor
    from stylish_tts.train.models.speech_predictor import SpeechPredictor
    from stylish_tts.train.multi_spectrogram import MultiSpectrogram
    from stylish_tts.train.utils import DurationProcessor
    from oracle.manifest import (duration_predictor_manifest, pitch_energy_predictor_manifest,
                                 pitch_style_encoder_manifest, speech_predictor_manifest, style_encoder_manifest)
    from oracle.weights import fill_state_dict
    from tests.cases import make_case

    torch.set_num_threads(8)
    cfg = load_config_yaml("/root/reference/config/config.yml")
    cs = make_case("sp_small")
    B, T = cs["pitch"].shape
    audio_gt = test_audio(B, 300 * T, 21)
    fx = load_file(os.path.join(G, "pdisc_small.safetensors"))

    def filled(mod, manifest, seed, strict=True):
        miss, unexp = mod.load_state_dict(fill_state_dict(manifest, seed), strict=False)
        assert not unexp and (not strict and all(".stft." in k for k in miss) or not miss), (miss, unexp)
        return mod.eval()

    se_args = (mc.style_encoder.n_mels, mc.style_dim, mc.style_encoder.max_channels, mc.style_encoder.skip_downsample)
    model = ref_import._Munch(
        duration_style_encoder=filled(MelStyleEncoder(*se_args), style_encoder_manifest(), 7),
        duration_predictor=filled(DurationPredictor(style_dim=mc.style_dim, inter_dim=mc.inter_dim,
                                                    text_config=mc.text_encoder, duration_config=mc.duration_predictor),
                                  duration_predictor_manifest(), 3),
        speech_style_encoder=filled(MelStyleEncoder(*se_args), style_encoder_manifest(), 0),
        speech_predictor=filled(SpeechPredictor(mc), speech_predictor_manifest(), 0, strict=False),
        pe_style_encoder=filled(PitchStyleEncoder(*se_args, coarse_multiplier=mc.coarse_multiplier),
                                pitch_style_encoder_manifest(), 5),
        pitch_energy_predictor=filled(PitchEnergyPredictor(style_dim=mc.style_dim, inter_dim=mc.pitch_energy_predictor.inter_dim,
                                                           text_config=mc.text_encoder, duration_config=mc.duration_predictor,
                                                           pitch_energy_config=mc.pitch_energy_predictor),
                                      pitch_energy_predictor_manifest(), 4))
    pitch_disc, dur_disc = PitchDiscriminator(dim_in=2, dim_hidden=64, kernel=21), PitchDiscriminator(dim_in=1, dim_hidden=64, kernel=5)
    pitch_disc.load_state_dict({k[len("pitch.w."):]: v for k, v in fx.items() if k.startswith("pitch.w.")})
    dur_disc.load_state_dict({k[len("dur.w."):]: v for k, v in fx.items() if k.startswith("dur.w.")})
    for p in list(model.speech_predictor.parameters()) + list(model.speech_style_encoder.parameters()):
        p.requires_grad_(False)  # eval_models of the textual stage; the stage never steps them

    class Opt:
        def zero_grad(self):
            for m in model.values():
                m.zero_grad()

    class Acc:
        def backward(self, loss):
            loss.backward()

    weights = torch.linspace(0.5, 2.0, 16)  # DurationLoss class weights (train.py:190-199 derives them from the data)
    train = types.SimpleNamespace(
        model=model, model_config=mc, config=cfg, logger=None, writer=None,
        to_mel=_Mel(mc.n_fft, mc.win_length, mc.hop_length),
        to_style_mel=_Mel(mc.style_encoder.n_fft, mc.style_encoder.win_length, mc.style_encoder.hop_length),
        normalization=types.SimpleNamespace(mel_log_mean=-4.0, mel_log_std=4.0),
        duration_processor=DurationProcessor(mc.duration_predictor.duration_classes, mc.duration_predictor.max_duration),
        duration_loss=DurationLoss(class_count=mc.duration_predictor.duration_classes, weight=weights),
        generator_loss=GeneratorLoss(mrd0=None, mrd1=None, mrd2=None, disc=None, pitch=pitch_disc, duration=dur_disc),
        multi_spectrogram=MultiSpectrogram(sample_rate=mc.sample_rate),
        stft_loss=MultiResolutionSTFTLoss(sample_rate=mc.sample_rate), stage=types.SimpleNamespace(optimizer=Opt()), accelerator=Acc())
    batch = types.SimpleNamespace(audio_gt=audio_gt, text=cs["texts"], text_length=cs["text_lengths"], pitch=cs["pitch"],
                                  alignment=cs["durations"].unsqueeze(1))
    out = {"class_weights": weights, "audio_gt": audio_gt}

    # ---- third stage ----
    log, tgt, pred, _, _ = ST.train_duration(batch, model, train, False, 0)
    for k, v in log.metrics.items():
        out["duration.log." + k] = torch.as_tensor(float(v)).reshape(1)
    out["duration.pred_duration"] = pred[0].squeeze(1).contiguous()
    for k in DP_KEYS:
        g_ = dict(model.duration_predictor.named_parameters())[k].grad
        out["duration.grad.dp." + k], out["duration.norm.dp." + k] = sub(g_), g_.norm().reshape(1)
    for k in DSE_KEYS:
        g_ = dict(model.duration_style_encoder.named_parameters())[k].grad
        out["duration.grad.se." + k], out["duration.norm.se." + k] = sub(g_), g_.norm().reshape(1)
    print("train_duration:", {k: round(float(v), 6) for k, v in log.metrics.items()})

    # ---- second stage ----
    # SineGen draws its noise from the global generator inside the predictor's forward (generator.py:440-442); the test
    # feeds the same draw through the explicit `noise` input
    torch.manual_seed(cs["noise_seed"])
    log, _, predcat, _, _ = ST.train_textual(batch, model, train, False, 0)
    for k, v in log.metrics.items():
        out["textual.log." + k] = torch.as_tensor(float(v)).reshape(1)
    out["textual.pred_pitchcat"] = predcat[0].contiguous()
    for k in PEP_KEYS:
        g_ = dict(model.pitch_energy_predictor.named_parameters())[k].grad
        out["textual.grad.pep." + k], out["textual.norm.pep." + k] = sub(g_), g_.norm().reshape(1)
    for k in PSE_KEYS:
        g_ = dict(model.pe_style_encoder.named_parameters())[k].grad
        out["textual.grad.pse." + k], out["textual.norm.pse." + k] = sub(g_), g_.norm().reshape(1)
    print("train_textual:", {k: round(float(v), 6) for k, v in log.metrics.items()})
    save_file({k: v.contiguous().float() for k, v in out.items()}, os.path.join(OUT, "stages_small.safetensors"),
              metadata={"audio_seed": "21", "case": "sp_small", "weights": "duration 3/7, textual 4/5, predictor 0/0",
                        "mel": "oracle/frontend.py stands in for torchaudio (absent)"})
    print("size KB", os.path.getsize(os.path.join(OUT, "stages_small.safetensors")) // 1024)


if __name__ == "__main__":
    main()
