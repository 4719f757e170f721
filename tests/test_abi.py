"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/stylish_hip.h declares,
and the nn.Module shells reproduce the reference's state_dict layout.  No compute calls (no GPU here)."""
import json
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from stylish_tts_amd import lib as L
    return L.load()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "stylish_hip.h")).read()
    declared = set(re.findall(r"\b(sty_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"sty_status"}
    from stylish_tts_amd import lib as L
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sty_version() >= 1


def test_every_exported_symbol_is_declared(lib):
    """the other direction: nothing `sty_*` leaves the library without a declaration in include/stylish_hip.h (round 4:
    sty_stft64_bases_host was exported and undeclared)"""
    import subprocess
    from stylish_tts_amd import build as sbuild, lib as L
    out = subprocess.run(["nm", "-D", "--defined-only", sbuild.LIB], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and re.fullmatch(r"sty_[a-z0-9_]+", ln.split()[-1])}
    assert exported == set(L.SYMBOLS), exported ^ set(L.SYMBOLS)


def test_default_stft_bases_are_the_reference_buffers(lib):
    """sty_stft64_bases_host (host only: no device needed) against the buffers the reference's STFT module registers
    (tests/golden/stft_buffers.safetensors, dumped by tools/gen_golden.py): within 2 ulp of fp32 (the library's host
    builder rounds a double-precision product once; the module shells bind the reference's buffers bit for bit, see
    test_module_shells_have_reference_state_dict_layout -- these defaults serve the stand-alone entry points)"""
    import ctypes as C
    from safetensors.torch import load_file
    bufs = load_file(os.path.join(ROOT, "tests", "golden", "stft_buffers.safetensors"))
    out = (C.c_float * (4 * 33 * 64))()
    lib.sty_stft64_bases_host(out)
    got = torch.tensor(list(out)).view(4, 33, 64)
    for i, k in enumerate(("weight_forward_real", "weight_forward_imag", "weight_backward_real", "weight_backward_imag")):
        ref = bufs[k][:, 0, :]
        assert (got[i] - ref).abs().max().item() <= 2.4e-7 * ref.abs().max().item(), k


def test_error_paths_without_gpu(lib):
    import ctypes as C
    h = C.c_void_p()
    assert lib.sty_model_create(b"no_such_kind", C.byref(h)) == -1
    assert b"unknown model kind" in lib.sty_last_error()
    assert lib.sty_model_create(b"speech_predictor", C.byref(h)) == 0
    # forward before finalize -> STY_ESTATE, never a silent fallback
    need = C.c_size_t()
    assert lib.sty_vocoder_workspace_bytes(h, 2, 80, C.byref(need)) == -5
    lib.sty_model_destroy(h)


def test_module_shells_have_reference_state_dict_layout():
    import stylish_tts_amd as S
    from safetensors.torch import load_file
    G = os.path.join(ROOT, "tests", "golden")
    sp = S.SpeechPredictor()
    ref = json.load(open(os.path.join(G, "manifest_speech_predictor.json")))
    assert {k: list(v.shape) for k, v in sp.state_dict().items()} == ref
    se = S.MelStyleEncoder()
    ref = json.load(open(os.path.join(G, "manifest_style_encoder.json")))
    assert {k: list(v.shape) for k, v in se.state_dict().items()} == ref
    # STFT bases registered by the shell are bit-identical to the reference's buffers
    bufs = load_file(os.path.join(G, "stft_buffers.safetensors"))
    for k, v in bufs.items():
        assert torch.equal(sp.state_dict()["generator.basegen.stft." + k], v), k


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "stylish_tts_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "/root/reference" not in src, f


def test_forward_refuses_autograd_and_cpu():
    import stylish_tts_amd as S
    sp = S.SpeechPredictor()
    x = torch.zeros(1, 128, 8)
    with pytest.raises(S.StyError):
        sp.vocoder_forward(mel=x, style=torch.zeros(1, 64), pitch=torch.zeros(1, 8), voiced=torch.zeros(1, 8))


def test_lr_schedule_matches_transformers_cosine():
    """optim.scheduled_lr vs the reference's scheduler: transformers.get_cosine_schedule_with_warmup(opt, 0, 10000) with
    last_epoch set to the logical step (train/optimizers.py:96-104, 119-123)."""
    import torch
    import transformers
    from stylish_tts_amd.optim import scheduled_lr
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=3e-4)
    sch = transformers.get_cosine_schedule_with_warmup(opt, num_warmup_steps=0, num_training_steps=10000)
    step_limit = 777
    for step in (0, 1, 100, 388, 700, 776, 777, 5000):
        logical = min(step * 10000 // step_limit, 10000 * 0.9)
        sch.last_epoch = logical
        sch.step()  # as MultiOptimizer.scheduler does: sets last_epoch, then steps
        want = opt.param_groups[0]["lr"]
        assert abs(scheduled_lr(3e-4, step, step_limit) - want) <= 1e-12, (step, want)
