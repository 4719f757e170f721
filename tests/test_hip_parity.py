"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference-generated golden vectors.

Runs on the GPU box only (`-m gpu`).  Tolerances: fp32, block level 1e-5 of the tensor's scale,
end-to-end audio: the north-star gates (waveform MSE <= 1e-8, mel-L1 <= 1e-3) plus a max-abs bound.
"""
import ctypes as C
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def rel_err(a, b):
    """max |a-b| relative to the scale of the reference tensor b."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-6)


class Report:
    def __init__(self):
        self.rows, self.bad = [], []

    def add(self, name, got, ref, tol):
        e = rel_err(got, ref)
        ok = e <= tol and bool(torch.isfinite(got).all())
        self.rows.append(f"  {name:32s} rel_err {e:9.3e}  tol {tol:.1e}  {'ok' if ok else 'FAIL'}")
        if not ok:
            self.bad.append(name)

    def done(self):
        print("\n" + "\n".join(self.rows))
        assert not self.bad, f"parity failures: {self.bad}"


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    import stylish_tts_amd as S
    from oracle import frontend, speech_predictor as osp
    from oracle.manifest import speech_predictor_manifest
    from oracle.weights import fill_state_dict
    from tests.cases import make_case
    P = fill_state_dict(speech_predictor_manifest(), 0)
    cs = make_case("sp_small")
    ali = frontend.duration_to_alignment(cs["durations"])
    voiced = (cs["pitch"] > 20).float()
    want = {}
    with torch.no_grad():
        ref_audio = osp.speech_predictor(P, cs["texts"], cs["text_lengths"], ali, cs["pitch"], cs["energy"], voiced,
                                         cs["style"], cs["pitch"], cs["noise"], want)
    m = S.SpeechPredictor()
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not unexpected and all(".stft." in k for k in missing)
    m = m.to(DEV)
    m._ensure(torch.device(DEV))
    return dict(S=S, m=m, P=P, cs=cs, ali=ali, voiced=voiced, want=want, ref_audio=ref_audio)


def dev(t):
    return t.to(DEV)


def test_library_is_loaded_and_keys_match_manifest(env):
    from oracle.manifest import speech_predictor_manifest
    req = set(env["m"].requested_keys())
    man = {k for k in speech_predictor_manifest() if "num_batches_tracked" not in k and not k.endswith("stft.window")}
    assert req == man, (sorted(man - req)[:5], sorted(req - man)[:5])
    maps = open("/proc/self/maps").read()
    assert "libstylish_hip.so" in maps


def test_stft64_and_istft64(env):
    from oracle.stft import stft_bases, stft_inverse, stft_transform
    from stylish_tts_amd import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(5)
    wave = torch.randn(3, 4800, generator=g) * 0.3
    bases = stft_bases(64)
    mag, x, y = stft_transform(wave, bases)
    spec_ref, ph_ref = mag[:, :32, :-1], torch.atan2(y, x)[:, :32, :-1]
    w = dev(wave)
    spec = torch.empty(3, 32, 1200, device=DEV)
    ph = torch.empty_like(spec)
    L.check(lib.sty_stft64_fwd(3, 4800, L.ptr(w), L.ptr(spec), L.ptr(ph), None))
    rep = Report()
    rep.add("stft64.spec", spec, spec_ref, 1e-5)
    # phase: compare on the unit circle (atan2 wraps) and only where the bin has energy
    rep.add("stft64.cos(phase)", torch.cos(ph), torch.cos(ph_ref), 2e-4)
    rep.add("stft64.sin(phase)", torch.sin(ph), torch.sin(ph_ref), 2e-4)
    # synthesis head
    F = 1200
    logamp = torch.randn(3, 32, F, generator=g) * 0.5 - 1.0
    real, imag = torch.randn(3, 32, F, generator=g), torch.randn(3, 32, F, generator=g)
    phase = torch.atan2(imag, real)
    la, php = torch.nn.functional.pad(logamp, (0, 1), mode="replicate"), torch.nn.functional.pad(phase, (0, 1), mode="replicate")
    z = torch.zeros_like(la[:, :1])
    ref = torch.tanh(stft_inverse(torch.cat([la.exp(), z], 1), torch.cat([php.cos(), z + 1], 1) * 1.0,
                                  torch.cat([php.sin(), z], 1), bases))
    audio = torch.empty(3, 1, 4 * F, device=DEV)
    d_la, d_re, d_im = dev(logamp), dev(real), dev(imag)  # keep alive: temporaries would be recycled mid-call
    L.check(lib.sty_istft64_fwd(3, F, L.ptr(d_la), L.ptr(d_re), L.ptr(d_im), L.ptr(audio), None))
    rep.add("istft64.audio", audio, ref, 1e-5)
    rep.done()


def test_harmonic_source(env):
    from oracle.vocoder import harmonic_source
    from stylish_tts_amd import lib as L
    lib = L.load()
    cs, P = env["cs"], env["P"]
    want = {}
    with torch.no_grad():
        ref = harmonic_source(P, "generator.basegen.m_source", cs["pitch"], env["voiced"], cs["noise"], want)
    B, T = cs["pitch"].shape
    need = C.c_size_t()
    L.check(lib.sty_source_workspace_bytes(B, T, C.byref(need)))
    ws = torch.empty(need.value, dtype=torch.uint8, device=DEV)
    out = torch.empty(B, 300 * T, device=DEV)
    lw, lb = dev(P["generator.basegen.m_source.l_linear.weight"]), dev(P["generator.basegen.m_source.l_linear.bias"])
    d_p, d_v, d_n = dev(cs["pitch"]), dev(env["voiced"]), dev(cs["noise"])
    L.check(lib.sty_source_fwd(B, T, L.ptr(d_p), L.ptr(d_v), L.ptr(d_n), 0,
                               L.ptr(lw), L.ptr(lb), L.ptr(out), L.ptr(ws), ws.numel(), None))
    err = (out.cpu() - ref).abs()
    print(f"\n  source: max|err| {err.max().item():.3e}  mean|err| {err.mean().item():.3e}  "
          f"frac>1e-5 {(err > 1e-5).float().mean().item():.4f}")
    # conditioning: the phase is ~1e5 rad in fp32 (ulp ~0.008 rad); one ulp moves the output by <= ~1e-3
    assert err.max().item() <= 2e-3 and err.mean().item() <= 2e-5
    # internal RNG path: finite, bounded, unvoiced regions look like noise
    out2 = torch.empty_like(out)
    L.check(lib.sty_source_fwd(B, T, L.ptr(d_p), L.ptr(d_v), None, 7, L.ptr(lw), L.ptr(lb),
                               L.ptr(out2), L.ptr(ws), ws.numel(), None))
    assert bool(torch.isfinite(out2).all()) and out2.abs().max().item() <= 1.0 and out2.std().item() > 1e-3


def _block_params(P, prefix):
    return {k: v for k, v in P.items() if k.startswith(prefix + ".")}


@pytest.mark.parametrize("prefix,C,T", [
    ("generator.basegen.phase_convnext.0", 32, 600),
    ("generator.basegen.phase_convnext.5", 32, 1000),
    ("generator.basegen.upblocks.2", 32, 517),
    ("generator.basegen.upblocks.1", 64, 300),
    ("generator.basegen.upblocks.0", 128, 240),
    ("generator.basegen.amp_convnext.2", 256, 80),
])
def test_convnext_block(env, prefix, C, T):
    from oracle import blocks
    from stylish_tts_amd import lib as L
    lib, m, P = L.load(), env["m"], env["P"]
    g = torch.Generator().manual_seed(C + T)
    x = torch.randn(2, C, T, generator=g)
    style = torch.randn(2, 64, generator=g)
    with torch.no_grad():
        ref = blocks.convnext_block(P, prefix, x, style)
    y = torch.empty(2, C, T, device=DEV)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    st = C_void(torch.cuda.current_stream().cuda_stream)
    d_x, d_s = dev(x), dev(style)
    L.check(lib.sty_convnext_fwd(m._handle, prefix.encode(), 2, C, T, L.ptr(d_x), L.ptr(d_s), L.ptr(y),
                                 L.ptr(ws), ws.numel(), st))
    torch.cuda.synchronize()
    rep = Report()
    rep.add(f"convnext C={C} T={T}", y, ref, 1e-5)
    rep.done()


def C_void(v):
    return C.c_void_p(v)


@pytest.mark.parametrize("prefix", ["generator.basegen.amp_prior_block", "generator.basegen.phase_prior_block"])
def test_adain_resblock(env, prefix):
    from oracle import blocks
    from stylish_tts_amd import lib as L
    lib, m, P = L.load(), env["m"], env["P"]
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 32, 700, generator=g)
    style = torch.randn(2, 64, generator=g)
    with torch.no_grad():
        ref = blocks.gen_resblock(P, prefix, x, style)
    y = torch.empty(2, 32, 700, device=DEV)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    d_x, d_s = dev(x), dev(style)
    L.check(lib.sty_resblock_fwd(m._handle, prefix.encode(), 2, 700, L.ptr(d_x), L.ptr(d_s), L.ptr(y),
                                 L.ptr(ws), ws.numel(), C_void(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    rep = Report()
    rep.add(prefix.rsplit(".", 1)[1], y, ref, 1e-5)
    rep.done()


def _f64(P):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}


@pytest.mark.parametrize("kind,prefix,C,T", [
    ("convnext", "generator.basegen.phase_convnext.0", 32, 600),
    ("convnext", "generator.basegen.upblocks.2", 32, 1031),
    ("convnext", "generator.basegen.upblocks.1", 64, 300),
    ("convnext", "generator.basegen.amp_convnext.2", 256, 80),
    ("resblock", "generator.basegen.amp_prior_block", 32, 700),
    ("resblock", "generator.basegen.phase_prior_block", 32, 513),
])
def test_block_backward_vs_float64_oracle(env, kind, prefix, C, T):
    """sty_block_fwd_bwd: ONE sub-module of the vocoder in the training graph -- the recompute-based fused ConvNeXt32
    backward (convnext32_bwd_kernel<1,2>), the generic ConvNeXt backward (depthwise conv, AdaLN, Snake, GRN), the
    AdaIN + Snake prologue backward (pro_bwd, adain_fold_bwd) around the resblock convs -- against the autograd of the
    oracle run in FLOAT64: input gradient, style gradient and every parameter gradient of the block at <= 1e-4 of the
    tensor's scale.  (End to end these kernels sit behind 3e-2 gates: 20 normalisation layers amplify the fp32
    summation-order noise there; one block does not.)"""
    import stylish_tts_amd as S
    from oracle import blocks
    P = {k: v.clone() for k, v in env["P"].items()}
    g = torch.Generator().manual_seed(C + T)
    x = torch.randn(2, C, T, generator=g)
    style = torch.randn(2, 64, generator=g)
    gy = torch.randn(2, C, T, generator=g)
    P64 = _f64(P)
    keys = [k for k in P64 if k.startswith(prefix + ".") and P64[k].is_floating_point()]
    for k in keys:
        P64[k].requires_grad_(True)
    x64, s64 = x.double().requires_grad_(True), style.double().requires_grad_(True)
    fn = blocks.convnext_block if kind == "convnext" else blocks.gen_resblock
    y64 = fn(P64, prefix, x64, s64)
    (y64 * gy.double()).sum().backward()
    m = S.SpeechPredictor()
    m.load_state_dict(P, strict=False)
    m = m.to(DEV).enable_training()
    m._ensure(torch.device(DEV))
    for p_ in m.parameters():
        p_.grad.zero_()
    y, gx, d_style = m.block_forward_backward(kind, prefix, dev(x), dev(style), dev(gy))
    torch.cuda.synchronize()
    rep = Report()
    rep.add("y", y, y64.detach().float(), 2e-5)
    rep.add("d x", gx, x64.grad.float(), 1e-4)
    rep.add("d style", d_style, s64.grad.float(), 1e-4)
    named = dict(m.named_parameters())
    for k in keys:
        if P64[k].grad is None or k not in named:
            continue
        ref = P64[k].grad.float()
        if ref.abs().max().item() < 1e-7 * max(1.0, gy.abs().max().item()):
            continue  # structurally zero gradients (a conv bias in front of an instance norm): noise on both sides
        got = named[k].grad
        k1 = k.replace("original0", "original1")
        if k.endswith(".original0") and k1 in keys:
            # d g_c = sum_ik dW[c,i,k] v[c,i,k] / |v_c|.  In front of an instance norm (convs1 -> adain2) the sum cancels to
            # ~0 (the norm removes the channel's scale), so the noise floor is that of the TERMS: measure the error against
            # sum_ik |d v| |v| / g, the size of what is being cancelled, when that is the larger scale.
            terms = (P64[k1].grad.abs() * P64[k1].detach().abs()).sum(dim=(1, 2), keepdim=True) / P64[k].detach().abs()
            if ref.abs().max().item() < 1e-3 * terms.max().item():
                e = (got.detach().float().cpu() - ref).abs().max().item() / terms.max().item()
                ok = e <= 1e-4
                rep.rows.append(f"  {'d ' + k[len(prefix) + 1:]:32s} err/cancelled {e:9.3e}  tol 1.0e-04  {'ok' if ok else 'FAIL'}")
                if not ok:
                    rep.bad.append(k)
                continue
        rep.add("d " + k[len(prefix) + 1:], got, ref, 1e-4)
    rep.done()


@pytest.mark.parametrize("kind,prefix,C,T", [
    ("convnext", "generator.basegen.phase_convnext.3", 32, 777),   # fused convnext32_kernel / _bwd_kernel<*,true>, wgrad_cnx
    ("convnext", "generator.basegen.phase_convnext.5", 32, 1032),  # T % 8 == 0: the LEAN backward (h kept as bf16, ds / dW2
    ("convnext", "generator.basegen.upblocks.2", 32, 520),         # from M = gY h^T, d alpha from dW1; convnext_bwd.hip)
    ("convnext", "generator.basegen.upblocks.2", 32, 1031),
    ("convnext", "generator.basegen.amp_convnext.2", 256, 120),    # generic plan: pointwise convs on the bf16 conv kernels
    ("resblock", "generator.basegen.amp_prior_block", 32, 700),    # conv32p_kernel<true>, wgradp32_kernel
])
def test_block_bf16_mode_vs_float64_oracle_on_rounded_operands(env, kind, prefix, C, T):
    """The bf16 compute mode of ONE vocoder sub-module (sty_block_fwd_bwd, compute_bf16 = 1) against the float64 oracle
    computing with the SAME bf16-rounded GEMM operands (oracle.blocks.bf16_operands: forward, input-gradient and
    weight-gradient GEMMs each round both operands, everything else exact) -- the method of test_dense_conv1d_vs_torch[bf16]
    applied to the fused ConvNeXt32 kernels, wgrad_cnx, conv32p and wgradp32.  What remains between the two: fp32 vs
    float64 around the GEMMs, the hardware sine of the bf16 kernels, and operands that round the other way in bf16 because
    they differ in the 7th digit -- one product in 128 moving by 2^-8 -- i.e. a few 1e-4 of the scale, against the 1e-2
    that separates the bf16 mode from fp32."""
    import stylish_tts_amd as S
    from oracle import blocks
    P = {k: v.clone() for k, v in env["P"].items()}
    g = torch.Generator().manual_seed(C + T)
    x = torch.randn(2, C, T, generator=g)
    style = torch.randn(2, 64, generator=g)
    gy = torch.randn(2, C, T, generator=g)
    P64 = _f64(P)
    keys = [k for k in P64 if k.startswith(prefix + ".") and P64[k].is_floating_point()]
    for k in keys:
        P64[k].requires_grad_(True)
    x64, s64 = x.double().requires_grad_(True), style.double().requires_grad_(True)
    fn = blocks.convnext_block if kind == "convnext" else blocks.gen_resblock
    with blocks.bf16_operands():
        y64 = fn(P64, prefix, x64, s64)
        (y64 * gy.double()).sum().backward()
    m = S.SpeechPredictor()
    m.load_state_dict(P, strict=False)
    m = m.to(DEV).enable_training()
    m._ensure(torch.device(DEV))
    for p_ in m.parameters():
        p_.grad.zero_()
    y, gx, d_style = m.block_forward_backward(kind, prefix, dev(x), dev(style), dev(gy), compute_bf16=True)
    torch.cuda.synchronize()
    rep = Report()
    # one ConvNeXt block: two chained GEMMs; the resblock: six convs of 352 products each, each behind an instance norm and
    # a Snake whose 7th-digit differences re-round their inputs -- the flips compound (its single convs are pinned at the
    # fp32 tolerance by test_persistent_conv32_vs_torch)
    tol = 1e-3 if kind == "convnext" else 6e-3
    if kind == "convnext" and C == 32 and T % 8 == 0:
        # the lean backward takes ds and dW2 from M = bf(gY) bf(h)^T -- h rounded BEFORE the GRN scale where the forward's GEMM
        # rounded h s, and where the two-pass kernel summed U h with h in fp32 -- and d alpha from the bf16-operand dW1 GEMM:
        # rounding points one factor earlier, 2^-9 per product, the error class of every weight gradient of the mode
        tol = 6e-3
    rep.add("y", y, y64.detach().float(), tol)
    rep.add("d x", gx, x64.grad.float(), tol)
    rep.add("d style", d_style, s64.grad.float(), tol)
    named = dict(m.named_parameters())
    for k in keys:
        if P64[k].grad is None or k not in named:
            continue
        ref = P64[k].grad.float()
        if ref.abs().max().item() < 1e-7 * max(1.0, gy.abs().max().item()):
            continue
        k1 = k.replace("original0", "original1")
        if k.endswith(".original0") and k1 in keys:
            terms = (P64[k1].grad.abs() * P64[k1].detach().abs()).sum(dim=(1, 2), keepdim=True) / P64[k].detach().abs()
            if ref.abs().max().item() < 1e-3 * terms.max().item():
                continue  # cancels to ~0 in front of an instance norm (see test_block_backward_vs_float64_oracle)
        rep.add("d " + k[len(prefix) + 1:], named[k].grad, ref, tol)
    rep.done()


@pytest.mark.parametrize("prefix,T", [("generator.basegen.amp_prior_block", 704), ("generator.basegen.phase_prior_block", 1544)])
def test_resblock_bf16_storage_vs_float64_oracle_with_the_same_rounding_points(env, prefix, T, monkeypatch):
    """bf16 STORAGE of the resblock's internal tensors (conv32p_kernel's two-byte input / residual /
    output stages, wgradp32_kernel's two-byte source, the fused prologue + instance-norm backward pro_bwd_adain_kernel) against
    the float64 oracle with the SAME rounding points (oracle.blocks.bf16_operands(storage=True): conv outputs inside the block
    rounded once after bias / residual, the input gradient of a conv on a stored tensor rounded once) at the tolerance of the
    operand-rounding test above.  The persistent kernel is forced onto these small shapes (STY_CONV32P_MIN_TILES=1); the
    same block with STY_NO_ACT16=1 (fp32 storage) must differ from it -- the rounding is real -- and sit within the same
    tolerance of the oracle WITHOUT the storage rule."""
    import stylish_tts_amd as S
    from oracle import blocks
    monkeypatch.setenv("STY_CONV32P_MIN_TILES", "1")
    P = {k: v.clone() for k, v in env["P"].items()}
    g = torch.Generator().manual_seed(T)
    x = torch.randn(2, 32, T, generator=g)
    style = torch.randn(2, 64, generator=g)
    gy = torch.randn(2, 32, T, generator=g)
    outs = {}
    for storage in (True, False):
        if not storage:
            monkeypatch.setenv("STY_NO_ACT16", "1")
        P64 = _f64(P)
        keys = [k for k in P64 if k.startswith(prefix + ".") and P64[k].is_floating_point()]
        for k in keys:
            P64[k].requires_grad_(True)
        x64, s64 = x.double().requires_grad_(True), style.double().requires_grad_(True)
        with blocks.bf16_operands(storage=storage):
            y64 = blocks.gen_resblock(P64, prefix, x64, s64)
            (y64 * gy.double()).sum().backward()
        m = S.SpeechPredictor()
        m.load_state_dict(P, strict=False)
        m = m.to(DEV).enable_training()
        m._ensure(torch.device(DEV))
        for p_ in m.parameters():
            p_.grad.zero_()
        y, gx, d_style = m.block_forward_backward("resblock", prefix, dev(x), dev(style), dev(gy), compute_bf16=True)
        torch.cuda.synchronize()
        outs[storage] = (y.cpu(), gx.cpu())
        rep = Report()
        # (a STORED value that rounds the other way because fp32 and float64 differ in its 7th digit moves by one bf16 ulp,
        # 2^-7 of its magnitude, where an operand flip moved one product of 352: the max-norm gate is two such flips wide,
        # and the flips are sparse -- the relative L2 distance stays at the operand test's level)
        tol = 1.6e-2 if storage else 8e-3  # (fp32 storage: the operand test's 6e-3 + the persistent kernels' hardware sine, round 5)
        rep.add(f"y (storage={storage})", y, y64.detach().float(), tol)
        rep.add("d x", gx, x64.grad.float(), tol)
        # (relative L2: y behind two stored tensors, d x behind five stored tensors and six stored input gradients)
        for name, got, ref_, t2 in (("y", y, y64.detach(), 4e-3), ("d x", gx, x64.grad, 1e-2)):
            l2 = ((got.detach().cpu().double() - ref_) ** 2).sum().sqrt().item() / ref_.norm().item()
            rep.rows.append(f"  {name + ' (relative L2)':32s} rel_err {l2:9.3e}  tol {t2:.1e}  {'ok' if l2 <= t2 else 'FAIL'}")
            if l2 > t2:
                rep.bad.append(name + " L2")
        rep.add("d style", d_style, s64.grad.float(), tol)
        named = dict(m.named_parameters())
        for k in keys:
            if P64[k].grad is None or k not in named:
                continue
            ref = P64[k].grad.float()
            if ref.abs().max().item() < 1e-7 * max(1.0, gy.abs().max().item()):
                continue
            k1 = k.replace("original0", "original1")
            if k.endswith(".original0") and k1 in keys:
                terms = (P64[k1].grad.abs() * P64[k1].detach().abs()).sum(dim=(1, 2), keepdim=True) / P64[k].detach().abs()
                if ref.abs().max().item() < 1e-3 * terms.max().item():
                    continue
            # (d alpha = sum_t u (z sin(2 a z) - sin^2(a z) / a) / a is a cancellation of two terms of the size of the tensor it
            # scales: a stored value that rounds the other way moves it by a multiple of its own 2^-8 -- measured 1.0e-2 ... 2.2e-2
            # over the library versions of round 5, against 2e-3 for the same tensors with fp32 storage)
            rep.add("d " + k[len(prefix) + 1:], named[k].grad, ref, 3e-2 if (storage and ".alpha" in k) else tol)
        rep.done()
    dy = (outs[True][0] - outs[False][0]).abs().max().item() / outs[False][0].abs().max().item()
    print(f"  two-byte storage vs fp32 storage: y differs by {dy:.2e} of its scale")
    assert 1e-5 < dy < 5e-2, dy


def test_convnext32_block_bf16_mode_vs_fp32_mode(env):
    """The fused ConvNeXt32 backward in the bf16 compute mode (its three GEMMs on v_mfma_f32_32x32x16_bf16, the chained
    one with the accumulator fragment as B operand in the permuted row order) against the SAME kernels in fp32 mode:
    only the operand rounding separates them (2^-9 relative per product, averaged down by the 32 / 128-term sums), so
    every output and every parameter gradient has to agree to 1e-2 of its scale -- a wrong fragment order is O(1)."""
    import stylish_tts_amd as S
    prefix, C, T = "generator.basegen.phase_convnext.3", 32, 777
    g = torch.Generator().manual_seed(12)
    x, style, gy = torch.randn(2, C, T, generator=g), torch.randn(2, 64, generator=g), torch.randn(2, C, T, generator=g)
    res = {}
    for mode in (False, True):
        m = S.SpeechPredictor()
        m.load_state_dict({k: v.clone() for k, v in env["P"].items()}, strict=False)
        m = m.to(DEV).enable_training()
        m._ensure(torch.device(DEV))
        for p_ in m.parameters():
            p_.grad.zero_()
        y, gx, d_style = m.block_forward_backward("convnext", prefix, dev(x), dev(style), dev(gy), compute_bf16=mode)
        torch.cuda.synchronize()
        res[mode] = dict(y=y.cpu(), gx=gx.cpu(), d_style=d_style.cpu(),
                         **{k[len(prefix) + 1:]: p_.grad.cpu().clone() for k, p_ in m.named_parameters()
                            if k.startswith(prefix + ".")})
    rep = Report()
    for k in res[False]:
        if res[False][k].abs().max().item() > 0:
            rep.add(k, res[True][k], res[False][k], 1e-2)
    rep.done()
    assert not torch.equal(res[True]["gx"], res[False]["gx"])  # the bf16 kernels really ran


@pytest.mark.parametrize("prefix,T", [("generator.basegen.phase_convnext.5", 1032), ("generator.basegen.phase_convnext.1", 520)])
def test_convnext32_two_byte_gradients_vs_float64_oracle_with_the_same_rounding_points(env, prefix, T, monkeypatch):
    """Two-byte GRADIENTS of the 32-channel ConvNeXt chain: inside the chain the lean fused backward
    reads its output gradient as a bf16 tensor, writes its input gradient (fused epilogue) and gU as bf16 -- each value rounded
    once where it is stored.  STY_BLOCK_G16=1 puts ONE block into that situation (gy rounded on the way in, a bf16 d x
    converted back on the way out) and the float64 oracle states the same rule (`round_grad` on the block's input, on its
    depthwise-conv output and on its output, inside bf16_operands(storage=True)): same tolerance as the operand-rounding
    test of the lean backward.  The same block with STY_NO_GRAD16=1 must differ: the rounding is real."""
    import stylish_tts_amd as S
    from oracle import blocks
    C = 32
    P = {k: v.clone() for k, v in env["P"].items()}
    g = torch.Generator().manual_seed(7 + T)
    x, style, gy = torch.randn(2, C, T, generator=g), torch.randn(2, 64, generator=g), torch.randn(2, C, T, generator=g)
    P64 = _f64(P)
    keys = [k for k in P64 if k.startswith(prefix + ".") and P64[k].is_floating_point()]
    for k in keys:
        P64[k].requires_grad_(True)
    x64, s64 = x.double().requires_grad_(True), style.double().requires_grad_(True)
    with blocks.bf16_operands(storage=True):
        y64 = blocks.round_grad(blocks.convnext_block(P64, prefix, x64, s64, grad16=True))
        (y64 * gy.double()).sum().backward()
    monkeypatch.setenv("STY_BLOCK_G16", "1")
    monkeypatch.setenv("STY_GRAD16_STREAM", "1")  # (round 6: the stream's two-byte gradient is opt-in; gU alone is the default)
    gxs = {}
    for g16 in (True, False):
        if g16:
            monkeypatch.delenv("STY_NO_GRAD16", raising=False)
        else:
            monkeypatch.setenv("STY_NO_GRAD16", "1")
        m = S.SpeechPredictor()
        m.load_state_dict(P, strict=False)
        m = m.to(DEV).enable_training()
        m._ensure(torch.device(DEV))
        for p_ in m.parameters():
            p_.grad.zero_()
        y, gx, d_style = m.block_forward_backward("convnext", prefix, dev(x), dev(style), dev(gy), compute_bf16=True)
        torch.cuda.synchronize()
        gxs[g16] = gx.cpu()
        if not g16:
            continue
        rep = Report()
        tol = 6e-3
        rep.add("y", y, y64.detach().float(), tol)
        # d x is a STORED bf16 value: where the two sides differ in the 7th digit across a rounding boundary the stored values are
        # one bf16 step apart -- 2^-8 of the element, 3.9e-3 of the scale for the largest ones -- on top of the 2.5e-3 of the
        # operand-rounding test; in relative L2 the tensor agrees to 4e-3
        rep.add("d x", gx, x64.grad.float(), 1.2e-2)
        l2 = ((gx.cpu().double() - x64.grad).norm() / x64.grad.norm()).item()
        print(f"  d x relative L2 {l2:.2e}")
        assert l2 < 4e-3, l2
        rep.add("d style", d_style, s64.grad.float(), tol)
        named = dict(m.named_parameters())
        for k in keys:
            if P64[k].grad is None or k not in named:
                continue
            ref = P64[k].grad.float()
            if ref.abs().max().item() < 1e-7 * max(1.0, gy.abs().max().item()):
                continue
            rep.add("d " + k[len(prefix) + 1:], named[k].grad, ref, tol)
        rep.done()
        # d x IS a bf16 tensor: every value survives a round trip through bf16
        assert torch.equal(gx.cpu(), gx.cpu().bfloat16().float())
    d = (gxs[True] - gxs[False]).abs().max().item() / gxs[False].abs().max().item()
    print(f"  two-byte gradients vs fp32 gradients: d x differs by {d:.2e} of its scale")
    assert 1e-5 < d < 2e-2, d


@pytest.mark.parametrize("T,bf16", [(8, True), (248, True), (256, True), (264, True), (504, True), (1032, True),
                                    (260, True), (264, False), (1032, False)])
def test_convnext32_backward_with_the_input_gradient_fused_equals_the_separate_kernels(env, T, bf16, monkeypatch):
    """Round 5: the lean ConvNeXt32 backward writes gX = gY + dwconv^T(gU) itself (overlapping tiles of 248 owned columns,
    convnext_bwd.hip) and leaves xn as bf16 for the dW1 GEMM.  Against the same block with both switched off
    (STY_NO_CNX_GX / STY_NO_CNX_XN16: dwconv7_bwd_dx_kernel, fp32 xn rounded at the GEMM's load): d x to fp32 summation order,
    every parameter gradient to the order of the per-tile partial sums (the tiles differ: 248 vs 256 columns).  T covers one
    tile, the tile edges (248, 256, 264: the second tile owns 4 ... 16 columns) and many tiles; T = 260 is the two-pass bf16
    kernel (T % 8 != 0), bf16 = False the fp32 mode (c2): there only the fused input gradient applies."""
    import stylish_tts_amd as S
    prefix, C = "generator.basegen.phase_convnext.2", 32
    g = torch.Generator().manual_seed(100 + T)
    x, style, gy = torch.randn(2, C, T, generator=g), torch.randn(2, 64, generator=g), torch.randn(2, C, T, generator=g)
    res = {}
    monkeypatch.setenv("STY_NO_GRAD16", "1")  # (the two-byte gU of the fused form has a test of its own: one thing at a time)
    for fused in (False, True):
        for k in ("STY_NO_CNX_GX", "STY_NO_CNX_XN16"):
            if fused:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, "1")
        m = S.SpeechPredictor()
        m.load_state_dict({k: v.clone() for k, v in env["P"].items()}, strict=False)
        m = m.to(DEV).enable_training()
        m._ensure(torch.device(DEV))
        for p_ in m.parameters():
            p_.grad.zero_()
        y, gx, d_style = m.block_forward_backward("convnext", prefix, dev(x), dev(style), dev(gy), compute_bf16=bf16)
        torch.cuda.synchronize()
        res[fused] = dict(y=y.cpu(), gx=gx.cpu(), d_style=d_style.cpu(),
                          **{k[len(prefix) + 1:]: p_.grad.cpu().clone() for k, p_ in m.named_parameters()
                             if k.startswith(prefix + ".")})
    rep = Report()
    assert torch.equal(res[True]["y"], res[False]["y"])
    for k in res[False]:
        if res[False][k].abs().max().item() > 0:
            rep.add(k, res[True][k], res[False][k], 2e-6 if k == "gx" else 2e-5)
    rep.done()


@pytest.mark.parametrize("DH,T,masked", [(16, 40, True), (16, 100, True), (64, 160, False), (64, 520, False),
                                         (160, 200, True), (160, 520, True), (96, 77, True), (64, 160, True)])
def test_attention_backward_vs_float64(DH, T, masked):
    """sty_attention_fwd_bwd: the text encoder's masked attention (DH = 16, VALU backward kernels attn_bwd_{a,b,c}), the
    conformer's (DH = 64, MFMA backward attn_bwd_{kv,q}_mfma) and the prosody encoder's (2 heads x 160 / 96 with the length
    mask, the same MFMA backward since round 6) vs float64 softmax attention: o, dq, dk, dv."""
    from stylish_tts_amd import lib as L
    lib = L.load()
    B, H = 3, (8 if DH <= 64 else 2)
    g = torch.Generator().manual_seed(DH + T)
    q, k, v, do = (torch.randn(B, H * DH, T, generator=g) for _ in range(4))
    lengths = torch.tensor([T, max(1, T - 7), max(1, T // 2)]) if masked else None
    valid = torch.ones(B, 1, T)
    if masked:
        # padded QUERY rows: the reference adds -1e4 to every score of the row, which in fp32 rounds the scores to 1e-3 --
        # its own output there is noise that the encoder masks right after (text_encoder.py:157-163), and no gradient comes
        # back through them.  The test does the same: zero d o on those rows and compare o on the valid ones.
        valid = (torch.arange(T)[None, :] < lengths[:, None]).float()[:, None, :]
        do = do * valid
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    qh, kh, vh = (t.view(B, H, DH, T).transpose(2, 3) for t in (q64, k64, v64))
    sc = qh @ kh.transpose(2, 3) / DH ** 0.5
    if masked:
        ar = torch.arange(T)
        ok = (ar[None, :] < lengths[:, None])
        mask2 = ok[:, None, :, None] & ok[:, None, None, :]
        sc = sc + torch.zeros_like(sc).masked_fill(~mask2, -1e4)  # the additive -1e4 mask SDPA gets (text_encoder.py:262-276)
    o64 = (torch.softmax(sc, dim=-1) @ vh).transpose(2, 3).reshape(B, H * DH, T)
    (o64 * do.double()).sum().backward()
    o, dq, dk, dv = (torch.empty(B, H * DH, T, device=DEV) for _ in range(4))
    need = C.c_size_t()
    L.check(lib.sty_attention_workspace_bytes(B, H, T, C.byref(need)))
    ws = torch.empty(need.value, dtype=torch.uint8, device=DEV)
    ln = dev(lengths) if masked else None
    dq_, dk_, dv_, ddo = dev(q), dev(k), dev(v), dev(do)  # named: a temporary's memory would be recycled by the next one
    L.check(lib.sty_attention_fwd_bwd(B, H, DH, T, L.ptr(dq_), L.ptr(dk_), L.ptr(dv_), L.ptr(ln), L.ptr(ddo),
                                      L.ptr(o), L.ptr(dq), L.ptr(dk), L.ptr(dv), L.ptr(ws), ws.numel(),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    rep = Report()
    rep.add("o", o * dev(valid), o64.detach().float() * valid, 1e-5)
    rep.add("dq", dq, q64.grad.float(), 1e-4)
    rep.add("dk", dk, k64.grad.float(), 1e-4)
    rep.add("dv", dv, v64.grad.float(), 1e-4)
    rep.done()


@pytest.mark.parametrize("T", [160, 520])
def test_attention_bf16_mode_vs_float64_on_rounded_operands(T, monkeypatch):
    """attn16.hip (the conformer's attention in the bf16 compute mode: 8 x 64 heads, no mask; the five contractions on the
    bf16 matrix cores, softmax statistics in fp32) through sty_attention_fwd_bwd with STY_ATTN_UNIT_BF16=1, against float64
    softmax attention on the bf16-ROUNDED q, k, v, d o.  What is left between the two is the rounding of P and dS (2^-9 per
    element, inside the kernels only): 1e-2 of the tensor scale; the same call without the switch (fp32 kernels) on the same
    rounded inputs must sit 100 x closer to the float64 result, so a wrong fragment map cannot hide in the tolerance."""
    from stylish_tts_amd import lib as L
    lib = L.load()
    B, H, DH = 2, 8, 64
    g = torch.Generator().manual_seed(T)
    q, k, v, do = (torch.randn(B, H * DH, T, generator=g).bfloat16().float() for _ in range(4))
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    qh, kh, vh = (t.view(B, H, DH, T).transpose(2, 3) for t in (q64, k64, v64))
    o64 = (torch.softmax(qh @ kh.transpose(2, 3) / DH ** 0.5, dim=-1) @ vh).transpose(2, 3).reshape(B, H * DH, T)
    (o64 * do.double()).sum().backward()
    ref = dict(o=o64.detach().float(), dq=q64.grad.float(), dk=k64.grad.float(), dv=v64.grad.float())
    need = C.c_size_t()
    L.check(lib.sty_attention_workspace_bytes(B, H, T, C.byref(need)))
    res = {}
    for mode in ("fp32", "bf16"):
        if mode == "bf16":
            monkeypatch.setenv("STY_ATTN_UNIT_BF16", "1")
        else:
            monkeypatch.delenv("STY_ATTN_UNIT_BF16", raising=False)
        ws = torch.empty(need.value, dtype=torch.uint8, device=DEV)
        o, dq, dk, dv = (torch.empty(B, H * DH, T, device=DEV) for _ in range(4))
        dq_, dk_, dv_, ddo = dev(q), dev(k), dev(v), dev(do)
        L.check(lib.sty_attention_fwd_bwd(B, H, DH, T, L.ptr(dq_), L.ptr(dk_), L.ptr(dv_), None, L.ptr(ddo), L.ptr(o), L.ptr(dq),
                                          L.ptr(dk), L.ptr(dv), L.ptr(ws), ws.numel(),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        res[mode] = dict(o=o.cpu(), dq=dq.cpu(), dk=dk.cpu(), dv=dv.cpu())
    rep = Report()
    for kx in ("o", "dq", "dk", "dv"):
        rep.add(f"fp32 kernels {kx}", res["fp32"][kx], ref[kx], 1e-4)
        rep.add(f"bf16 kernels {kx}", res["bf16"][kx], ref[kx], 1e-2)
    rep.done()
    assert not torch.equal(res["fp32"]["o"], res["bf16"]["o"])  # the bf16 kernels really ran


def test_alignment(env):
    from oracle import frontend
    from stylish_tts_amd import lib as L
    lib = L.load()
    dur = env["cs"]["durations"]
    ref = frontend.duration_to_alignment(dur)
    B, Lt = dur.shape
    T = ref.shape[2]
    out = torch.empty(B, Lt, T, device=DEV)
    d_dur = dev(dur)
    L.check(lib.sty_alignment_fwd(B, Lt, T, L.ptr(d_dur), L.ptr(out), None))
    rep = Report()
    rep.add("alignment", out, ref, 1e-6)
    rep.done()


def _mel_l1(a, b):
    from oracle.frontend import calculate_mel
    return (calculate_mel(a.squeeze(1), 512, 512, 300) - calculate_mel(b.squeeze(1), 512, 512, 300)).abs().mean().item()


def test_vocoder_end_to_end(env):
    m, cs, want = env["m"], env["cs"], env["want"]
    B, T = cs["pitch"].shape
    Tu = 75 * T
    taps = {k: torch.zeros(*shape, device=DEV) for k, shape in dict(
        tap_conformer_out=(B, 256, T), tap_prior=(B, 300 * T), tap_har_spec=(B, 32, Tu), tap_har_phase=(B, 32, Tu),
        tap_logamp_prior=(B, 32, Tu), tap_phase_prior=(B, 32, Tu), tap_trunk=(B, 32, Tu), tap_logamp=(B, 32, Tu)).items()}
    mel = want["decoder_out"]
    with torch.no_grad():
        # explicit source (isolates the ill-conditioned phase accumulation), then the built-in source
        a1 = m.vocoder_forward(mel=dev(mel), style=dev(cs["style"]), pitch=dev(cs["pitch"]), voiced=dev(env["voiced"]),
                               noise=dev(cs["noise"]), prior_override=dev(want["prior"]), taps=taps).audio
        torch.cuda.synchronize()
        rep = Report()
        rep.add("conformer_out", taps["tap_conformer_out"], want["conformer_out"], 1e-5)
        rep.add("har_spec", taps["tap_har_spec"], want["har_spec"], 1e-5)
        rep.add("cos(har_phase)", torch.cos(taps["tap_har_phase"]), torch.cos(want["har_phase"]), 1e-3)
        rep.add("logamp_prior", taps["tap_logamp_prior"], want["logamp_prior"], 1e-4)
        rep.add("phase_prior", taps["tap_phase_prior"], want["phase_prior"], 5e-3)
        rep.add("trunk", taps["tap_trunk"], want["trunk"], 1e-5)
        rep.add("logamp", taps["tap_logamp"], want["logamp"], 1e-5)
        ref = env["ref_audio"]
        mse = ((a1.cpu() - ref) ** 2).mean().item()
        print(f"\n  audio (explicit source): max|err| {(a1.cpu() - ref).abs().max().item():.3e} mse {mse:.3e} "
              f"mel-L1 {_mel_l1(a1.cpu(), ref):.3e}")
        rep.done()
        assert mse <= 1e-8 and _mel_l1(a1.cpu(), ref) <= 1e-3
        a2 = m.vocoder_forward(mel=dev(mel), style=dev(cs["style"]), pitch=dev(cs["pitch"]), voiced=dev(env["voiced"]),
                               noise=dev(cs["noise"])).audio
        mse2 = ((a2.cpu() - ref) ** 2).mean().item()
        print(f"  audio (built-in source): max|err| {(a2.cpu() - ref).abs().max().item():.3e} mse {mse2:.3e} "
              f"mel-L1 {_mel_l1(a2.cpu(), ref):.3e}")
        assert mse2 <= 1e-6 and _mel_l1(a2.cpu(), ref) <= 1e-3


def test_vocoder_bf16_compute_reported(env):
    """Inference with bf16 operands on the dense convs (sty_train_opts.compute_bf16, bench workload c5-bf16): outside
    the fp32 parity gates by definition; reported against the oracle audio with a loose bound (mel-L1 <= 0.05,
    waveform MSE <= 1e-3 on a tanh-bounded signal) and required to differ from the fp32 result."""
    import stylish_tts_amd as S
    cs, want = env["cs"], env["want"]
    P = {k: v.clone() for k, v in env["P"].items()}
    m = S.SpeechPredictor()
    m.load_state_dict(P, strict=False)
    m = m.to(DEV).set_train_opts(compute_bf16=True)
    with torch.no_grad():
        a = m.vocoder_forward(mel=dev(want["decoder_out"]), style=dev(cs["style"]), pitch=dev(cs["pitch"]),
                              voiced=dev(env["voiced"]), noise=dev(cs["noise"]), prior_override=dev(want["prior"])).audio
    torch.cuda.synchronize()
    ref = env["ref_audio"]
    mse = ((a.cpu() - ref) ** 2).mean().item()
    l1 = _mel_l1(a.cpu(), ref)
    print(f"\n  bf16-operand vocoder vs fp32 oracle: waveform mse {mse:.3e} (signal power {(ref ** 2).mean().item():.3e}), "
          f"mel-L1 {l1:.3e}")
    assert 1e-12 < mse <= 1e-3 and l1 <= 5e-2


def test_vocoder_bf16_inference_on_the_small_launch_kernels(env, monkeypatch):
    """Round 5: in inference the K = 1 GEMM kernel takes pwconv1 of the generic ConvNeXt blocks with its Snake output stage
    (convk1_kernel<.., 2>), the heads run a LayerNorm pass + the persistent 32-channel kernel, and dwconv_adaln_kernel<7> has
    its taps unrolled.  At the test size those launches are below the dispatch thresholds, so this test lowers them
    (STY_CONVK1_MIN_TILES = STY_CONV32P_MIN_TILES = 1) and holds the bf16 audio to the bound of
    test_vocoder_bf16_compute_reported; the SAME graph with the thresholds out of reach (every one of those convs on the tiled
    kernel with its fused epilogue / LayerNorm prologue) must agree with it in mel-L1 to the distance two bf16 computations of
    this vocoder keep (1.5e-2; measured 3.9e-3), and convk1_kernel / conv32p_kernel must have run."""
    import stylish_tts_amd as S
    from stylish_tts_amd import lib as L
    lib = L.load()
    cs, want = env["cs"], env["want"]
    P = {k: v.clone() for k, v in env["P"].items()}
    ref = env["ref_audio"]
    out = {}
    for mode in ("small", "tiled"):
        v = "1" if mode == "small" else "100000000"
        monkeypatch.setenv("STY_CONVK1_MIN_TILES", v)
        monkeypatch.setenv("STY_CONV32P_MIN_TILES", v)
        m = S.SpeechPredictor()
        m.load_state_dict(P, strict=False)
        m = m.to(DEV).set_train_opts(compute_bf16=True)
        L.prof_report(512)
        lib.sty_prof_enable(1)
        try:
            with torch.no_grad():
                a = m.vocoder_forward(mel=dev(want["decoder_out"]), style=dev(cs["style"]), pitch=dev(cs["pitch"]),
                                      voiced=dev(env["voiced"]), noise=dev(cs["noise"]), prior_override=dev(want["prior"])).audio
            torch.cuda.synchronize()
        finally:
            lib.sty_prof_enable(0)
        names = [r["name"] for r in L.prof_report(512)]
        out[mode] = (a.cpu(), names)
        mse, l1 = ((a.cpu() - ref) ** 2).mean().item(), _mel_l1(a.cpu(), ref)
        print(f"\n  {mode}: waveform mse {mse:.3e}  mel-L1 {l1:.3e}  convk1 families "
              f"{sum(n.startswith('convk1_kernel') for n in names)}  conv32p {sum(n.startswith('conv32p_kernel') for n in names)}")
        assert 1e-12 < mse <= 1e-3 and l1 <= 5e-2
    assert any(n.startswith("convk1_kernel") for n in out["small"][1]) and any(n.startswith("conv32p_kernel") for n in out["small"][1])
    assert not any(n.startswith("convk1_kernel") or n.startswith("conv32p_kernel") for n in out["tiled"][1])
    d = _mel_l1(out["small"][0], out["tiled"][0])
    print(f"  small-launch kernels vs tiled kernels: mel-L1 {d:.3e}")
    assert d <= 1.5e-2


def test_bf16_weight_fragments_follow_a_weight_update_in_inference(env, monkeypatch):
    """The bf16 A-fragment buffers of the persistent / GEMM kernels are made once per weight and kept until the model's
    prepare step re-makes them (convp16.hip, q_frags; round 5: in inference too, where every launch re-packed its weight
    before).  A weight update through load_state_dict on the SAME model object must reach the next forward: its audio has to
    equal, bit for bit, that of a fresh model built from the updated weights -- with the kernels forced onto the test's small
    launches."""
    import stylish_tts_amd as S
    for k in ("STY_CONVP16_MIN_TILES", "STY_CONVK1_MIN_TILES", "STY_CONV32P_MIN_TILES"):
        monkeypatch.setenv(k, "1")
    cs, want = env["cs"], env["want"]
    P = {k: v.clone() for k, v in env["P"].items()}

    def run(m):
        with torch.no_grad():
            a = m.vocoder_forward(mel=dev(want["decoder_out"]), style=dev(cs["style"]), pitch=dev(cs["pitch"]),
                                  voiced=dev(env["voiced"]), noise=dev(cs["noise"]), prior_override=dev(want["prior"])).audio
        torch.cuda.synchronize()
        return a.cpu()

    m = S.SpeechPredictor()
    m.load_state_dict(P, strict=False)
    m = m.to(DEV).set_train_opts(compute_bf16=True)
    a0 = run(m)
    assert torch.equal(a0, run(m))  # (second forward: every fragment buffer comes from the cache)
    P2 = {k: v.clone() for k, v in P.items()}
    g = torch.Generator().manual_seed(3)
    changed = 0
    for k in P2:
        if k.startswith("generator.") and k.endswith(".weight") and P2[k].dim() == 3 and P2[k].shape[1] >= 64:
            P2[k] = P2[k] * (1.0 + 0.05 * torch.randn(P2[k].shape, generator=g))
            changed += 1
    assert changed >= 4
    m.load_state_dict(P2, strict=False)
    a1 = run(m)
    m2 = S.SpeechPredictor()
    m2.load_state_dict(P2, strict=False)
    m2 = m2.to(DEV).set_train_opts(compute_bf16=True)
    a2 = run(m2)
    assert not torch.equal(a1, a0)
    assert torch.equal(a1, a2), (a1 - a2).abs().max().item()


def test_speech_predictor_end_to_end_vs_oracle_and_golden(env):
    from safetensors.torch import load_file
    m, cs, want, ali = env["m"], env["cs"], env["want"], env["ali"]
    B, T = cs["pitch"].shape
    Lt = cs["texts"].shape[1]
    taps = dict(tap_text_encoding=torch.zeros(B, 128, Lt, device=DEV), tap_decoder_out=torch.zeros(B, 128, T, device=DEV))
    with torch.no_grad():
        out = m(dev(cs["texts"]), dev(cs["text_lengths"]), dev(ali), dev(cs["pitch"]), dev(cs["energy"]),
                dev(env["voiced"]), dev(cs["style"]), dev(cs["pitch"]), noise=dev(cs["noise"]),
                prior_override=dev(want["prior"]), taps=taps).audio
    torch.cuda.synchronize()
    rep = Report()
    rep.add("text_encoding", taps["tap_text_encoding"], want["text_encoding"], 1e-5)
    rep.add("decoder_out", taps["tap_decoder_out"], want["decoder_out"], 1e-5)
    gold = load_file(os.path.join(G, "sp_small.safetensors"))
    rep.add("text_encoding vs reference", taps["tap_text_encoding"], gold["text_encoding"], 1e-5)
    rep.add("decoder_out vs reference", taps["tap_decoder_out"], gold["decoder_out"], 1e-5)
    for name, ref in (("oracle", env["ref_audio"]), ("reference golden", gold["audio"])):
        err = (out.cpu() - ref).abs()
        mse = (err ** 2).mean().item()
        print(f"\n  audio vs {name}: max|err| {err.max().item():.3e} mse {mse:.3e} mel-L1 {_mel_l1(out.cpu(), ref):.3e}")
        assert mse <= 1e-8 and _mel_l1(out.cpu(), ref) <= 1e-3
    rep.done()


def test_vocoder_determinism_full_size(env):
    """BASELINE config c5 shape (B=8, T=800): the forward is deterministic and bounded.  (The comparison with the
    oracle at this size, and the batch-independence check, are in tests/test_full_size.py.)"""
    m = env["m"]
    g = torch.Generator().manual_seed(7)
    B, T = 8, 800
    mel = torch.randn(B, 128, T, generator=g)
    style = torch.randn(B, 64, generator=g)
    pitch = torch.rand(B, T, generator=g) * 200 + 80
    pitch[torch.rand(B, T, generator=g) < 0.3] = 0
    voiced = (pitch > 20).float()
    with torch.no_grad():
        a = m.vocoder_forward(mel=dev(mel), style=dev(style), pitch=dev(pitch), voiced=dev(voiced), seed=3).audio
        b = m.vocoder_forward(mel=dev(mel), style=dev(style), pitch=dev(pitch), voiced=dev(voiced), seed=3).audio
    torch.cuda.synchronize()
    assert a.shape == (B, 1, 300 * T) and bool(torch.isfinite(a).all()) and a.abs().max().item() <= 1.0
    assert torch.equal(a, b), "forward is not deterministic"


@pytest.mark.parametrize("T", [80, 161, 222])
def test_mel_style_encoder(T):
    """A2: MelStyleEncoder (conv2d stack) vs the oracle, and vs the reference's golden vector at T=80."""
    import stylish_tts_amd as S
    from oracle import style_encoder as ose
    from oracle.manifest import style_encoder_manifest
    from oracle.weights import fill_state_dict
    from safetensors.torch import load_file
    from tests.cases import make_case
    P = fill_state_dict(style_encoder_manifest(), 0)
    m = S.MelStyleEncoder()
    m.load_state_dict(P)
    m = m.to(DEV)
    if T == 80:
        x = make_case("se_small")["mel"]
    else:
        x = torch.randn(3, 1, 80, T, generator=torch.Generator().manual_seed(T))
    with torch.no_grad():
        ref = ose.mel_style_encoder(P, "", x)
        out = m(dev(x))
    torch.cuda.synchronize()
    rep = Report()
    rep.add(f"style T={T} vs oracle", out, ref, 1e-5)
    if T == 80:
        rep.add("style vs reference golden", out, load_file(os.path.join(G, "se_small.safetensors"))["style"], 1e-5)
    rep.done()


def _test_audio(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(N) / 24000.0
    f0 = 110.0 + 60.0 * torch.rand(B, 1, generator=g)
    x = sum(torch.sin(2 * torch.pi * f0 * (h + 1) * t) / (h + 1) for h in range(6)) * 0.2
    return x + 0.01 * torch.randn(B, N, generator=g)


@pytest.mark.parametrize("n_fft,win", [(512, 512), (2048, 1200)])
def test_mel_front_end(n_fft, win):
    """A1: calculate_mel + log energy vs the oracle restatement (torchaudio boundary: parity unpinned)."""
    from oracle import frontend as ofe
    from stylish_tts_amd.frontend import MelSpec, calculate_mel
    audio = _test_audio(3, 24000 + 300 * 7, n_fft)
    ref = ofe.calculate_mel(audio, n_fft, win, 300)
    ref_e = ofe.log_energy(ref)
    mel, length, energy = calculate_mel(dev(audio), MelSpec(n_fft, win, 300), -4.0, 4.0, want_energy=True)
    torch.cuda.synchronize()
    assert mel.shape == ref.shape and int(length[0]) == ref.shape[2]
    l1 = (mel.cpu() - ref).abs().mean().item()
    print(f"\n  mel n_fft={n_fft}: L1 {l1:.3e}  max {(mel.cpu() - ref).abs().max().item():.3e}")
    assert l1 <= 1e-3  # north-star gate
    rep = Report()
    rep.add(f"mel n_fft={n_fft}", mel, ref, 1e-4)
    rep.add("energy", energy, ref_e, 1e-4)
    rep.done()


def test_multi_spectrogram():
    """A10: three-resolution STFT features vs the oracle (torch.stft + restated MelScale)."""
    from oracle import frontend as ofe
    from stylish_tts_amd.frontend import MultiSpectrogram
    audio = _test_audio(2, 24000, 5)
    ms = MultiSpectrogram(sample_rate=24000)
    with torch.no_grad():
        mags, phases, ffts = ms.calculate(dev(audio))
    torch.cuda.synchronize()
    rep = Report()
    for i, (fft, hop, win) in enumerate(ofe.RESOLUTIONS):
        r_mag, r_phase, r_fft = ofe.multi_spectrogram_single(audio, fft, hop, win)
        rep.add(f"fft_mag {fft}", ffts[i], r_fft, 1e-5)
        rep.add(f"log1p mel128 {fft}", mags[i], r_mag, 1e-5)
        # phase: compare where both sides are gated on and away from the +-pi wrap
        gate = (r_fft[:, 0] > 2e-3)
        d = torch.remainder(phases[i].cpu() - r_phase + torch.pi, 2 * torch.pi) - torch.pi
        err = (d.abs() * gate).max().item()
        rep.rows.append(f"  phase {fft:5d} (gated)              max wrapped err {err:9.3e}")
        if err > 5e-3:
            rep.bad.append(f"phase {fft}")
    rep.done()


def test_acoustic_step_forward(env):
    """A0: the AcousticStep tensor flow end to end (mel -> style encoder -> predictor -> multi-spectrogram)."""
    import stylish_tts_amd as S
    from oracle import frontend as ofe, speech_predictor as osp
    from oracle.manifest import style_encoder_manifest
    from oracle.weights import fill_state_dict
    from stylish_tts_amd.acoustic import acoustic_forward
    from stylish_tts_amd.frontend import MultiSpectrogram
    cs = env["cs"]
    B, T = cs["pitch"].shape
    audio_gt = _test_audio(B, 300 * T, 21)
    Pse = fill_state_dict(style_encoder_manifest(), 0)
    se = S.MelStyleEncoder()
    se.load_state_dict(Pse)
    se = se.to(DEV)
    want = {}
    with torch.no_grad():
        ref = osp.acoustic_forward(env["P"], Pse, audio_gt, cs["texts"], cs["text_lengths"], cs["pitch"],
                                   cs["durations"], cs["noise"], want)
    out = acoustic_forward(env["m"], se, audio_gt=dev(audio_gt), texts=dev(cs["texts"]),
                           text_lengths=dev(cs["text_lengths"]), pitch=dev(cs["pitch"]), durations=dev(cs["durations"]),
                           noise=dev(cs["noise"]), prior_override=dev(want["prior"]),
                           multi_spectrogram=MultiSpectrogram(sample_rate=24000))
    torch.cuda.synchronize()
    rep = Report()
    rep.add("mel", out.mel, want["mel"], 1e-4)
    rep.add("style_mel", out.style_mel, want["style_mel"], 1e-4)
    rep.add("energy", out.energy, want["energy"], 1e-4)
    rep.add("alignment", out.alignment, want["alignment"], 1e-6)
    rep.add("speech_style", out.speech_style, want["style"], 1e-4)
    err = (out.pred.audio.cpu() - ref).abs()
    mse = (err ** 2).mean().item()
    print(f"\n  acoustic audio: max|err| {err.max().item():.3e} mse {mse:.3e} mel-L1 {_mel_l1(out.pred.audio.cpu(), ref):.3e}")
    r_mag, _, r_fft = ofe.multi_spectrogram_single(ref.squeeze(1), 1024, 256, 1024)
    rep.add("pred_fft[1]", out.pred_fft[1], r_fft, 1e-3)
    rep.add("pred_spec[1]", out.pred_spec[1], r_mag, 1e-3)
    rep.done()
    assert mse <= 1e-7 and _mel_l1(out.pred.audio.cpu(), ref) <= 1e-3


def test_vocoder_backward(env):
    """K15 (vocoder): gradients of mean|audio| w.r.t. mel, style and parameters vs the oracle's autograd."""
    import stylish_tts_amd as S
    from oracle import vocoder as ov
    cs, want = env["cs"], env["want"]
    P = {k: v.clone() for k, v in env["P"].items()}
    m = S.SpeechPredictor()
    m.load_state_dict(P, strict=False)
    m = m.to(DEV).enable_training()
    mel = want["decoder_out"].clone()
    audio = m.vocoder_forward_train(mel=dev(mel), style=dev(cs["style"]), pitch=dev(cs["pitch"]), voiced=dev(env["voiced"]),
                                    noise=dev(cs["noise"]), prior_override=dev(want["prior"]))
    torch.cuda.synchronize()
    ref_fwd = env["ref_audio"]
    mse = ((audio.cpu() - ref_fwd) ** 2).mean().item()
    print(f"\n  training-graph forward: max|err| {(audio.cpu() - ref_fwd).abs().max().item():.3e} mse {mse:.3e}")
    assert mse <= 1e-8
    d_audio = torch.sign(audio) / audio.numel()
    d_mel, d_style = m.vocoder_backward(d_audio)
    torch.cuda.synchronize()
    # oracle gradients
    keys = [
        "generator.basegen.phase_output_real_conv.weight", "generator.basegen.phase_output_real_conv.bias",
        "generator.basegen.phase_final_layer_norm.weight",
        "generator.basegen.phase_convnext.7.pwconv2.weight", "generator.basegen.phase_convnext.7.pwconv2.bias",
        "generator.basegen.phase_convnext.7.grn.gamma", "generator.basegen.phase_convnext.7.grn.beta",
        "generator.basegen.phase_convnext.7.snake", "generator.basegen.phase_convnext.7.pwconv1.weight",
        "generator.basegen.phase_convnext.7.norm.fc.weight", "generator.basegen.phase_convnext.7.dwconv.weight",
        "generator.basegen.phase_convnext.0.dwconv.bias", "generator.basegen.phase_input_conv.weight",
        "generator.basegen.amp_output_conv.weight", "generator.basegen.amp_final_layer_norm.bias",
        "generator.basegen.amp_prior_block.convs1.1.parametrizations.weight.original1",
        "generator.basegen.amp_prior_block.convs1.1.parametrizations.weight.original0",
        "generator.basegen.amp_prior_block.alpha1.1", "generator.basegen.amp_prior_block.adain2.2.fc.weight",
        "generator.basegen.phase_prior_conv.weight",
        "generator.basegen.upconvs.2.weight", "generator.basegen.upconvs.0.bias",
        "generator.basegen.upblocks.1.pwconv1.weight", "generator.basegen.amp_convnext.3.pwconv2.weight",
        "generator.amp_conformer.layers.0.attn.fn.to_q.weight", "generator.amp_conformer.layers.0.attn.fn.to_kv.weight",
        "generator.amp_conformer.layers.0.attn.fn.to_out.bias", "generator.amp_conformer.layers.0.ff1.fn.fn.net.0.weight",
        "generator.amp_conformer.layers.0.conv.net.1.weight", "generator.amp_conformer.layers.0.conv.net.3.conv.weight",
        "generator.amp_conformer.layers.0.conv.net.4.weight", "generator.amp_conformer.layers.0.post_norm.fc.bias",
        "generator.amp_norm.weight", "generator.amp_input_conv.weight",
    ]
    for k in keys:
        P[k].requires_grad_(True)
    mel_r = mel.clone().requires_grad_(True)
    style_r = cs["style"].clone().requires_grad_(True)
    ref = ov.multi_generator(P, "generator", mel_r, style_r, cs["pitch"], env["voiced"], cs["noise"], prior=want["prior"])
    ref.abs().mean().backward()
    rep = Report()
    # conditioning: see tests/test_oracle_golden.py::test_backward_vs_reference (fp64 vs fp32 differ by ~1e-2 on row 1)
    rep.add("d_mel", d_mel, mel_r.grad, 3e-2)
    rep.add("d_style", d_style, style_r.grad, 3e-2)
    named = dict(m.named_parameters())
    for k in keys:
        rep.add("d " + k.replace("generator.", "")[-44:], named[k].grad, P[k].grad, 3e-2)
    rep.done()


def test_layernorm32_backward_one_thread_per_column_equals_the_general_kernel(env, monkeypatch):
    """chan_ln32_bwd_kernel (round 5: the LayerNorms of the vocoder's 75T-rate heads, one thread per column, parameter sums
    through per-workgroup partial rows) against the general three-combine kernel + chan_ln_bwd_param_kernel on the same
    vocoder backward (B = 2, T = 440: B x 75T >= 65536 positions selects it; STY_NO_LN32_BWD=1 the general pair): input
    gradients, the three LayerNorms' weight / bias gradients and everything upstream of them."""
    import stylish_tts_amd as S
    P = {k: v.clone() for k, v in env["P"].items()}
    g = torch.Generator().manual_seed(3)
    B, T = 2, 440
    mel = torch.randn(B, 256, T, generator=g)
    style = torch.randn(B, 64, generator=g)
    pitch = torch.rand(B, T, generator=g) * 200 + 80
    voiced = torch.ones(B, T)
    noise = torch.randn(B, 300 * T, 9, generator=g)
    out = {}
    for mode in ("general", "ln32"):
        if mode == "general":
            monkeypatch.setenv("STY_NO_LN32_BWD", "1")
        else:
            monkeypatch.delenv("STY_NO_LN32_BWD")
        m = S.SpeechPredictor()
        m.load_state_dict(P, strict=False)
        m = m.to(DEV).enable_training()
        audio = m.vocoder_forward_train(mel=dev(mel), style=dev(style), pitch=dev(pitch), voiced=dev(voiced), noise=dev(noise))
        d_mel, d_style = m.vocoder_backward(torch.sign(audio) / audio.numel())
        torch.cuda.synchronize()
        out[mode] = dict(d_mel=d_mel.cpu(), d_style=d_style.cpu(),
                         g={k: p.grad.cpu().clone() for k, p in m.named_parameters() if p.grad is not None})
    rep = Report()
    rep.add("d mel", out["ln32"]["d_mel"], out["general"]["d_mel"], 2e-5)
    rep.add("d style", out["ln32"]["d_style"], out["general"]["d_style"], 2e-5)
    worst, worst_k = 0.0, None
    gmax = max(a.abs().max().item() for a in out["general"]["g"].values())
    for k, a in out["general"]["g"].items():
        if not k.startswith("generator."):
            continue
        b = out["ln32"]["g"][k]
        den = a.abs().max().item()
        if den > 1e-6 * gmax:  # (a conv bias in front of an instance norm has a structurally zero gradient: noise on both sides)
            e = (b - a).abs().max().item() / den
            if e > worst:
                worst, worst_k = e, k
            if "layer_norm" in k or k.endswith("phase_norm.weight") or k.endswith("phase_norm.bias"):
                rep.add("d " + k[-44:], b, a, 2e-5)
    print(f"\n  worst generator parameter gradient, one-thread-per-column vs general: {worst:.2e} ({worst_k})")
    rep.done()
    assert worst <= 2e-4  # (fp32 summation orders differ; behind nine ConvNeXt blocks and two resblocks)


def test_three_source_conv_on_the_persistent_kernel_equals_the_tiled_kernel(env, monkeypatch):
    """phase_input_conv (k = 21 over three concatenated 32-channel sources) as three accumulating conv32p_kernel launches
    forward and three direct input-gradient launches backward (Trainer::conv_cat3, bf16 mode, round 5) against the tiled
    kernel with its 96-channel input gradient + PRO_NONE copies (STY_NO_CAT3=1) in the same vocoder training graph: same bf16
    operands, another fp32 summation order.  In the bf16 mode a 1e-7 difference in front of the eight phase ConvNeXt blocks
    flips bf16 roundings inside them and the phase head amplifies that to ~1e-2 of the audio (measured here:
    audio 7.5e-3, gradients 0.11-0.19 in relative L2), so this A/B can only exclude a WRONG block of the weight (relative L2 ~1);
    the values are pinned by the float64-anchored bf16 gates of tests/test_full_size.py, whose shapes take this path."""
    import stylish_tts_amd as S
    from stylish_tts_amd import lib as L
    monkeypatch.setenv("STY_CONV32P_MIN_TILES", "1")
    P = {k: v.clone() for k, v in env["P"].items()}
    cs, want = env["cs"], env["want"]
    out = {}
    for mode in ("tiled", "cat3"):
        if mode == "tiled":
            monkeypatch.setenv("STY_NO_CAT3", "1")
        else:
            monkeypatch.delenv("STY_NO_CAT3")
        m = S.SpeechPredictor()
        m.load_state_dict(P, strict=False)
        m = m.to(DEV).enable_training().set_train_opts(compute_bf16=True)
        L.load().sty_prof_enable(1)
        audio = m.vocoder_forward_train(mel=dev(want["decoder_out"]), style=dev(cs["style"]), pitch=dev(cs["pitch"]),
                                        voiced=dev(env["voiced"]), noise=dev(cs["noise"]), prior_override=dev(want["prior"]))
        d_mel, d_style = m.vocoder_backward(torch.sign(audio) / audio.numel())
        torch.cuda.synchronize()
        L.load().sty_prof_enable(0)
        n32p = sum(r["launches"] for r in L.prof_report(512) if r["name"].startswith("conv32p"))
        named = dict(m.named_parameters())
        out[mode] = dict(audio=audio.cpu(), d_mel=d_mel.cpu(), d_style=d_style.cpu(), n32p=n32p,
                         gw=named["generator.basegen.phase_input_conv.weight"].grad.cpu().clone(),
                         gb=named["generator.basegen.phase_input_conv.bias"].grad.cpu().clone())
    assert out["cat3"]["n32p"] == out["tiled"]["n32p"] + 6, (out["cat3"]["n32p"], out["tiled"]["n32p"])

    def l2(a, b):
        return ((a - b).norm() / b.norm()).item()
    rows = {k: l2(out["cat3"][k], out["tiled"][k]) for k in ("audio", "gw", "gb", "d_mel", "d_style")}
    print("\n  three launches vs tiled kernel, relative L2: " + "  ".join(f"{k} {v:.2e}" for k, v in rows.items()))
    assert rows["audio"] <= 3e-2 and max(rows["gw"], rows["gb"], rows["d_mel"], rows["d_style"]) <= 0.4, rows


def test_speech_predictor_backward_vs_oracle_and_reference_golden(env):
    """K15: SpeechPredictor backward (text encoder, alignment expand, decoder, vocoder) vs the oracle's autograd and
    vs the gradients the REFERENCE produced (tests/golden/sp_small_grads.safetensors)."""
    import stylish_tts_amd as S
    from oracle import speech_predictor as osp
    from safetensors.torch import load_file
    cs, want, ali = env["cs"], env["want"], env["ali"]
    P = {k: v.clone() for k, v in env["P"].items()}
    m = S.SpeechPredictor()
    m.load_state_dict(P, strict=False)
    m = m.to(DEV).enable_training()
    audio = m.forward_train(dev(cs["texts"]), dev(cs["text_lengths"]), dev(ali), dev(cs["pitch"]), dev(cs["energy"]),
                            dev(env["voiced"]), dev(cs["style"]), dev(cs["pitch"]), noise=dev(cs["noise"]),
                            prior_override=dev(want["prior"]))
    torch.cuda.synchronize()
    mse = ((audio.cpu() - env["ref_audio"]) ** 2).mean().item()
    print(f"\n  training-graph forward: mse {mse:.3e}")
    assert mse <= 1e-8
    d_style, d_energy = m.backward(torch.sign(audio) / audio.numel())
    torch.cuda.synchronize()
    gold = load_file(os.path.join(G, "sp_small_grads.safetensors"))
    keys = [k[len("grad."):] for k in gold if k not in ("grad.style", "grad.energy")]
    extra = ["text_encoder.prenet.conv_layers.1.weight", "text_encoder.prenet.norm_layers.2.gamma",
             "text_encoder.encoder.ffn_layers.3.conv_2.weight", "text_encoder.encoder.norm_layers_2.7.beta",
             "text_encoder.encoder.attn_layers.0.conv_k.bias", "text_encoder.proj_m.weight",
             "decoder.encode.conv1x1.parametrizations.weight.original1", "decoder.decode.0.norm1.fc.weight",
             "decoder.N_conv.parametrizations.weight.original1", "decoder.N_conv.parametrizations.weight.original0",
             "decoder.F0_conv.bias", "decoder.asr_res.0.parametrizations.weight.original1"]
    for k in keys + extra:
        P[k].requires_grad_(True)
    style_r = cs["style"].clone().requires_grad_(True)
    energy_r = cs["energy"].clone().requires_grad_(True)
    ref = osp.speech_predictor(P, cs["texts"], cs["text_lengths"], ali, cs["pitch"], energy_r, env["voiced"], style_r,
                               cs["pitch"], cs["noise"], prior=want["prior"])
    ref.abs().mean().backward()
    rep = Report()
    named = dict(m.named_parameters())
    rep.add("d_style vs oracle", d_style, style_r.grad, 3e-2)
    rep.add("d_energy vs oracle", d_energy, energy_r.grad, 3e-2)
    rep.add("d_style vs REFERENCE", d_style, gold["grad.style"], 3e-2)
    rep.add("d_energy vs REFERENCE", d_energy, gold["grad.energy"], 3e-2)
    for k in keys:
        rep.add("REF d " + k[-40:], named[k].grad, gold["grad." + k], 3e-2)
    for k in keys + extra:
        rep.add("d " + k[-44:], named[k].grad, P[k].grad, 3e-2)
    rep.done()


def test_acoustic_losses_forward_backward():
    """N1 + A10 backward: mel / multi-phase losses and d seed / d audio_pred vs the oracle's autograd."""
    from oracle import losses as ol
    from stylish_tts_amd.losses import acoustic_loss
    gt = _test_audio(2, 24000, 3)
    g = torch.Generator().manual_seed(4)
    pred = (0.8 * gt + 0.05 * torch.randn(2, 24000, generator=g)).requires_grad_(True)
    mel, mph, tot = ol.acoustic_losses(gt, pred)
    tot.backward()
    losses, d = acoustic_loss(dev(gt), dev(pred.detach()))
    torch.cuda.synchronize()
    print(f"\n  mel {losses[0].item():.6f} vs {mel.item():.6f}   multi_phase {losses[1].item():.6f} vs {mph.item():.6f}")
    assert abs(losses[0].item() - mel.item()) <= 1e-5 * abs(mel.item()) + 1e-7
    assert abs(losses[1].item() - mph.item()) <= 1e-4 * abs(mph.item())
    # the loss is piecewise linear (|.| and the wrap): elements within rounding of a kink flip their sign, so
    # compare the gradients in aggregate
    ref = pred.grad
    err = (d.cpu() - ref)
    rel = err.norm().item() / ref.norm().item()
    cos = torch.nn.functional.cosine_similarity(d.cpu().flatten(), ref.flatten(), dim=0).item()
    print(f"  d_audio: relative L2 error {rel:.3e}, cosine {cos:.6f}, max|ref| {ref.abs().max().item():.3e}")
    assert rel <= 2e-2 and cos >= 0.9995


@pytest.mark.parametrize("B,N", [(2, 24000), (3, 156000 + 77)])
def test_fft_front_end_equals_the_gemm_front_end(B, N, tmp_path):
    """The LDS FFT of the mel / loss front ends (stft_fft_kernel and its adjoint) and the band-limited mel filter bank
    (fb_sparse_*_kernel), the defaults, against the folded-DFT and dense filter-bank GEMMs they replace (STY_DFT_GEMM=1,
    STY_FB_GEMM=1), each in a process of its own on the same seeded input: mel spectrograms, log energy, both
    losses and the seed gradient d loss / d audio_pred.  The second case has the benchmark's utterance length plus a ragged
    tail (partial frame tiles at all three resolutions)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for mode in ("fft", "gemm"):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop("STY_DFT_GEMM", None)
        env.pop("STY_FB_GEMM", None)
        if mode == "gemm":
            env["STY_DFT_GEMM"] = "1"  # folded-DFT GEMMs instead of the LDS FFT
            env["STY_FB_GEMM"] = "1"   # dense mel filter-bank GEMMs instead of the band-limited sums
        f = str(tmp_path / f"{mode}.pt")
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "frontend_ab_worker.py"), f, str(B), str(N)],
                           capture_output=True, text=True, env=env, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = torch.load(f)
    a, b = outs["fft"], outs["gemm"]
    rep = Report()
    for k in ("mel512", "mel512_energy", "mel2048", "mel2048_energy"):
        rep.add(k, a[k], b[k], 1e-4)  # the gate of test_mel_front_end against the oracle
    rep.done()
    la, lb = a["losses"], b["losses"]
    print(f"\n  losses fft {la.tolist()}  gemm {lb.tolist()}")
    assert abs(la[0] - lb[0]) <= 1e-5 * abs(lb[0]) and abs(la[1] - lb[1]) <= 1e-4 * abs(lb[1])
    # piecewise-linear losses: elements within rounding of a kink flip sign, compare the gradients in aggregate
    da, db = a["d_pred"].flatten(), b["d_pred"].flatten()
    rel = ((da - db).norm() / db.norm()).item()
    cos = torch.nn.functional.cosine_similarity(da, db, dim=0).item()
    print(f"  d_pred: relative L2 {rel:.3e}  cosine {cos:.7f}")
    assert rel <= 2e-2 and cos >= 0.9995


@pytest.mark.parametrize("bf16", [0, 1])
def test_stem_weight_gradient_streaming_kernel_equals_the_tiled_kernel(tmp_path, bf16):
    """stem_wgrad_kernel (wgrad.hip: the 3 x 3 conv of the one-channel mel image, G read once, sums in registers) against the
    general tiled weight-gradient kernel it replaced (STY_NO_STEM_WGRAD=1), same process input, both modes.  Same operands
    (bf16 mode: G * mask and x rounded to bf16, exact products), fp32 sums in a different order: every element of the stem's
    weight and bias gradient within 2e-6 of the tensor scale (measured: 2e-7); every OTHER gradient of the encoder within 1e-6 of its scale
    (nothing else may change; not bit-equal because the style head's weight gradient is a float-atomic sum over the batch,
    DESIGN.md section 7 item 7)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for mode in ("stream", "tiled"):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop("STY_NO_STEM_WGRAD", None)
        if mode == "tiled":
            env["STY_NO_STEM_WGRAD"] = "1"
        f = str(tmp_path / f"{mode}.pt")
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "stem_wgrad_ab_worker.py"), f, "3", "130", str(bf16)],
                           capture_output=True, text=True, env=env, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = torch.load(f)
    a, b = outs["stream"], outs["tiled"]
    rep = Report()
    rep.add("d stem weight", a["stem_w"], b["stem_w"], 2e-6)
    rep.add("d stem bias", a["stem_b"], b["stem_b"], 2e-6)
    rep.done()
    assert float(b["stem_w"].abs().max()) > 0
    for k in a:
        if k in ("stem_key", "stem_w", "stem_b") or k.startswith("shared.0."):
            continue
        scale = float(b[k].abs().max())
        assert float((a[k] - b[k]).abs().max()) <= 1e-6 * scale, f"{k} changed with the stem's weight-gradient kernel"


@pytest.mark.parametrize("T", [80, 161])
def test_mel_style_encoder_backward(T):
    """A2 backward: gradients of every MelStyleEncoder parameter (through the eval-mode spectral norm) vs the
    oracle's autograd.  T=161 exercises the odd-width column replication of the average-pool shortcut."""
    import stylish_tts_amd as S
    from oracle import style_encoder as ose
    from oracle.manifest import style_encoder_manifest
    from oracle.weights import fill_state_dict
    P = fill_state_dict(style_encoder_manifest(), 0)
    m = S.MelStyleEncoder()
    m.load_state_dict(P, strict=False)
    m = m.to(DEV).enable_training()
    g = torch.Generator().manual_seed(T)
    x = torch.randn(2, 1, 80, T, generator=g) * 0.8 - 0.3
    cot = torch.randn(2, 64, generator=g)
    out = m.forward_train(dev(x))
    m.backward(dev(cot))
    torch.cuda.synchronize()
    Pr = {k: v.clone() for k, v in P.items()}
    names = [k for k, _ in m.named_parameters()]
    for k in names:
        Pr[k].requires_grad_(True)
    ref = ose.mel_style_encoder(Pr, "", x)
    (ref * cot).sum().backward()
    rep = Report()
    rep.add("style (training graph)", out, ref.detach(), 1e-5)
    named = dict(m.named_parameters())
    for k in names:
        rep.add("d " + k[-44:], named[k].grad, Pr[k].grad, 2e-3)
    rep.done()


@pytest.mark.parametrize("W", [61, 62, 63, 80, 130])
def test_mel_style_encoder_block_taps_forward_and_gradient(W):
    """A2 per block, element by element: the stem's output, the four ResBlk outputs, the head conv's output and the input
    of every ResBlk's second LeakyReLU of the style encoder's training graph, and d loss / d each of the first six after
    the backward, against the oracle (autograd with retained gradients).  The widths straddle the flat-image row pitch at
    which a buffer load with a negative lane offset lost two samples per channel row: the pooled
    [B,64] style vector averages such an error away, a per-element gate on the activations does not.
    Forward: every element within 1e-5 of the tensor scale.  Gradients: EVERY element within 1e-4, no exceptions.
    What made an exception necessary before (round 3): among the ~3 M LeakyReLU inputs of a run a few dozen lie within
    fp32 rounding of zero, the two implementations take different slopes (1 vs 0.2) for such an element and the
    difference spreads over the receptive field behind it.  Now the ORACLE is told the slope: at its nine spatial
    LeakyReLU sites it takes, for exactly the elements whose own pre-activation lies within the forward gate of zero
    (|pre| <= 1e-5 max|pre|: the only ones the forward gate lets the two sides decide differently), the sign the HIP run
    saw (taps 0..4 and 6..9), its own sign everywhere else.  The forward value moves by <= 0.8e-5 of the scale for those
    elements; the gradients must then agree everywhere."""
    import stylish_tts_amd as S
    from oracle import style_encoder as ose
    from oracle.manifest import style_encoder_manifest
    from oracle.weights import fill_state_dict
    P = fill_state_dict(style_encoder_manifest(), 0)
    m = S.MelStyleEncoder()
    m.load_state_dict(P, strict=False)
    m = m.to(DEV).enable_training()
    g = torch.Generator().manual_seed(W)
    x = torch.randn(2, 1, 80, W, generator=g) * 0.8 - 0.3
    cot = torch.randn(2, 64, generator=g)
    out = m.forward_train(dev(x))
    acts = [m.tap(i).cpu() for i in range(10)]
    m.backward(dev(cot))
    grads = [m.tap(i, grad=True) for i in range(6)]
    torch.cuda.synchronize()
    hip_pre = {"head.pre": acts[4]}
    for j in range(1, 5):
        hip_pre[f"shared.{j}.pre1"] = acts[j - 1]
        hip_pre[f"shared.{j}.pre2"] = acts[5 + j]
    overridden = []

    def lrelu(site, t):
        td = t.detach()
        pos = td > 0
        if site in hip_pre:
            assert hip_pre[site].shape == td.shape, (site, hip_pre[site].shape, td.shape)
            amb = td.abs() <= 1e-5 * td.abs().max()
            flips = amb & ((hip_pre[site] > 0) != pos)
            if flips.any():
                overridden.append((site, int(amb.sum()), int(flips.sum())))
            pos = torch.where(amb, hip_pre[site] > 0, pos)
        return torch.where(pos, t, 0.2 * t)

    want = {}
    Pr = {k: v.clone() for k, v in P.items()}
    Pr["shared.0.bias"].requires_grad_(True)  # (any parameter in front of the first tap: the graph needs a leaf)
    ref = ose.mel_style_encoder(Pr, "", x, want, lrelu)
    names = [f"se.block{i}" for i in range(5)] + ["se.head"]
    for k in names:
        want[k].retain_grad()
    (ref * cot).sum().backward()
    print(f"\n  W={W}: LeakyReLU sites where the HIP run's slope was taken (site, elements within the gate of zero, of those "
          f"decided differently): {overridden if overridden else 'none'}")
    rep = Report()
    rep.add("style", out, ref.detach(), 1e-5)
    for j in range(1, 5):
        rep.add(f"se.block{j}.pre2 W={W}", acts[5 + j], want[f"shared.{j}.pre2"].detach(), 1e-5)
    for i, k in enumerate(names):
        r = want[k].detach()
        a, ga = acts[i], grads[i]
        if k == "se.head":  # computed at every position, valid where the 5 x 5 window fits
            a, ga = a[:, :, :r.shape[2], :r.shape[3]], ga[:, :, :r.shape[2], :r.shape[3]]
        assert a.shape == r.shape, (k, a.shape, r.shape)
        rep.add(f"{k} W={W}", a, r, 1e-5)
        rep.add(f"d {k} W={W}", ga, want[k].grad, 1e-4)
    rep.done()


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_style_encoder_deferred_gates_equal_the_separate_passes(compute, monkeypatch):
    """The style encoder's backward with the LeakyReLU gates deferred into avgpool2_bwd_kernel / dwconv2d_s2_bwd_kernel and
    the no-prologue input gradients written straight into the gradient buffers against the same backward
    with a pro_bwd_kernel pass after every input-gradient conv (STY_NO_DEFERRED_GATE=1): the same products and sums in the
    same order, so every tap gradient and every parameter gradient must agree to fp32 rounding (the head's Linear weight
    gradient is summed over the batch with float atomics in both runs: 1e-5)."""
    import stylish_tts_amd as S
    from oracle.manifest import style_encoder_manifest
    from oracle.weights import fill_state_dict
    P = fill_state_dict(style_encoder_manifest(), 0)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 1, 80, 84, generator=g) * 0.8 - 0.3
    cot = torch.randn(3, 64, generator=g)
    runs = []
    for plain in (False, True):
        if plain:
            monkeypatch.setenv("STY_NO_DEFERRED_GATE", "1")
        else:
            monkeypatch.delenv("STY_NO_DEFERRED_GATE", raising=False)
        m = S.MelStyleEncoder()
        m.load_state_dict(P, strict=False)
        m = m.to(DEV).enable_training()
        m.set_train_opts(compute_bf16=(compute == "bf16"))
        m.forward_train(dev(x))
        m.backward(dev(cot))
        taps = [m.tap(i, grad=True).cpu() for i in range(6)]
        grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.grad is not None}
        torch.cuda.synchronize()
        runs.append((taps, grads))
    (ta, ga), (tb, gb) = runs
    rep = Report()
    for i, (a, b) in enumerate(zip(ta, tb)):
        rep.add(f"d tap {i}", a, b, 1e-6)
    assert ga.keys() == gb.keys() and len(ga) > 20
    for k in ga:
        rep.add(f"d {k}", ga[k], gb[k], 1e-5 if k.startswith("unshared") or "fc" in k else 2e-6)
    rep.done()


@pytest.mark.parametrize("W", [84, 131, 62])
def test_style_encoder_operand_twins_equal_the_fp32_operand_path(W, monkeypatch):
    """The bf16 compute mode of the style encoder with its GEMM operands stored in HBM as bf16 TWINS (ConvArgs::x16 / g16 /
    y16: written by the producing kernel's output stage -- stem, convp16_kernel, learned down-sampling, pooling and their
    backward kernels -- or by the cast pass; read by convp16_kernel with two-byte loads and by wgradb16_kernel with no
    conversion) against the same mode with fp32 operands converted on the way into LDS (STY_NO_TWINS=1): a twin holds
    exactly the value the fp32 path rounds to, so every forward tap and every gradient tap must agree BIT FOR BIT, and every
    weight gradient up to the fp32 summation order of the two kernels' different splits (2e-6; the head's Linear weight
    gradient is a float-atomic sum in both runs; conv BIAS gradients are row sums of the rounded twin in one run and of the
    unrounded gradient in the other: 2^-8) -- a stale twin (a
    gradient buffer written again after its twin was taken), a wrong mask or a wrong row end shows up here.  Widths: even
    row pitch, odd width (the pooled rows replicate the last column), and the width at which an image row holds fewer
    than one 64-column group.  convp16 / wgradb16 are forced onto the small shapes and checked to have run."""
    import stylish_tts_amd as S
    from oracle.manifest import style_encoder_manifest
    from oracle.weights import fill_state_dict
    from stylish_tts_amd import lib as L
    lib = L.load()
    monkeypatch.setenv("STY_CONVP16_MIN_TILES", "1")
    P = fill_state_dict(style_encoder_manifest(), 0)
    g = torch.Generator().manual_seed(W)
    x = torch.randn(3, 1, 80, W, generator=g) * 0.8 - 0.3
    cot = torch.randn(3, 64, generator=g)
    runs = []
    for twins in (False, True):
        if twins:
            monkeypatch.delenv("STY_NO_TWINS", raising=False)
        else:
            monkeypatch.setenv("STY_NO_TWINS", "1")
        m = S.MelStyleEncoder()
        m.load_state_dict(P, strict=False)
        m = m.to(DEV).enable_training()
        m.set_train_opts(compute_bf16=True)
        L.prof_report(512)
        lib.sty_prof_enable(1)
        try:
            out = m.forward_train(dev(x))
            acts = [m.tap(i).cpu() for i in range(10)]
            m.backward(dev(cot))
            torch.cuda.synchronize()
        finally:
            lib.sty_prof_enable(0)
        names = {r["name"]: r["launches"] for r in L.prof_report(512)}
        taps = [m.tap(i, grad=True).cpu() for i in range(6)]
        grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.grad is not None}
        torch.cuda.synchronize()
        runs.append((out.cpu(), acts, taps, grads, names))
    (oa, aa, ta, ga, na), (ob, ab, tb, gb, nb) = runs
    print(f"\n  W={W}: kernels with twins: " + ", ".join(f"{k} x{v}" for k, v in sorted(nb.items()) if "16" in k or "twin" in k))
    assert not any(k.startswith("wgradb16") or k.startswith("twin_cast") for k in na), na
    assert any(k.startswith("wgradb16") for k in nb) and any(k.startswith("convp16") for k in nb), nb
    assert torch.equal(oa, ob)
    for i, (a, b) in enumerate(zip(aa, ab)):
        assert torch.equal(a, b), f"activation tap {i} differs: max {(a - b).abs().max().item():.3e}"
    for i, (a, b) in enumerate(zip(ta, tb)):
        assert torch.equal(a, b), f"gradient tap {i} differs: max {(a - b).abs().max().item():.3e}"
    assert ga.keys() == gb.keys() and len(ga) > 20
    for k in ga:
        if k.startswith("unshared"):
            assert rel_err(gb[k], ga[k]) <= 1e-6, k
        elif k.endswith(".bias"):
            # the one place where the two runs see different numbers: with twins the bias gradient is the row sum of the
            # bf16 twin of G (wgradb16_kernel: an MFMA against ones), without them wgradb_kernel sums G before rounding
            assert rel_err(gb[k], ga[k]) <= 4e-3, (k, rel_err(gb[k], ga[k]))
        else:
            # same rounded operands, but wgradb16_kernel and wgradb_kernel walk the (batch, time) chunks in different
            # splits: fp32 summation order (measured 2e-7)
            assert rel_err(gb[k], ga[k]) <= 2e-6, f"d {k} differs: {rel_err(gb[k], ga[k]):.3e}"


def test_adamw_matches_torch():
    """sty_adamw_step on a flat bucket vs torch.optim.AdamW (the reference's optimizer, optimizers.py:110-118)."""
    from stylish_tts_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(5)
    shapes = [(33, 7), (129,), (4, 5, 3), (1,), (1000,)]
    ref_p = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    hip_p = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref_p]
    ref = torch.optim.AdamW(ref_p, lr=1e-3, weight_decay=1e-4, betas=(0.85, 0.99), eps=1e-9)
    opt = FlatAdamW(hip_p, lr=1e-3, weight_decay=1e-4, betas=(0.85, 0.99), eps=1e-9, bucket_bytes=4096)
    assert len(opt.grads.buckets) > 1
    for it in range(4):
        for pr, ph in zip(ref_p, hip_p):
            gr = torch.randn(pr.shape, generator=g) * (10.0 ** (it - 2))
            pr.grad = gr.clone()
            ph.grad.copy_(gr)
        ref.step()
        opt.step()
    torch.cuda.synchronize()
    for pr, ph in zip(ref_p, hip_p):
        err = (ph.detach().cpu() - pr.detach()).abs().max().item()
        assert err <= 2e-6, err


def _train_setup(env, lr, train_mode=False, compute="fp32"):
    import stylish_tts_amd as S
    from oracle.manifest import style_encoder_manifest
    from oracle.weights import fill_state_dict
    from stylish_tts_amd.acoustic import AcousticTrainer
    P = {k: v.clone() for k, v in env["P"].items()}
    Pse = fill_state_dict(style_encoder_manifest(), 0)
    sp = S.SpeechPredictor()
    sp.load_state_dict(P, strict=False)
    se = S.MelStyleEncoder()
    se.load_state_dict(Pse)
    tr = AcousticTrainer(sp.to(DEV), se.to(DEV), lr=lr, train_mode=train_mode, compute=compute)
    return tr, P, Pse


def test_acoustic_train_step_gradients(env):
    """A0 forward+backward assembled (mel -> style encoder -> predictor -> losses -> backward of both models):
    losses and parameter gradients of ONE train_acoustic step (lr = 0) vs the oracle's autograd."""
    from oracle import losses as ol, speech_predictor as osp
    cs = env["cs"]
    B, T = cs["pitch"].shape
    audio_gt = _test_audio(B, 300 * T, 21)
    tr, P, Pse = _train_setup(env, 0.0)
    want = {}
    sp_keys = ["generator.basegen.amp_output_conv.weight", "generator.basegen.phase_output_real_conv.bias",
               "decoder.decode.0.norm1.fc.weight", "text_encoder.proj_m.weight", "text_encoder.emb.weight",
               "generator.conformer.ff1.1.weight" if "generator.conformer.ff1.1.weight" in P else None]
    sp_keys = [k for k in sp_keys if k and k in P and P[k].is_floating_point()]
    se_keys = ["shared.0.weight_orig", "shared.2.conv1.weight_orig", "shared.4.conv2.bias", "unshared.weight"]
    for k in sp_keys:
        P[k].requires_grad_(True)
    for k in se_keys:
        Pse[k].requires_grad_(True)
    ref = osp.acoustic_forward(P, Pse, audio_gt, cs["texts"], cs["text_lengths"], cs["pitch"], cs["durations"],
                               cs["noise"], want)
    mel, mph, tot = ol.acoustic_losses(audio_gt, ref.squeeze(1))
    tot.backward()
    losses = tr.train_batch(audio_gt=dev(audio_gt), texts=dev(cs["texts"]), text_lengths=dev(cs["text_lengths"]),
                            pitch=dev(cs["pitch"]), durations=dev(cs["durations"]), noise=dev(cs["noise"]),
                            prior_override=dev(want["prior"]))
    torch.cuda.synchronize()
    print(f"\n  mel {losses[0].item():.6f} vs {mel.item():.6f}   multi_phase {losses[1].item():.6f} vs {mph.item():.6f}")
    assert abs(losses[0].item() - mel.item()) <= 1e-4 * abs(mel.item())
    assert abs(losses[1].item() - mph.item()) <= 1e-3 * abs(mph.item())
    rep = Report()
    nsp, nse = dict(tr.sp.named_parameters()), dict(tr.se.named_parameters())
    for k in sp_keys:
        rep.add("d " + k[-44:], nsp[k].grad, P[k].grad, 5e-2)
        rep.add("p " + k[-44:], nsp[k], P[k].detach(), 1e-7)  # lr = 0: parameters untouched
    for k in se_keys:
        rep.add("d se." + k[-40:], nse[k].grad, Pse[k].grad, 5e-2)
    rep.done()


def test_acoustic_train_step_with_spectrogram_discriminators(env):
    """train_acoustic + the discriminator step of stage.py:124-146 with the three spectrogram discriminators:
    the mel / phase losses are those of the plain step; the generator's gradients change by the adversarial term; only
    mrd{disc_index} is stepped, with lr = generator lr x the helper's multiplier and gradients scaled by sqrt(batch);
    the helpers' tracked losses follow losses.py:288; the waveform discriminator (weight 3) is stepped every batch."""
    from safetensors.torch import load_file
    from stylish_tts_amd.acoustic import AcousticTrainer
    from stylish_tts_amd.discriminators import SpecDiscriminator
    import stylish_tts_amd as S
    cs = env["cs"]
    B, T = cs["pitch"].shape
    audio_gt = _test_audio(B, 300 * T, 21)
    fx = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "disc_small.safetensors"))
    dparams = {k[2:]: v for k, v in fx.items() if k.startswith("w.")}
    tr0, _, _ = _train_setup(env, 1e-4)
    kw = dict(audio_gt=dev(audio_gt), texts=dev(cs["texts"]), text_lengths=dev(cs["text_lengths"]), pitch=dev(cs["pitch"]),
              durations=dev(cs["durations"]), noise=dev(cs["noise"]), seed=5)
    l0 = tr0.train_batch(**kw).cpu()
    g0 = {k: p.grad.detach().cpu().clone() for k, p in tr0.sp.named_parameters() if p.grad is not None}
    mrd = []
    for r in range(3):
        m = SpecDiscriminator().to(DEV)
        m.load_state_dict(dparams)
        mrd.append(m)
    P = {k: v.clone() for k, v in env["P"].items()}
    from oracle.manifest import style_encoder_manifest
    from oracle.weights import fill_state_dict
    sp = S.SpeechPredictor()
    sp.load_state_dict(P, strict=False)
    se = S.MelStyleEncoder()
    se.load_state_dict(fill_state_dict(style_encoder_manifest(), 0))
    from stylish_tts_amd.discriminators import ContextFreeDiscriminator
    cfx = load_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfdisc_small.safetensors"))
    wave_disc = ContextFreeDiscriminator()
    wave_disc.load_state_dict({k[2:]: v for k, v in cfx.items() if k.startswith("w.")})
    wave_disc = wave_disc.to(DEV)
    wd_before = {k: p.detach().cpu().clone() for k, p in wave_disc.named_parameters()}
    tr = AcousticTrainer(sp.to(DEV), se.to(DEV), lr=1e-4, train_mode=False, mrd=mrd, w_gen=1.0, disc=wave_disc)
    before = [{k: p.detach().cpu().clone() for k, p in m.named_parameters()} for m in mrd]
    l1 = tr.train_batch(disc_index=2, **kw).cpu()
    torch.cuda.synchronize()
    gan = tr.gan.cpu()
    print(f"\n  generator {gan[0].item():.4f}  discriminator losses {[round(gan[1 + 2 * r].item(), 4) for r in range(3)]}")
    assert torch.allclose(l0, l1, rtol=1e-5, atol=0)
    assert torch.isfinite(gan).all() and gan[0].item() > 0
    g1 = {k: p.grad.detach().cpu().clone() for k, p in tr.sp.named_parameters() if p.grad is not None}
    k = "generator.basegen.amp_output_conv.weight"
    assert (g1[k] - g0[k]).abs().max().item() > 1e-6 * g0[k].abs().max().item()  # the adversarial term arrived
    for r, m in enumerate(mrd):
        moved = max((p.detach().cpu() - before[r][n]).abs().max().item() for n, p in m.named_parameters())
        assert (moved > 0) == (r == 2), (r, moved)
        assert abs(tr.disc_helpers[r].last_loss - (2.5 * 0.95 + gan[2 + 2 * r].item() * 0.05)) <= 1e-5
    gw = tr.gan_wave.cpu()
    assert torch.isfinite(gw).all() and gw[0].item() > 0 and gw[1].item() > 0
    assert all((p.detach().cpu() - wd_before[k]).abs().max().item() > 0 for k, p in wave_disc.named_parameters())
    assert abs(tr.disc_helper.last_loss - (0.5 * 0.95 + gw[2].item() * 0.05)) <= 1e-5
    assert int(dict(wave_disc.named_buffers())["conv.0.net.1.num_batches_tracked"].item()) == 2
    # first AdamW step: |delta| = lr (1 + weight decay shrink) for every element with a non-zero gradient
    lr_d = 1e-4 * 1.0  # tracked loss == ideal loss before the first step -> multiplier 1 (losses.py:241-256)
    p2 = dict(mrd[2].named_parameters())["discriminators.1.parametrizations.weight.original1"].detach().cpu()
    b2 = before[2]["discriminators.1.parametrizations.weight.original1"]
    step = (p2 - b2 * (1 - lr_d * 1e-4)).abs()
    assert abs(step.median().item() - lr_d) <= 0.02 * lr_d, step.median().item()


def test_acoustic_train_step_bf16_compute_vs_fp32(env):
    """Config c3's compute mode (bf16 operands on the dense convs / Linears, fp32 accumulation) on a whole training
    step, REPORTED against the fp32 step ("bf16 reported, not gated").  The kernels of the mode
    are pinned exactly by test_dense_conv1d_vs_torch[bf16]; what is left here is how far operand rounding moves the
    losses and gradients.  The gradients of the randomly initialised model are ill-conditioned (phase of small
    magnitudes): a relative weight perturbation of bf16 size (+-2^-9) in the pure fp32 step already moves the
    per-tensor gradients by ~0.6 in relative L2.  So the yardstick is that control run: the bf16 step (which rounds
    activations too) must stay within 2x of the control's median deviation and 0.1 of its total-gradient cosine, and
    the losses within 2 %."""
    cs = env["cs"]
    B, T = cs["pitch"].shape
    audio_gt = _test_audio(B, 300 * T, 21)
    out = {}
    for mode in ("fp32", "bf16", "control"):
        tr, _, _ = _train_setup(env, 0.0, compute="bf16" if mode == "bf16" else "fp32")
        if mode == "control":
            g = torch.Generator().manual_seed(3)
            with torch.no_grad():
                for p in list(tr.sp.parameters()) + list(tr.se.parameters()):
                    if p.is_floating_point():
                        p.mul_(1 + (torch.rand(p.shape, generator=g).to(p.device) - 0.5) * 2.0 ** -8)
        losses = tr.train_batch(audio_gt=dev(audio_gt), texts=dev(cs["texts"]), text_lengths=dev(cs["text_lengths"]),
                                pitch=dev(cs["pitch"]), durations=dev(cs["durations"]), noise=dev(cs["noise"]), seed=5)
        torch.cuda.synchronize()
        out[mode] = (losses.cpu(), {k: p.grad.detach().float().cpu().clone() for k, p in tr.sp.named_parameters()
                                    if p.grad is not None})
    l32, g32 = out["fp32"]

    def deviation(mode):
        l, gr = out[mode]
        rels = sorted((a - gr[k]).norm().item() / a.norm().item() for k, a in g32.items()
                      if a.norm().item() >= 1e-6 and a.numel() >= 64)
        a = torch.cat([v.flatten() for v in g32.values()])
        b = torch.cat([gr[k].flatten() for k in g32])
        cos = (a @ b).item() / (a.norm().item() * b.norm().item())
        print(f"  {mode:8s} losses {l.tolist()}  per-tensor relative L2 gradient difference: median "
              f"{rels[len(rels) // 2]:.4f}, 90th percentile {rels[int(0.9 * len(rels))]:.4f}, max {rels[-1]:.4f};"
              f" total-gradient cosine {cos:.5f}")
        return l, rels[len(rels) // 2], rels[0], cos

    print(f"\n  fp32     losses {l32.tolist()}")
    l16, med16, min16, cos16 = deviation("bf16")
    _, medc, _, cosc = deviation("control")
    assert ((l16 - l32).abs() <= 2e-2 * l32.abs()).all()
    assert min16 > 0                      # the bf16 kernels ran
    assert med16 <= 2.0 * medc and cos16 >= cosc - 0.1


def test_multi_stream_step_equals_single_stream_step(env):
    """The four-stream training step (weight-gradient streams, style encoder beside the text encoder) against the same
    step with every internal stream off (sty_set_single_stream): identical losses and gradients up to the one
    atomically accumulated tensor (pool_fc) -- a race between the streams would show up here."""
    from stylish_tts_amd import lib as L
    lib = L.load()
    cs = env["cs"]
    B, T = cs["pitch"].shape
    audio_gt = _test_audio(B, 300 * T, 21)
    res = []
    try:
        for single in (0, 1, 0):
            lib.sty_set_single_stream(single)
            tr, _, _ = _train_setup(env, 0.0)
            tr.single_stream = bool(single)
            losses = tr.train_batch(audio_gt=dev(audio_gt), texts=dev(cs["texts"]), text_lengths=dev(cs["text_lengths"]),
                                    pitch=dev(cs["pitch"]), durations=dev(cs["durations"]), noise=dev(cs["noise"]), seed=5)
            torch.cuda.synchronize()
            g = torch.cat([p.grad.detach().flatten().cpu() for m in (tr.sp, tr.se) for p in m.parameters()
                           if p.grad is not None])
            res.append((losses.cpu(), g))
    finally:
        lib.sty_set_single_stream(0)
    for i in (0, 2):
        rel = ((res[i][1] - res[1][1]).norm() / res[1][1].norm()).item()
        print(f"\n  multi-stream run {i} vs single-stream: gradient relative L2 {rel:.3e}, losses {res[i][0].tolist()} / {res[1][0].tolist()}")
        assert rel <= 1e-6 and torch.allclose(res[i][0], res[1][0], rtol=1e-6, atol=0)


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_grouped_weight_gradient_reduction_equals_the_per_launch_reductions(env, compute, monkeypatch):
    """The deferred, grouped reduction of the weight-gradient partial sums (wgrad.hip: one launch over a device-side job
    table per gradient segment instead of one reduction launch behind every weight-gradient kernel; every kernel gets a
    partial buffer of its own) against the per-launch reductions (STY_NO_DEFERRED_REDUCE=1): the same sums in the same
    order, so losses and every parameter gradient must agree to the bit -- except the one atomically accumulated tensor
    (pool_fc), hence 1e-6 on the whole vector.  Single-stream and four-stream steps, two steps each (the second step
    re-uses the cached job tables)."""
    from stylish_tts_amd import lib as L
    lib = L.load()
    cs = env["cs"]
    B, T = cs["pitch"].shape
    audio_gt = _test_audio(B, 300 * T, 21)
    res = {}
    try:
        for single in (1, 0):
            lib.sty_set_single_stream(single)
            for plain in (True, False):
                if plain:
                    monkeypatch.setenv("STY_NO_DEFERRED_REDUCE", "1")
                else:
                    monkeypatch.delenv("STY_NO_DEFERRED_REDUCE", raising=False)
                tr, _, _ = _train_setup(env, 0.0, compute=compute)
                tr.single_stream = bool(single)
                for it in range(2):
                    losses = tr.train_batch(audio_gt=dev(audio_gt), texts=dev(cs["texts"]), text_lengths=dev(cs["text_lengths"]),
                                            pitch=dev(cs["pitch"]), durations=dev(cs["durations"]), noise=dev(cs["noise"]), seed=5)
                torch.cuda.synchronize()
                g = torch.cat([p.grad.detach().flatten().cpu() for m in (tr.sp, tr.se) for p in m.parameters()
                               if p.grad is not None])
                assert bool(torch.isfinite(g).all())
                res[(single, plain)] = (losses.cpu(), g)
    finally:
        lib.sty_set_single_stream(0)
    for single in (1, 0):
        a, b = res[(single, True)], res[(single, False)]
        rel = ((a[1] - b[1]).norm() / a[1].norm()).item()
        same = (a[1] == b[1]).float().mean().item()
        print(f"\n  {'single' if single else 'four'}-stream, {compute}: grouped vs per-launch reductions: gradient relative L2 "
              f"{rel:.3e}, {100 * same:.3f} % of the elements bit-equal, losses {a[0].tolist()} / {b[0].tolist()}")
        # the compared gradients are those of the SECOND step: the first step's float-atomic sum (pool_fc) differs in the last
        # bit between any two runs, AdamW carries that into the parameters, and in the bf16 mode a last-bit difference of an
        # operand flips bf16 roundings downstream (measured 0 ... 4e-6 between runs of the same configuration)
        assert rel <= (3e-5 if compute == "bf16" else 1e-6) and torch.allclose(a[0], b[0], rtol=1e-6, atol=0)


def test_acoustic_training_reduces_loss(env):
    """A few optimizer steps on one fixed batch lower both losses (forward, backward, AdamW and the weight
    re-preparation between steps all act on the same parameters)."""
    cs = env["cs"]
    B, T = cs["pitch"].shape
    audio_gt = _test_audio(B, 300 * T, 21)
    tr, _, _ = _train_setup(env, 2e-4, train_mode=True)
    hist = []
    for it in range(6):
        losses = tr.train_batch(audio_gt=dev(audio_gt), texts=dev(cs["texts"]), text_lengths=dev(cs["text_lengths"]),
                                pitch=dev(cs["pitch"]), durations=dev(cs["durations"]), noise=dev(cs["noise"]))
        hist.append(losses.cpu().tolist())
    print("\n  (mel, multi_phase) per step: " + "  ".join(f"({a:.4f},{b:.4f})" for a, b in hist))
    assert all(torch.isfinite(torch.tensor(hist)).flatten().tolist())
    assert hist[-1][0] < hist[0][0]


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 37, 80, 5, 2, 203), (3, 128, 130, 1, 1, 64), (1, 33, 32, 21, 1, 1500),
                                   (2, 512, 64, 3, 1, 37), (2, 32, 32, 11, 3, 700), (2, 96, 200, 3, 1, 300)])
def test_dense_conv1d_vs_torch(shape, compute):
    """The MFMA implicit-GEMM conv every dense layer runs on, its input gradient (same kernel, flipped weights) and
    the weight-gradient kernels (general K, 64x64 blocked K <= 5, K = 1) vs torch.nn.functional.conv1d in float64.
    compute = bf16 (config c3): the reference multiplies the SAME bf16-rounded operands, so the bound stays an fp32
    accumulation bound -- the rounding itself is the mode's definition, not an error."""
    import ctypes as C
    from stylish_tts_amd import lib as L
    lib = L.load()
    B, Ci, Co, K, d, T = shape
    bf = compute == "bf16"
    g = torch.Generator().manual_seed(sum(shape))
    x, w, b = torch.randn(B, Ci, T, generator=g), torch.randn(Co, Ci, K, generator=g) / (Ci * K) ** 0.5, torch.randn(Co, generator=g)
    gy = torch.randn(B, Co, T, generator=g)
    rnd = (lambda t: t.bfloat16().double()) if bf else (lambda t: t.double())
    pad = (K - 1) * d // 2
    ref = torch.nn.functional.conv1d(rnd(x), rnd(w), b.double(), padding=pad, dilation=d).float()
    xr, wr, gr = rnd(x).requires_grad_(True), rnd(w).requires_grad_(True), rnd(gy)
    (torch.nn.functional.conv1d(xr, wr, None, padding=pad, dilation=d) * gr).sum().backward()
    ref_dw, ref_dx, ref_db = wr.grad.float(), xr.grad.float(), gy.double().sum((0, 2)).float()
    xd, wd, bd, gd = dev(x), dev(w), dev(b), dev(gy)
    y = torch.empty(B, Co, T, device=DEV)
    need = C.c_size_t()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.sty_conv1d_workspace_bytes(Co, Ci, K, C.byref(need)))
    ws = torch.empty(need.value, dtype=torch.uint8, device=DEV)
    L.check(lib.sty_conv1d_fwd(B, Ci, Co, K, d, T, L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(y), L.ptr(ws), ws.numel(),
                               int(bf), st))
    L.check(lib.sty_conv1d_bwd_workspace_bytes(B, Ci, Co, K, T, C.byref(need)))
    ws2 = torch.empty(need.value, dtype=torch.uint8, device=DEV)
    dw, db, dx = torch.empty(Co, Ci, K, device=DEV), torch.empty(Co, device=DEV), torch.empty(B, Ci, T, device=DEV)
    want_db = K <= 12
    L.check(lib.sty_conv1d_bwd(B, Ci, Co, K, d, T, L.ptr(xd), L.ptr(wd), L.ptr(gd), L.ptr(dw),
                               L.ptr(db) if want_db else None, L.ptr(dx), L.ptr(ws2), ws2.numel(), int(bf), st))
    torch.cuda.synchronize()
    err = (y.cpu() - ref).abs().max().item()
    e_dw = (dw.cpu() - ref_dw).abs().max().item() / ref_dw.abs().max().item()
    e_dx = (dx.cpu() - ref_dx).abs().max().item() / ref_dx.abs().max().item()
    e_db = (db.cpu() - ref_db).abs().max().item() / ref_db.abs().max().item() if want_db else 0.0
    print(f"\n  conv1d {shape} {compute}: max|err| y {err:.3e}  dw {e_dw:.3e}  dx {e_dx:.3e}  dbias {e_db:.3e} (relative)")
    assert err <= 2e-5 and e_dw <= 2e-5 and e_dx <= 2e-5 and e_db <= 2e-5
    if bf:  # the bf16 kernels really ran: the result differs from the exact fp32 product
        exact = torch.nn.functional.conv1d(x.double(), w.double(), b.double(), padding=pad, dilation=d).float()
        assert (y.cpu() - exact).abs().max().item() > 1e-4


@pytest.mark.parametrize("shape", [(3, 128, 130, 1, 1, 64), (2, 512, 64, 3, 1, 38), (2, 96, 200, 3, 1, 300), (2, 64, 96, 3, 1, 1000),
                                   (3, 130, 33, 5, 1, 258), (48, 240, 80, 3, 1, 1500), (2, 64, 64, 3, 1, 1001),
                                   # round 6, convq_kernel (K = 1 / 3, T % 4 == 0): tile edges (T = 256 k +- 4), Cout 80 / 160 / 384
                                   # on 96-cout tiles, channel counts that are no multiple of 8 / 32, one tile, many chunks
                                   (2, 160, 160, 3, 1, 2620), (3, 130, 84, 3, 1, 252), (2, 72, 384, 3, 1, 260), (1, 64, 80, 1, 1, 516),
                                   (2, 1152, 96, 3, 1, 256), (5, 80, 97, 1, 1, 1028)])
def test_persistent_conv16_on_bf16_operand_twins_vs_torch(shape, monkeypatch):
    """convp16_kernel reading its input as a bf16 operand twin (ConvArgs::x16: two-byte loads, no conversion, no prologue)
    and writing the twin of its OUTPUT from the output stage (ConvArgs::y16, here LeakyReLU(0.2)(y) rounded to bf16): forward
    and input gradient through the unit entry points with compute_bf16 = 2 against float64 on the same rounded operands
    (the gates of test_persistent_conv16_vs_torch), against the fp32-operand kernel (compute_bf16 = 1: the same products in
    the same order, so bit-equal), and the output twin against bf16(lrelu(y)) of the fp32 output it was written beside
    (bit-equal, an odd row length included: the row-end lanes store sample by sample)."""
    import ctypes as C
    from stylish_tts_amd import lib as L
    lib = L.load()
    monkeypatch.setenv("STY_CONVP16_MIN_TILES", "1")
    monkeypatch.setenv("STY_CONVQ_MIN_TILES", "1")  # (the twin path's kernel since round 6 where K = 1 / 3 and T % 4 == 0)
    B, Ci, Co, K, d, T = shape
    g = torch.Generator().manual_seed(sum(shape))
    x, w, b = torch.randn(B, Ci, T, generator=g), torch.randn(Co, Ci, K, generator=g) / (Ci * K) ** 0.5, torch.randn(Co, generator=g)
    gy = torch.randn(B, Co, T, generator=g)
    rnd = lambda t: t.bfloat16().double()
    pad = (K - 1) * d // 2
    ref = torch.nn.functional.conv1d(rnd(x), rnd(w), b.double(), padding=pad, dilation=d).float()
    xr, wr, gr = rnd(x).requires_grad_(True), rnd(w).requires_grad_(True), rnd(gy)
    (torch.nn.functional.conv1d(xr, wr, None, padding=pad, dilation=d) * gr).sum().backward()
    ref_dx = xr.grad.float()
    xd, wd, bd, gd = dev(x), dev(w), dev(b), dev(gy)
    need, need2 = C.c_size_t(), C.c_size_t()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.sty_conv1d_workspace_bytes(Co, Ci, K, C.byref(need)))
    L.check(lib.sty_conv1d_bwd_workspace_bytes(B, Ci, Co, K, T, C.byref(need2)))
    tw_bytes = B * (Ci + Co) * T * 2 + 1024
    out = {}
    for mode in (1, 2):
        ws = torch.zeros(need.value + tw_bytes, dtype=torch.uint8, device=DEV)
        y = torch.empty(B, Co, T, device=DEV)
        L.prof_report(256)
        lib.sty_prof_enable(1)
        try:
            L.check(lib.sty_conv1d_fwd(B, Ci, Co, K, d, T, L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(y), L.ptr(ws), ws.numel(), mode, st))
            ws2 = torch.empty(need2.value, dtype=torch.uint8, device=DEV)
            dw, dx = torch.empty(Co, Ci, K, device=DEV), torch.empty(B, Ci, T, device=DEV)
            L.check(lib.sty_conv1d_bwd(B, Ci, Co, K, d, T, L.ptr(xd), L.ptr(wd), L.ptr(gd), L.ptr(dw), None, L.ptr(dx),
                                       L.ptr(ws2), ws2.numel(), mode, st))
            torch.cuda.synchronize()
        finally:
            lib.sty_prof_enable(0)
        rows = L.prof_report(256)
        assert sum(r["launches"] for r in rows if r["name"].startswith(("convp16_kernel", "convq_kernel", "convk1_kernel"))) >= 2, rows
        if mode == 2 and K == 3 and T % 4 == 0:  # (K = 1: convk1_kernel takes the launch where it is eligible)
            assert sum(r["launches"] for r in rows if r["name"].startswith("convq_kernel")) >= 2, rows
        out[mode] = (y.cpu(), dx.cpu())
        if mode == 2:  # the output twin: [B][Ci][T] bf16 of the input, then [B][Co][T] bf16 of lrelu(y)
            base = (ws.data_ptr() + need.value + 255) // 256 * 256 - ws.data_ptr()
            tw = ws[base:base + 2 * B * (Ci + Co) * T].view(torch.bfloat16)
            x16, y16 = tw[:B * Ci * T].view(B, Ci, T).cpu(), tw[B * Ci * T:].view(B, Co, T).cpu()
            assert torch.equal(x16, x.bfloat16())
            assert torch.equal(y16, torch.nn.functional.leaky_relu(y.cpu(), 0.2).bfloat16())
    err = (out[2][0] - ref).abs().max().item()
    e_dx = (out[2][1] - ref_dx).abs().max().item() / ref_dx.abs().max().item()
    print(f"\n  convp16 on twins {shape}: max|err| y {err:.3e}  dx {e_dx:.3e}; bit-equal to the fp32-operand kernel: "
          f"y {torch.equal(out[1][0], out[2][0])}  dx {torch.equal(out[1][1], out[2][1])}")
    assert err <= 2e-5 and e_dx <= 2e-5
    assert torch.equal(out[1][0], out[2][0]) and torch.equal(out[1][1], out[2][1])
    # NEGATIVE CONTROL of this gate (round 6): the same launch with the weights scaled by 1 + 2^-8 -- one ulp of bf16 on the products,
    # what a kernel with a wrong constant or a biased rounding would do -- has to be RED here, at the block level, because at
    # full size it is below what 60 bf16 layers leave of most gradients (tests/test_full_size.py, C3_WELL_CONDITIONED)
    ws = torch.zeros(need.value + tw_bytes, dtype=torch.uint8, device=DEV)
    yb = torch.empty(B, Co, T, device=DEV)
    L.check(lib.sty_conv1d_fwd(B, Ci, Co, K, d, T, L.ptr(xd), L.ptr(dev(w * (1.0 + 2.0 ** -8))), L.ptr(bd), L.ptr(yb), L.ptr(ws),
                               ws.numel(), 2, st))
    torch.cuda.synchronize()
    err_b = (yb.cpu() - ref).abs().max().item()
    print(f"  negative control (weights x (1 + 2^-8)): max|err| {err_b:.3e} against the gate 2e-5")
    assert err_b > 20 * 2e-5


@pytest.mark.parametrize("shape", [(3, 128, 130, 1, 1, 64), (2, 512, 64, 3, 1, 38), (2, 96, 200, 3, 1, 300), (2, 64, 96, 3, 1, 1000),
                                   (3, 130, 70, 5, 1, 258), (48, 240, 80, 3, 1, 1500), (4, 256, 384, 1, 1, 2100),
                                   (2, 100, 64, 3, 1, 130), (1, 64, 64, 1, 1, 6)])
def test_weight_gradient_on_bf16_operand_twins_vs_torch(shape):
    """wgradb16_kernel: the K = 1 / 3 / 5 weight gradient (and its bias by-product) with BOTH operands read as bf16 twins
    from HBM (ConvArgs::x16 / g16; the unit entry point makes them with the cast pass when compute_bf16 = 2) against float64
    on the same bf16-rounded operands -- the rounding is the one the fp32-operand path applies on the way into LDS, so the
    gate is the fp32 accumulation bound of test_dense_conv1d_vs_torch -- and against the fp32-operand kernel (compute_bf16 =
    1) element by element: same products, same order within a chunk.  The bias gradient sums the ROUNDED gradient here
    (one more MFMA against ones), the unrounded one there: 2^-9 per element, gate 4e-3.  Shapes: row lengths that are not
    multiples of 8 (row ends inside a group), T = 6 (everything on the sample-by-sample path), channel counts that are not
    multiples of 64, the 128 x 128 blocks of the wide K = 1 layers."""
    import ctypes as C
    from stylish_tts_amd import lib as L
    lib = L.load()
    B, Ci, Co, K, d, T = shape
    g = torch.Generator().manual_seed(sum(shape))
    x, w = torch.randn(B, Ci, T, generator=g), torch.randn(Co, Ci, K, generator=g) / (Ci * K) ** 0.5
    gy = torch.randn(B, Co, T, generator=g)
    rnd = lambda t: t.bfloat16().double()
    pad = (K - 1) * d // 2
    xr, wr, gr = rnd(x).requires_grad_(True), rnd(w).requires_grad_(True), rnd(gy)
    (torch.nn.functional.conv1d(xr, wr, None, padding=pad, dilation=d) * gr).sum().backward()
    ref_dw, ref_db = wr.grad.float(), gy.double().sum((0, 2)).float()
    xd, wd, gd = dev(x), dev(w), dev(gy)
    need = C.c_size_t()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.sty_conv1d_bwd_workspace_bytes(B, Ci, Co, K, T, C.byref(need)))
    out = {}
    for mode in (1, 2):
        ws2 = torch.empty(need.value, dtype=torch.uint8, device=DEV)
        dw, db = torch.empty(Co, Ci, K, device=DEV), torch.empty(Co, device=DEV)
        L.prof_report(256)
        lib.sty_prof_enable(1)
        try:
            L.check(lib.sty_conv1d_bwd(B, Ci, Co, K, d, T, L.ptr(xd), L.ptr(wd), L.ptr(gd), L.ptr(dw), L.ptr(db), None,
                                       L.ptr(ws2), ws2.numel(), mode, st))
            torch.cuda.synchronize()
        finally:
            lib.sty_prof_enable(0)
        names = [r["name"] for r in L.prof_report(256)]
        if mode == 2:
            assert any(n.startswith("wgradb16_kernel") for n in names), names
        out[mode] = (dw.cpu(), db.cpu())
    e_dw = (out[2][0] - ref_dw).abs().max().item() / ref_dw.abs().max().item()
    e_db = (out[2][1] - ref_db).abs().max().item() / ref_db.abs().max().item()
    e_ab = (out[2][0] - out[1][0]).abs().max().item() / ref_dw.abs().max().item()
    print(f"\n  wgrad on twins {shape}: dw vs float64 {e_dw:.3e}  dbias {e_db:.3e}  dw vs the fp32-operand kernel {e_ab:.3e}")
    assert e_dw <= 2e-5 and e_ab <= 2e-5 and e_db <= 4e-3


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 32, 32, 11, 5, 700), (3, 32, 32, 21, 1, 1333), (2, 22, 32, 7, 1, 513),
                                   (1, 32, 30, 11, 3, 100), (5, 32, 32, 11, 3, 2049), (300, 32, 32, 11, 1, 1100)])
def test_persistent_conv32_vs_torch(shape, compute, monkeypatch):
    """conv32p_kernel (persistent, producer / consumer waves; the 32 -> 32 channel convs at the 75T rate) through the
    same unit entry points: forward and input gradient vs float64 on the same (bf16-rounded) operands.  Small shapes
    are forced onto it with STY_CONV32P_MIN_TILES=1; the last shape takes it by itself (900 tiles, 3-4 per workgroup)."""
    from stylish_tts_amd import lib as L
    lib = L.load()
    if shape[0] < 300:
        monkeypatch.setenv("STY_CONV32P_MIN_TILES", "1")
    L.prof_report(256)
    lib.sty_prof_enable(1)
    try:
        test_dense_conv1d_vs_torch(shape, compute)
    finally:
        lib.sty_prof_enable(0)
    rows = L.prof_report(256)
    names = [r["name"] for r in rows]
    assert any(n.startswith("conv32p_kernel") for n in names), names
    assert sum(r["launches"] for r in rows if r["name"].startswith("conv32p_kernel")) >= 2, rows  # forward + input gradient


@pytest.mark.parametrize("shape", [(2, 37, 80, 5, 2, 203), (3, 128, 130, 1, 1, 64), (2, 512, 64, 3, 1, 37), (2, 96, 200, 3, 1, 300),
                                   (2, 64, 96, 3, 1, 1000), (3, 130, 33, 5, 1, 257), (48, 240, 80, 3, 1, 1500),
                                   (32, 128, 512, 3, 1, 100), (32, 512, 128, 3, 1, 100), (32, 128, 128, 5, 1, 100)])
def test_persistent_conv16_vs_torch(shape, monkeypatch):
    """convp16_kernel (bf16 compute mode, Cin >= 64: producer / consumer waves over 32-channel chunks, bf16 LDS tiles)
    through the unit entry points: forward and input gradient vs float64 on the same bf16-rounded operands.  Small shapes
    are forced onto it with STY_CONVP16_MIN_TILES=1; the last one takes it by itself."""
    from stylish_tts_amd import lib as L
    lib = L.load()
    if shape[0] < 48:
        monkeypatch.setenv("STY_CONVP16_MIN_TILES", "1")
    L.prof_report(256)
    lib.sty_prof_enable(1)
    try:
        test_dense_conv1d_vs_torch(shape, "bf16")
    finally:
        lib.sty_prof_enable(0)
    rows = L.prof_report(256)
    assert sum(r["launches"] for r in rows if r["name"].startswith("convp16_kernel")) >= 1, rows


@pytest.mark.parametrize("shape", [(3, 128, 130, 1, 1, 64), (2, 256, 1024, 1, 1, 520), (2, 1024, 256, 1, 1, 132), (5, 320, 96, 1, 1, 1000),
                                   (40, 64, 512, 1, 1, 520)])
def test_pointwise_gemm_kernel_vs_torch(shape, monkeypatch):
    """convk1_kernel (bf16 compute mode, K = 1: a plain GEMM whose B operands come out of LDS through the transposing read
    ds_read_b64_tr_b16, 16-byte loads of the [B][C][T] activations) through the unit entry points: forward and input
    gradient vs float64 on the same bf16-rounded operands -- Cin / Cout that are not multiples of the 64-channel chunk or
    the 128-row tile, T that is not a multiple of the 128-column tile, the last shape big enough to take the kernel by
    itself."""
    from stylish_tts_amd import lib as L
    lib = L.load()
    if shape[0] < 40:
        monkeypatch.setenv("STY_CONVK1_MIN_TILES", "1")
    L.prof_report(256)
    lib.sty_prof_enable(1)
    try:
        test_dense_conv1d_vs_torch(shape, "bf16")
    finally:
        lib.sty_prof_enable(0)
    rows = L.prof_report(256)
    assert sum(r["launches"] for r in rows if r["name"].startswith("convk1_kernel")) >= 1, [r["name"] for r in rows]


def test_persistent_kernels_match_the_tiled_kernel_in_the_bf16_graphs(env, monkeypatch):
    """The bf16 compute mode with the persistent kernels forced on at the small test size (conv32p, convp16: flat 2-D
    style-encoder convs with masks and residuals, LeakyReLU / AdaIN prologues, ReLU, decoder and vocoder convs, forward
    and input gradients) and the bf16-tile weight-gradient kernel (wgradb: the same layers' weight gradients, flat 2-D
    rows, pad-column masks on G, prologues on x) against the same graphs on the tiled kernels: the operands are rounded identically, so the
    kernels differ by fp32 summation order only -- but in the bf16 mode a 1e-7 difference flips bf16 roundings downstream
    (2^-9 each), which the vocoder's phase path amplifies: the predictor end to end is REPORTED (it is as far from the
    tiled bf16 run as that is from fp32, test_acoustic_train_step_bf16_compute_vs_fp32); the style encoder, a plain conv
    stack, is gated tightly, and the kernels themselves are pinned by test_persistent_conv16_vs_torch / _conv32_."""
    import stylish_tts_amd as S
    from oracle.manifest import style_encoder_manifest
    from oracle.weights import fill_state_dict
    from stylish_tts_amd import lib as L
    cs = env["cs"]
    mel = torch.randn(3, 1, 80, 161, generator=torch.Generator().manual_seed(9))
    gs = torch.randn(3, 64, generator=torch.Generator().manual_seed(10))
    out = {}
    for mode in ("tiled", "persistent"):
        for k in ("STY_NO_CONVP16", "STY_NO_CONV32P", "STY_NO_WGRADB", "STY_NO_CONVK1", "STY_CONVP16_MIN_TILES",
                  "STY_CONV32P_MIN_TILES", "STY_CONVK1_MIN_TILES"):
            monkeypatch.delenv(k, raising=False)
        if mode == "tiled":
            monkeypatch.setenv("STY_NO_CONVP16", "1")
            monkeypatch.setenv("STY_NO_CONV32P", "1")
            monkeypatch.setenv("STY_NO_WGRADB", "1")
            monkeypatch.setenv("STY_NO_CONVK1", "1")
        else:
            monkeypatch.setenv("STY_CONVP16_MIN_TILES", "1")
            monkeypatch.setenv("STY_CONV32P_MIN_TILES", "1")
            monkeypatch.setenv("STY_CONVK1_MIN_TILES", "1")  # (round 5: the K = 1 GEMM kernel from 24 tiles; here every eligible launch)
        se = S.MelStyleEncoder()
        se.load_state_dict(fill_state_dict(style_encoder_manifest(), 0))
        se = se.to(DEV).enable_training().set_train_opts(compute_bf16=True)
        L.prof_report(512)
        L.load().sty_prof_enable(1)
        style = se.forward_train(dev(mel))
        se.backward(dev(gs))
        sp = S.SpeechPredictor()
        sp.load_state_dict({k: v.clone() for k, v in env["P"].items()}, strict=False)
        sp = sp.to(DEV).enable_training().set_train_opts(compute_bf16=True)
        audio = sp.forward_train(dev(cs["texts"]), dev(cs["text_lengths"]), dev(env["ali"]), dev(cs["pitch"]),
                                 dev(cs["energy"]), dev(env["voiced"]), dev(cs["style"]), dev(cs["pitch"]),
                                 noise=dev(cs["noise"]))
        d_style, _ = sp.backward(torch.sign(audio) / audio.numel(), want_energy=False)
        torch.cuda.synchronize()
        L.load().sty_prof_enable(0)
        names = {r["name"] for r in L.prof_report(512)}
        has = (any(n.startswith("convp16") for n in names), any(n.startswith("conv32p") for n in names),
               any(n.startswith("wgradb") for n in names))
        assert has == ((True, True, True) if mode == "persistent" else (False, False, False)), names
        assert mode == "persistent" or not any(n.startswith("convk1") for n in names), names
        out[mode] = dict(style=style.cpu(), audio=audio.cpu(), d_style=d_style.cpu(),
                         gse=torch.cat([p.grad.flatten().cpu() for p in se.parameters()]),
                         gsp=torch.cat([p.grad.flatten().cpu() for p in sp.parameters()]))
    a, b = out["tiled"], out["persistent"]
    rel = lambda x, y: ((x - y).norm() / y.norm()).item()
    print(f"\n  persistent vs tiled (bf16 mode): style {rel(b['style'], a['style']):.2e}  audio {rel(b['audio'], a['audio']):.2e}  "
          f"style-encoder grads {rel(b['gse'], a['gse']):.2e}  predictor grads {rel(b['gsp'], a['gsp']):.2e}  "
          f"d_style {rel(b['d_style'], a['d_style']):.2e}")
    assert rel(b["style"], a["style"]) <= 1e-4 and rel(b["gse"], a["gse"]) <= 1e-3
    # (reported; the bounds only catch a broken kernel -- NaN, a missing tile -- not the chaotic divergence described above)
    assert rel(b["audio"], a["audio"]) <= 0.1 and rel(b["gsp"], a["gsp"]) <= 1.5 and rel(b["d_style"], a["d_style"]) <= 1.5
    assert all(bool(torch.isfinite(v).all()) for v in b.values())


def test_bf16_weight_gradient_kernels_match_the_fp32_tile_kernels_in_the_graphs(env, monkeypatch):
    """A weight gradient is a leaf of the backward graph: switching ONLY the weight-gradient kernels (wgradb / wgradp32,
    bf16 tiles in LDS, vs the fp32-tile kernels they replace in the bf16 mode; STY_NO_WGRADB) leaves every activation
    and every input gradient bit-identical, so each parameter gradient of the two runs differs by fp32 summation order
    alone -- a tight gate on the new kernels inside the real graphs: flat 2-D rows, pad-column masks on G, AdaIN / Snake /
    LeakyReLU prologues, the 3 x 32-channel concatenated input of phase_input_conv, k = 1 / 3 / 11 / 21."""
    import stylish_tts_amd as S
    from oracle.manifest import style_encoder_manifest
    from oracle.weights import fill_state_dict
    from stylish_tts_amd import lib as L
    cs = env["cs"]
    mel = torch.randn(3, 1, 80, 161, generator=torch.Generator().manual_seed(9))
    gs = torch.randn(3, 64, generator=torch.Generator().manual_seed(10))
    out = {}
    monkeypatch.setenv("STY_WGRADB_WIDE_MIN", "1")  # the 128 x 128 blocks of the wide K = 1 layers at the test size too
    # the lean ConvNeXt32 backward (its own test: test_block_bf16_mode_vs_float64_oracle_on_rounded_operands) changes the
    # block's INPUT gradient too (ds from bf16 h): off here, so that the weight-gradient kernels are the only difference
    monkeypatch.setenv("STY_NO_CNX_LEAN", "1")
    for mode in ("old", "new"):
        monkeypatch.delenv("STY_NO_WGRADB", raising=False)
        if mode == "old":
            monkeypatch.setenv("STY_NO_WGRADB", "1")
        se = S.MelStyleEncoder()
        se.load_state_dict(fill_state_dict(style_encoder_manifest(), 0))
        se = se.to(DEV).enable_training().set_train_opts(compute_bf16=True)
        L.prof_report(512)
        L.load().sty_prof_enable(1)
        se.forward_train(dev(mel))
        se.backward(dev(gs))
        sp = S.SpeechPredictor()
        sp.load_state_dict({k: v.clone() for k, v in env["P"].items()}, strict=False)
        sp = sp.to(DEV).enable_training().set_train_opts(compute_bf16=True)
        audio = sp.forward_train(dev(cs["texts"]), dev(cs["text_lengths"]), dev(env["ali"]), dev(cs["pitch"]),
                                 dev(cs["energy"]), dev(env["voiced"]), dev(cs["style"]), dev(cs["pitch"]),
                                 noise=dev(cs["noise"]))
        d_style, _ = sp.backward(torch.sign(audio) / audio.numel(), want_energy=False)
        torch.cuda.synchronize()
        L.load().sty_prof_enable(0)
        names = {r["name"] for r in L.prof_report(512)}
        has = any(n.startswith("wgradb") for n in names), any(n.startswith("wgradp32") for n in names)
        assert has == ((True, True) if mode == "new" else (False, False)), names
        out[mode] = dict(audio=audio.cpu(), d_style=d_style.cpu(),
                         g={("se." + k): p.grad.cpu().clone() for k, p in se.named_parameters()}
                         | {("sp." + k): p.grad.cpu().clone() for k, p in sp.named_parameters()})
    assert torch.equal(out["old"]["audio"], out["new"]["audio"])
    ds = ((out["new"]["d_style"] - out["old"]["d_style"]).norm() / out["old"]["d_style"].norm()).item()
    print(f"\n  d_style new vs old: {ds:.2e}")
    worst = []
    for k, a in out["old"]["g"].items():
        b = out["new"]["g"][k]
        den = a.norm().item()
        if k.endswith(".bias"):
            # a bias in front of an instance norm (convs1.* of the resblocks, conv1 of the decoder blocks) has a structurally
            # zero gradient: rounding noise on both sides.  Recognised by its size next to the layer's weight gradient.
            wk = [k[:-4] + s_ for s_ in ("weight", "parametrizations.weight.original1", "weight_orig")]
            wn = max([out["old"]["g"][q].norm().item() for q in wk if q in out["old"]["g"]] + [0.0])
            if den < 1e-4 * wn:
                continue
        if den > 0:
            e = (b - a).norm().item() / den
            if k.endswith(".pwconv1.bias") and "phase_convnext" in k or k.endswith("upblocks.2.pwconv1.bias"):
                # fused ConvNeXt32 blocks: in the new path gH0 is STORED as bf16 for the weight-gradient GEMM (whose MFMA
                # rounds it to bf16 anyway), so this bias gradient is a sum of bf16-rounded values: 2^-9 per term
                assert e <= 2e-3, (k, e)
                continue
            if k.startswith("se.") and k.endswith(".bias"):
                # style encoder with operand twins (the default of the new path): wgradb16_kernel has only the bf16 twin of
                # G, its bias gradient is the row sum of that twin (one more MFMA against ones, fp32 accumulation) -- what
                # autocast's conv backward sums as well; the fp32-operand kernels sum the values before rounding.  2^-9 per
                # term, 2^-8 as the bound for a sum with cancellation
                assert e <= 4e-3, (k, e)
                continue
            if "_prior_block.convs" in k:
                # the resblock convs: since round 5 the persistent kernels (conv32p_kernel, wgradp32_kernel) evaluate the Snake
                # of their AdaIN + Snake prologue with the hardware sine behind one range check per group, the tiled kernels of
                # the old path with the polynomial: operands that sit on a bf16 rounding boundary round the other way in one of
                # the two (1e-6 apart before rounding, 2^-8 after).  The forward and the weight gradient of the NEW path see the
                # same operands (both kernels take the same sine); measured 3e-4 ... 8.5e-3, the largest on the weight-norm
                # gains (original0: a cancellation, see test_block_backward_vs_float64_oracle)
                assert e <= (2e-2 if k.endswith("original0") else 5e-3), (k, e)
                continue
            worst.append((e, k))
    worst.sort(reverse=True)
    print("\n  weight-gradient kernels, new vs old (relative L2 per tensor), worst five: " +
          "  ".join(f"{k} {e:.2e}" for e, k in worst[:5]))
    assert worst[0][0] <= 2e-4 and ds <= 1e-5, worst[:5]


def _sub(t, stride=97):
    t = t.detach().flatten()
    return t[::stride] if t.numel() > 4096 else t


def test_speech_predictor_train_mode_vs_reference_golden(env):
    """module.train() behaviour on the HIP path: BatchNorm batch statistics + running-buffer update, Decoder box
    smoothing (widths 7 / 15) and the gradients through both vs what the REFERENCE produced in .train() with dropout
    off (tests/golden/sp_train_small.safetensors, tools/gen_golden_train.py)."""
    import stylish_tts_amd as S
    from safetensors.torch import load_file
    gold = load_file(os.path.join(G, "sp_train_small.safetensors"))
    cs, ali = env["cs"], env["ali"]
    P = {k: v.clone() for k, v in env["P"].items()}
    m = S.SpeechPredictor()
    m.load_state_dict(P, strict=False)
    m = m.to(DEV).enable_training().set_train_opts(bn_batch_stats=True, f0_smooth=7, energy_smooth=15)
    audio = m.forward_train(dev(cs["texts"]), dev(cs["text_lengths"]), dev(ali), dev(cs["pitch"]), dev(cs["energy"]),
                            dev(env["voiced"]), dev(cs["style"]), dev(cs["pitch"]), noise=dev(cs["noise"]))
    d_style, d_energy = m.backward(torch.sign(audio) / audio.numel())
    torch.cuda.synchronize()
    mse = ((audio.cpu() - gold["audio"]) ** 2).mean().item()
    print(f"\n  train-mode forward vs reference: mse {mse:.3e}")
    assert mse <= 1e-8
    rep = Report()
    sd = m.state_dict()
    bn = "generator.amp_conformer.layers.0.conv.net.4."
    rep.add("BN running_mean", sd[bn + "running_mean"], gold["bn.running_mean"], 1e-5)
    rep.add("BN running_var", sd[bn + "running_var"], gold["bn.running_var"], 1e-5)
    rep.add("d_style", d_style, gold["grad.style"], 3e-2)
    rep.add("d_energy (through the smoothing)", d_energy, gold["grad.energy"], 3e-2)
    named = dict(m.named_parameters())
    for k in [k[len("grad."):] for k in gold if k.startswith("grad.") and k not in ("grad.style", "grad.energy")]:
        rep.add("d " + k[-44:], _sub(named[k].grad), gold["grad." + k], 3e-2)
    rep.done()


def test_style_encoder_train_mode_vs_reference_golden():
    """Spectral-norm power iteration on the HIP path (u, v buffers refreshed in place once per training forward) and the
    gradients through W / sigma vs the REFERENCE in .train() (tests/golden/se_train_small.safetensors)."""
    import stylish_tts_amd as S
    from oracle.manifest import style_encoder_manifest
    from oracle.weights import fill_state_dict
    from safetensors.torch import load_file
    from tests.cases import make_case
    gold = load_file(os.path.join(G, "se_train_small.safetensors"))
    se = S.MelStyleEncoder()
    se.load_state_dict(fill_state_dict(style_encoder_manifest(), 0))
    se = se.to(DEV).enable_training().set_train_opts(sn_power_iter=True)
    out = se.forward_train(dev(make_case("se_small")["mel"]))
    se.backward(dev(gold["cotangent"]))
    torch.cuda.synchronize()
    rep = Report()
    rep.add("style", out, gold["style"], 1e-5)
    sd = se.state_dict()
    named = dict(se.named_parameters())
    for k in ("shared.0", "shared.2.conv1", "shared.2.downsample_res.conv", "shared.6"):
        rep.add(k + ".weight_u", sd[k + ".weight_u"], gold[k + ".weight_u"], 1e-5)
        rep.add(k + ".weight_v", sd[k + ".weight_v"], gold[k + ".weight_v"], 1e-5)
        rep.add("d " + k + ".weight_orig", _sub(named[k + ".weight_orig"].grad), gold["grad." + k + ".weight_orig"], 1e-3)
    rep.done()


def test_style_encoder_prepare_train_then_forward_matches_reference_golden():
    """sty_style_prepare_train (the weight-side half issued ahead of the input, here on another stream) followed by
    forward_train must be the same training forward: one power iteration, not two, same style and gradients as the
    REFERENCE in .train() (tests/golden/se_train_small.safetensors); the call after it prepares for itself again."""
    import stylish_tts_amd as S
    from oracle.manifest import style_encoder_manifest
    from oracle.weights import fill_state_dict
    from safetensors.torch import load_file
    from tests.cases import make_case
    gold = load_file(os.path.join(G, "se_train_small.safetensors"))
    se = S.MelStyleEncoder()
    se.load_state_dict(fill_state_dict(style_encoder_manifest(), 0))
    se = se.to(DEV).enable_training().set_train_opts(sn_power_iter=True)
    mel = dev(make_case("se_small")["mel"])
    side = torch.cuda.Stream(device=DEV)
    main = torch.cuda.current_stream(DEV)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        se.prepare_train(DEV)
        out = se.forward_train(mel)
        se.backward(dev(gold["cotangent"]))
    main.wait_stream(side)
    torch.cuda.synchronize()
    rep = Report()
    rep.add("style", out, gold["style"], 1e-5)
    sd = se.state_dict()
    named = dict(se.named_parameters())
    for k in ("shared.0", "shared.2.conv1", "shared.6"):
        rep.add(k + ".weight_u", sd[k + ".weight_u"], gold[k + ".weight_u"], 1e-5)
        rep.add("d " + k + ".weight_orig", _sub(named[k + ".weight_orig"].grad), gold["grad." + k + ".weight_orig"], 1e-3)
    rep.done()
    # a second forward without prepare_train runs its own power iteration: u moves on
    u1 = sd["shared.0.weight_u"].clone()
    se.forward_train(mel)
    torch.cuda.synchronize()
    assert not torch.equal(u1, se.state_dict()["shared.0.weight_u"])


def test_speech_predictor_prepare_train_then_forward_equals_the_plain_forward(env):
    """sty_speech_prepare_train (the weight-side half of the predictor's training forward issued ahead of time, here on
    another stream) followed by forward_train + backward: the audio equals that of a model that prepares inside
    forward_train bit for bit, d style and every parameter gradient to 1e-5 (float-atomic sums); a second forward_train
    without the call prepares itself again (the flag is consumed), and a parameter change after prepare_train cancels it."""
    import stylish_tts_amd as S
    cs, ali = env["cs"], env["ali"]

    def run(early):
        m = S.SpeechPredictor()
        m.load_state_dict({k: v.clone() for k, v in env["P"].items()}, strict=False)
        m = m.to(DEV).enable_training()
        args = (dev(cs["texts"]), dev(cs["text_lengths"]), dev(ali), dev(cs["pitch"]), dev(cs["energy"]), dev(env["voiced"]),
                dev(cs["style"]), dev(cs["pitch"]))
        a0 = m.forward_train(*args, noise=dev(cs["noise"]))          # builds the plan, leaves the model "stale"
        m.backward(torch.sign(a0) / a0.numel(), want_energy=False)
        for p in m.parameters():
            if p.grad is not None:
                p.grad.zero_()
        if early:
            side = torch.cuda.Stream(device=DEV)
            side.wait_stream(torch.cuda.current_stream(DEV))
            with torch.cuda.stream(side):
                m.prepare_train(DEV)
            torch.cuda.current_stream(DEV).wait_stream(side)
        audio = m.forward_train(*args, noise=dev(cs["noise"]))
        d_style, _ = m.backward(torch.sign(audio) / audio.numel(), want_energy=False)
        torch.cuda.synchronize()
        return audio.cpu(), d_style.cpu(), {k: p.grad.cpu().clone() for k, p in m.named_parameters() if p.grad is not None}, m

    a1, s1, g1, _ = run(False)
    a2, s2, g2, m = run(True)
    assert torch.equal(a1, a2)  # same prepared weights -> the same forward, bit for bit
    # the backward holds float-atomic sums (d style over the style projections, Snake alpha, GRN gamma): two runs of the
    # SAME configuration differ in the last bit there, so the gradients are compared at 1e-5 of their scale
    assert rel_err(s2, s1) <= 1e-5
    assert g1.keys() == g2.keys() and len(g1) > 100
    for k in g1:
        assert rel_err(g2[k], g1[k]) <= 1e-5, (k, rel_err(g2[k], g1[k]))
    # a parameter change after prepare_train must not be lost: load_state_dict bumps the version counters -> invalidate
    m.prepare_train(DEV)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    k0 = "text_encoder.prenet.conv_layers.0.weight"  # a PACKED weight: a stale pack would reproduce a2
    sd[k0] = sd[k0] * 1.5
    m.load_state_dict(sd, strict=False)
    a3 = m.forward_train(dev(cs["texts"]), dev(cs["text_lengths"]), dev(ali), dev(cs["pitch"]), dev(cs["energy"]),
                         dev(env["voiced"]), dev(cs["style"]), dev(cs["pitch"]), noise=dev(cs["noise"]))
    torch.cuda.synchronize()
    assert not torch.equal(a3.cpu(), a2)


def test_speech_predictor_dropout_vs_patched_reference_golden(env):
    """Dropout on the HIP path (counter-based hash masks in the TextEncoder: prenet, attention probabilities inside the
    MFMA attention kernel and its backward, post-attention, FFN) vs the REFERENCE run in .train() with F.dropout / SDPA
    patched to the same mask function (tests/golden/sp_train_dropout_small.safetensors)."""
    import stylish_tts_amd as S
    from safetensors.torch import load_file
    gold = load_file(os.path.join(G, "sp_train_dropout_small.safetensors"))
    cs, ali = env["cs"], env["ali"]
    P = {k: v.clone() for k, v in env["P"].items()}
    m = S.SpeechPredictor()
    m.load_state_dict(P, strict=False)
    m = m.to(DEV).enable_training().set_train_opts(bn_batch_stats=True, f0_smooth=15, energy_smooth=0,
                                                   dropout_seed=1234, text_dropout=0.2)
    audio = m.forward_train(dev(cs["texts"]), dev(cs["text_lengths"]), dev(ali), dev(cs["pitch"]), dev(cs["energy"]),
                            dev(env["voiced"]), dev(cs["style"]), dev(cs["pitch"]), noise=dev(cs["noise"]))
    d_style, _ = m.backward(torch.sign(audio) / audio.numel(), want_energy=False)
    torch.cuda.synchronize()
    mse = ((audio.cpu() - gold["audio"]) ** 2).mean().item()
    print(f"\n  dropout forward vs patched reference: mse {mse:.3e}")
    assert mse <= 1e-8
    rep = Report()
    rep.add("d_style", d_style, gold["grad.style"], 3e-2)
    named = dict(m.named_parameters())
    for k in [k[len("grad."):] for k in gold if k.startswith("grad.") and k != "grad.style"]:
        rep.add("d " + k[-44:], _sub(named[k].grad), gold["grad." + k], 3e-2)
    rep.done()


def test_training_from_sample_dataset_files(tmp_path, env):
    """N2 + A0: a dataset in the reference's sample_dataset layout (wav-dir, training-list, pitch / alignment safetensors)
    goes through the loader counterparts (stylish_tts_amd.data) into train_acoustic steps on the HIP path."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from make_sample_dataset import make
    from stylish_tts_amd import data as D
    root = str(tmp_path)
    make(root, 12, 7)
    lines = open(os.path.join(root, "training-list.txt"), encoding="utf-8").read().splitlines()
    ds = D.SampleDataset(data_list=lines, root_path=os.path.join(root, "wav-dir"),
                         pitch_path=os.path.join(root, "pitch.safetensors"),
                         alignment_path=os.path.join(root, "alignment.safetensors"))
    bins, _ = ds.time_bins()
    loader = torch.utils.data.DataLoader(ds, batch_sampler=D.LengthBinSampler(bins, lambda k: 4),
                                         collate_fn=D.Collater(stage="acoustic", hop_length=300))
    tr, _, _ = _train_setup(env, 1e-4, train_mode=True)
    seen = 0
    for batch in loader:
        kw = D.to_step_inputs(batch, DEV)
        losses = tr.train_batch(**kw)
        assert bool(torch.isfinite(losses).all()), losses
        assert tr.audio.shape == (kw["audio_gt"].shape[0], 1, kw["audio_gt"].shape[1])
        seen += kw["audio_gt"].shape[0]
    torch.cuda.synchronize()
    assert seen == len(lines)


def test_train_entry_point_runs_c1_from_the_yaml_files(tmp_path):
    """BASELINE.json configs[0] (c1): "sample_dataset single-speaker, batch=2 ... via config/config.yml" as a CONFIG, not a test
    body: config.yml + model.yml (the reference's sections and field names) -> stylish_tts_amd.train.train(...) -> three
    train_acoustic steps at batch 2 (adversarial terms on, as the reference's stage has them) -> accelerate-layout
    checkpoint_final with manifest / normalization; then the same command with --checkpoint resumes (optimizer step counts
    continue), and the NEXT stage (textual, stage_type.py:394) starts from the acoustic checkpoint with the speech predictor
    frozen; a duration step closes the chain."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from make_sample_dataset import make
    from stylish_tts_amd import stage_io as IO
    from stylish_tts_amd import train as T
    from tests.test_boundary import _default_config_yaml, _default_model_yaml
    root = tmp_path / "data"
    make(str(root), 8, 11)
    cfg, mdl, out = tmp_path / "config.yml", tmp_path / "model.yml", tmp_path / "out"
    cfg.write_text(_default_config_yaml(root, acoustic=dict(epochs=3)))
    mdl.write_text(_default_model_yaml())
    import json
    logs = []
    ctx = T.train(str(cfg), str(mdl), str(out), "acoustic", max_steps=3, log=logs.append)
    torch.cuda.synchronize()
    assert ctx.manifest.current_total_step == 3 and ctx.manifest.stage == "acoustic"
    assert ctx.normalization.frames > 0  # computed from the train split, written beside the stage and the dataset
    final = os.path.join(str(out), "acoustic", "checkpoint_final")
    for k in ("speech_predictor", "speech_style_encoder", "mrd0", "mrd1", "mrd2", "disc"):
        assert os.path.exists(os.path.join(final, IO.model_file(k))), k
        assert os.path.exists(os.path.join(final, IO.optimizer_file(k))), k
    assert os.path.exists(os.path.join(final, IO.COMPLETE_MARKER))
    assert os.path.exists(os.path.join(str(out), "acoustic", "config.yml"))
    assert json.load(open(os.path.join(str(out), "acoustic", "acoustic_batch_sizes.json")))  # every bin at probe_batch_max = 2
    assert any("acoustic epoch" in ln for ln in logs)
    w_before = ctx.models["speech_predictor"].state_dict()["decoder.asr_res.0.weight"].clone() if \
        "decoder.asr_res.0.weight" in ctx.models["speech_predictor"].state_dict() else None
    t_before = ctx.stage.trainer.opt["speech_predictor"].t
    assert t_before == 3
    del ctx
    # resume in the same stage
    ctx = T.train(str(cfg), str(mdl), str(out), "acoustic", checkpoint=final, max_steps=1, log=logs.append)
    assert ctx.manifest.current_total_step == 4 and ctx.stage.trainer.opt["speech_predictor"].t == 4
    if w_before is not None:
        assert ctx.models["speech_predictor"].state_dict()["decoder.asr_res.0.weight"].shape == w_before.shape
    del ctx
    # the next stage from the acoustic checkpoint: trained models are new, the frozen speech predictor comes from the file
    sp_file = torch.load(os.path.join(final, IO.model_file("speech_predictor")), weights_only=True)
    ctx = T.train(str(cfg), str(mdl), str(out), "textual", checkpoint=final, max_steps=1, log=logs.append)
    torch.cuda.synchronize()
    assert ctx.manifest.stage == "textual" and ctx.manifest.current_step == 1
    k0 = next(k for k, v in sp_file.items() if v.is_floating_point() and v.ndim > 1)
    got = ctx.models["speech_predictor"].state_dict()[k0].cpu()
    # (resumed one more step above: the file holds the state after step 4)
    sp4 = torch.load(os.path.join(final, IO.model_file("speech_predictor")), weights_only=True)[k0]
    assert torch.equal(got, sp4)
    assert os.path.exists(os.path.join(str(out), "textual", "checkpoint_final", IO.model_file("pitch_energy_predictor")))
    del ctx
    tfinal = os.path.join(str(out), "textual", "checkpoint_final")
    assert os.path.exists(os.path.join(tfinal, IO.model_file("speech_predictor")))  # (every model the run holds is saved)
    ctx = T.train(str(cfg), str(mdl), str(out), "duration", checkpoint=tfinal, max_steps=1, log=logs.append)
    torch.cuda.synchronize()
    dfinal = os.path.join(str(out), "duration", "checkpoint_final")
    for k in ("duration_predictor", "pitch_energy_predictor", "speech_predictor"):  # the chain's last checkpoint is complete
        assert os.path.exists(os.path.join(dfinal, IO.model_file(k))), k
    assert any("duration epoch" in ln for ln in logs)
    del ctx
    # ... which is what `convert` takes (train/cli.py convert): checkpoint -> the exported text -> audio program
    res = T.convert(str(cfg), str(mdl), str(tmp_path / "export"), dfinal, log=logs.append)
    assert os.path.exists(res["program"]) and res["onnx"] is None
    ep = torch.export.load(res["program"])
    assert torch.equal(ep.state_dict["speech_predictor." + k0].cpu(), sp4)


def test_exported_program_runs_the_hip_graph(tmp_path):
    """`convert` (train/convert_to_onnx.py:23-108), the torch.export half, on the device: the ExportedProgram of
    stylish_tts_amd.export.ExportGraph (four custom ops = the four library calls of export_model.py:7-63) gives the audio of
    stylish_tts_amd.ExportModel bit for bit; saved, loaded and run with the op registry EMPTY (as in a fresh process: the ops
    rebuild their shells from the weights and the model config the program carries) it still does, also for a token string of
    another length (the dynamic axis) -- and `convert` says plainly that the ONNX half needs a package this image lacks."""
    import stylish_tts_amd as S
    from stylish_tts_amd import export as X
    from stylish_tts_amd.config import load_model_config_yaml
    from stylish_tts_amd.manifest import (duration_predictor_manifest, pitch_energy_predictor_manifest,
                                          speech_predictor_manifest)
    from stylish_tts_amd.synthetic_weights import fill_state_dict
    from tests.test_boundary import _default_model_yaml, _export_models
    mc = load_model_config_yaml(_default_model_yaml())
    models = _export_models(mc)
    models["speech_predictor"].load_state_dict(fill_state_dict(speech_predictor_manifest(), 0), strict=False)
    models["duration_predictor"].load_state_dict(fill_state_dict(duration_predictor_manifest(), 3))
    models["pitch_energy_predictor"].load_state_dict(fill_state_dict(pitch_energy_predictor_manifest(), 4))
    models = {k: m.to(DEV) for k, m in models.items()}
    logs = []
    out = X.convert(mc, str(tmp_path), models, DEV, log=logs.append)
    assert os.path.exists(out["program"]) and out["onnx"] is None and any("onnx" in ln for ln in logs)
    ref_model = S.ExportModel(**models)
    g = torch.Generator().manual_seed(5)
    for Lt in (96, 41):
        texts = torch.randint(1, 178, (1, Lt), generator=g).to(DEV)
        tl = torch.tensor([Lt], device=DEV)
        styles = [torch.rand(1, 64, generator=g).to(DEV) for _ in range(3)]
        want = ref_model(texts, tl, *styles, seed=0)
        X._REG.clear()  # a fresh process has no live shells: the ops rebuild them from what the program carries
        ep = torch.export.load(out["program"])
        got = ep.module()(texts, tl, *styles)
        torch.cuda.synchronize()
        assert got.shape == want.shape and got.numel() % 300 == 0 and bool(torch.isfinite(got).all())
        assert torch.equal(got, want), (Lt, (got - want).abs().max().item())


def _free_port():
    """a TCP port nobody listens on right now (a fixed number collides with a rendezvous socket still lingering from an
    earlier two-rank test of the same session)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("workload", ["c2", "c2-gan"])
def test_bench_two_ranks_on_one_device(workload):
    """The N > 1 path of bench.py end to end (torchrun, utterance sharding, bucketed gradient all-reduce, max-over-ranks
    timing, one JSON line on rank 0) with both ranks on device 0 and gloo as the transport (STY_BENCH_SHARE_DEVICE=1);
    c2-gan: the same with the discriminators' gradient buckets and optimizer steps in the exchange."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STY_BENCH_SHARE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--workload", workload],
                       capture_output=True, text=True, env=env, timeout=600, cwd=root)
    errs = [ln for ln in r.stderr.splitlines() if "Error" in ln or "error" in ln or "assert" in ln.lower()]
    assert r.returncode == 0, "\n".join(errs[:20]) + "\n...\n" + r.stderr[-6000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert "roofline" in rec and rec["config"]["workload"].startswith(workload)



def test_bench_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus 2` called DIRECTLY (no torchrun around it, as the driver's N=1 command line with N=2):
    the script has to start one process per rank itself and report n_gpus = 2 from a 2-rank process group."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["STY_BENCH_SHARE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--workload", "c2"], capture_output=True, text=True, env=env, timeout=600,
                       cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks"]["world_size"] == 2 and rec["value"] > 0
    assert rec["config"]["workload"].startswith("c2: B=16/GPU")


def _n3_models():
    import stylish_tts_amd as S
    from oracle.manifest import duration_predictor_manifest, pitch_energy_predictor_manifest
    from oracle.weights import fill_state_dict
    Pd = fill_state_dict(duration_predictor_manifest(), 3)
    Pp = fill_state_dict(pitch_energy_predictor_manifest(), 4)
    dp, pe = S.DurationPredictor(), S.PitchEnergyPredictor()
    dp.load_state_dict(Pd)
    pe.load_state_dict(Pp)
    return dp.to(DEV), pe.to(DEV), Pd, Pp


def test_second_stage_predictors_vs_reference_golden(env):
    """SURVEY.md 8(f) N3 on the HIP path: DurationPredictor and PitchEnergyPredictor (ProsodyEncoder, 2 heads of 160
    channels, rotary width 80; AdaptiveDecoderBlocks with learned and identity shortcuts) vs what the REFERENCE produced
    (tests/golden/n3_small.safetensors) and vs the oracle; DurationProcessor glue vs the reference's alignments."""
    import stylish_tts_amd as S
    from safetensors.torch import load_file
    from oracle import predictors as OP
    gold = load_file(os.path.join(G, "n3_small.safetensors"))
    cs = env["cs"]
    dp, pe, Pd, Pp = _n3_models()
    with torch.no_grad():
        pred = dp(dev(cs["texts"]), dev(cs["text_lengths"]), dev(gold["duration_style"]))
        proc = S.DurationProcessor(16, 50)
        dur = proc.prediction_to_duration(pred, dev(cs["text_lengths"]))
        ali = proc(pred, dev(cs["text_lengths"]))
        ali3 = proc(pred, dev(cs["text_lengths"]), multiplier=3)
        f0, en = pe(dev(cs["texts"]), dev(cs["text_lengths"]), dev(gold["alignment"]), dev(gold["pe_style"]))
        # float64 oracle for the pitch / energy stacks (their fp32 rounding is amplified, see test_oracle_golden.py)
        P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in Pp.items()}
        f64, e64 = OP.pitch_energy_predictor(P64, cs["texts"], cs["text_lengths"], gold["alignment"].double(),
                                             gold["pe_style"].double())
    torch.cuda.synchronize()
    rep = Report()
    rep.add("dur_pred vs reference", pred, gold["dur_pred"], 5e-5)
    rep.add("duration vs reference", dur, gold["duration"], 5e-5)
    rep.add("alignment vs reference", ali, gold["alignment"], 5e-5)
    rep.add("alignment x3 vs reference", ali3, gold["alignment_x3"], 5e-5)
    rep.add("pitch vs reference", f0, gold["pitch"], 1e-3)
    rep.add("energy vs reference", en, gold["energy"], 1e-3)
    rep.add("pitch vs float64 oracle", f0, f64.float(), 1e-3)
    rep.add("energy vs float64 oracle", en, e64.float(), 1e-3)
    rep.done()


def test_export_model_text_to_audio_vs_oracle(env):
    """ExportModel.forward (export_model.py:40-63) with every model on the HIP path: text -> durations -> alignment
    -> pitch / energy -> audio, vs the same chain of oracle functions fed with the HIP path's own alignment (the
    alignment's frame count depends on a rounded sum, so it is compared separately above)."""
    import stylish_tts_amd as S
    from oracle import predictors as OP, speech_predictor as osp
    cs = env["cs"]
    dp, pe, Pd, Pp = _n3_models()
    P = {k: v.clone() for k, v in env["P"].items()}
    sp = S.SpeechPredictor()
    sp.load_state_dict(P, strict=False)
    ex = S.ExportModel(speech_predictor=sp.to(DEV), pitch_energy_predictor=pe, duration_predictor=dp)
    g = torch.Generator().manual_seed(5)
    sstyle, pstyle, dstyle = (torch.randn(2, 64, generator=g) for _ in range(3))
    with torch.no_grad():
        pred = OP.duration_predictor(Pd, cs["texts"], cs["text_lengths"], dstyle)
        ali = OP.duration_to_alignment(OP.prediction_to_duration(pred, cs["text_lengths"]))
        f0, en = OP.pitch_energy_predictor(Pp, cs["texts"], cs["text_lengths"], ali, pstyle)
        T = ali.shape[2]
        gn = torch.Generator().manual_seed(9)
        noise = torch.randn(2, 300 * T, 9, generator=gn)
        ref = osp.speech_predictor(P, cs["texts"], cs["text_lengths"], ali, f0, en, (f0 > 20).float(), sstyle, f0, noise)
        audio = ex(dev(cs["texts"]), dev(cs["text_lengths"]), dev(sstyle), dev(pstyle), dev(dstyle), noise=dev(noise))
    torch.cuda.synchronize()
    assert audio.shape == (2, 300 * T), (audio.shape, T)
    mse = ((audio.cpu() - ref.squeeze(1)) ** 2).mean().item()
    print(f"\n  text -> audio, {T} frames: waveform mse vs oracle chain {mse:.3e}, mel-L1 {_mel_l1(audio.cpu().unsqueeze(1), ref):.3e}")
    assert mse <= 1e-6 and _mel_l1(audio.cpu().unsqueeze(1), ref) <= 1e-3


def test_pitch_energy_predictor_training_graph_vs_oracle_autograd(env):
    """PitchEnergyPredictor forward_train + backward (the trainable model of train_textual, stage_type.py:119-127), dropout
    off: outputs, d_style and parameter gradients (text encoder, prosody encoder incl. the 2 x 160 attention with partial
    RoPE and AdaLN, both AdaptiveDecoderBlock stacks with learned and identity shortcuts, heads) against autograd on the
    float64 oracle (pinned to the reference's forward by n3_small)."""
    import stylish_tts_amd as S
    from safetensors.torch import load_file
    from oracle import predictors as OP
    gold = load_file(os.path.join(G, "n3_small.safetensors"))
    cs = env["cs"]
    _, _, _, Pp = _n3_models()
    keys = [k for k in ("text_encoder.emb.weight", "text_encoder.proj_m.weight", "prosody_encoder.attn_layers.0.conv_q.weight",
                        "prosody_encoder.attn_layers.2.conv_k.weight", "prosody_encoder.attn_layers.1.conv_v.bias",
                        "prosody_encoder.attn_layers.1.conv_o.weight", "prosody_encoder.norm_layers_1.0.fc.weight",
                        "prosody_encoder.ffn_layers.1.conv_1.weight", "prosody_encoder.ffn_layers.2.conv_2.bias",
                        "prosody_encoder.norm_layers_2.2.fc.bias", "prosody_encoder.proj_layers.1.weight",
                        "F0.0.conv1.parametrizations.weight.original1", "F0.0.conv1x1.parametrizations.weight.original0",
                        "F0.2.conv2.parametrizations.weight.original1", "F0.3.norm1.fc.weight", "N.1.conv1.bias",
                        "N.3.conv2.parametrizations.weight.original0", "F0_proj.weight", "N_proj.bias") if k in Pp]
    assert len(keys) >= 12, [k for k in Pp][:40]
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in Pp.items()}
    for k in keys:
        P64[k].requires_grad_(True)
    style64 = gold["pe_style"].double().requires_grad_(True)
    f64, e64 = OP.pitch_energy_predictor(P64, cs["texts"], cs["text_lengths"], gold["alignment"].double(), style64)
    g = torch.Generator().manual_seed(4)
    s1, s2 = torch.randn(f64.shape, generator=g), torch.randn(e64.shape, generator=g)
    (f64 * s1.double()).sum().backward(retain_graph=True)
    (e64 * s2.double()).sum().backward()
    pe = S.PitchEnergyPredictor()
    pe.load_state_dict(Pp)
    pe = pe.to(DEV).enable_training()
    f0, en = pe.forward_train(dev(cs["texts"]), dev(cs["text_lengths"]), dev(gold["alignment"]), dev(gold["pe_style"]))
    d_style = pe.backward(dev(s1), dev(s2))
    torch.cuda.synchronize()
    rep = Report()
    rep.add("pitch", f0, f64.detach().float(), 1e-3)
    rep.add("energy", en, e64.detach().float(), 1e-3)
    # gate: the fp32 ORACLE's own gradients sit 1.5-2.5e-2 (of the tensor maximum) from the float64 ones on this graph
    # (four AdaIN blocks behind three AdaLN layers: the conditioning noted in DESIGN.md section 2), so 5e-2 as for the
    # acoustic step; biases in front of an instance norm have a structurally zero gradient and are skipped
    rep.add("d_style", d_style, style64.grad.float(), 5e-2)
    nm = dict(pe.named_parameters())
    for k in keys:
        ref = P64[k].grad.float()
        if ref.abs().max().item() < 1e-9:
            assert nm[k].grad.abs().max().item() < 1e-4, k
            continue
        rep.add("d " + k[-44:], nm[k].grad, ref, 5e-2)
    rep.done()


def test_duration_predictor_training_graph_vs_oracle_autograd(env):
    """DurationPredictor forward_train + backward (the trainable model of train_duration, stage_type.py:495-556), dropout
    off: output, d_style and parameter gradients (cross attention between two AdaLN views, weight-normed depthwise conv,
    AdaptiveConvNeXt blocks with GELU + GRN, the cumulative class head) against autograd on the float64 oracle."""
    import stylish_tts_amd as S
    from safetensors.torch import load_file
    from oracle import predictors as OP
    gold = load_file(os.path.join(G, "n3_small.safetensors"))
    cs = env["cs"]
    _, _, Pd, _ = _n3_models()
    want_keys = ("text_encoder.proj_m.weight", "query_norm.fc.weight", "key_norm.fc.bias", "cross_attention.conv_q.weight",
                 "cross_attention.conv_k.weight", "cross_attention.conv_v.bias", "cross_attention.conv_o.weight",
                 "cross_post.0.parametrizations.weight.original0", "cross_post.0.parametrizations.weight.original1",
                 "cross_post.0.bias", "cross_post.2.parametrizations.weight.original1", "conv_next.0.dwconv.weight",
                 "conv_next.1.pwconv1.weight", "conv_next.2.pwconv2.weight", "conv_next.1.grn.gamma", "conv_next.1.grn.beta",
                 "conv_next.0.norm.fc.weight", "duration_proj.linear_layer.weight", "duration_proj.linear_layer.bias")
    keys = [k for k in want_keys if k in Pd]
    assert len(keys) >= 10, sorted(Pd)[:60]
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in Pd.items()}
    for k in keys:
        P64[k].requires_grad_(True)
    style64 = gold["duration_style"].double().requires_grad_(True)
    out64 = OP.duration_predictor(P64, cs["texts"], cs["text_lengths"], style64)
    seed = torch.randn(out64.shape, generator=torch.Generator().manual_seed(6))
    (out64 * seed.double()).sum().backward()
    dp = S.DurationPredictor()
    dp.load_state_dict(Pd)
    dp = dp.to(DEV).enable_training()
    out = dp.forward_train(dev(cs["texts"]), dev(cs["text_lengths"]), dev(gold["duration_style"]))
    d_style = dp.backward(dev(seed))
    torch.cuda.synchronize()
    rep = Report()
    rep.add("out", out, out64.detach().float(), 5e-5)
    rep.add("d_style", d_style, style64.grad.float(), 5e-4)
    nm = dict(dp.named_parameters())
    for k in keys:
        ref = P64[k].grad.float()
        if ref.abs().max().item() < 1e-9:
            continue
        rep.add("d " + k[-44:], nm[k].grad, ref, 5e-4)
    rep.done()


def test_duration_train_step_vs_oracle(env):
    """train_duration (stage_type.py:495-556) assembled: trainable duration_style_encoder + duration_predictor,
    prediction_to_duration, the per-utterance smooth-L1 and the weighted cross entropy with LossLog normalisation, the
    dur_disc generator term; lr = 0.  Losses, predicted durations and gradients against (1) what the REFERENCE's own
    train_duration logged and left on the parameters (tests/golden/stages_small, tools/gen_golden_stages.py) and (2)
    autograd on the oracle's assembly (oracle/stages.py, itself pinned to the same fixture)."""
    import stylish_tts_amd as S
    from oracle import stages
    from stylish_tts_amd.discriminators import PitchDiscriminator
    from stylish_tts_amd.duration import DurationTrainer
    from tests.test_oracle_golden import STAGE_KEYS, stage_inputs, stage_sub
    fx, P, cs = stage_inputs()
    Pdp, Pse, Pd = P["dp"], P["dse"], P["dur_disc"]
    dp_keys, se_keys = STAGE_KEYS["dp"], STAGE_KEYS["se"]
    for k in dp_keys:
        Pdp[k].requires_grad_(True)
    for k in se_keys:
        Pse[k].requires_grad_(True)
    audio_gt, weights, tlen = fx["audio_gt"], fx["class_weights"], cs["text_lengths"]
    assert torch.equal(audio_gt, _test_audio(2, audio_gt.shape[1], 21))
    olog, total, duration = stages.train_duration(Pdp, Pse, Pd, audio_gt, cs["texts"], tlen, cs["durations"], weights)
    total.backward()
    l_dur, l_ce, l_gen = olog["duration"], olog["duration_ce"], olog["generator"]
    # ---- HIP ----
    def shell(cls, P):
        m = cls()
        m.load_state_dict({k: v.detach() for k, v in P.items()})
        return m.to(DEV)

    dd = PitchDiscriminator(dim_in=1, kernel=5)
    dd.load_state_dict(Pd)
    tr = DurationTrainer(shell(S.DurationPredictor, Pdp), shell(S.MelStyleEncoder, Pse), dd.to(DEV), weights, lr=0.0,
                         train_mode=False)
    log = tr.train_batch(audio_gt=dev(audio_gt), texts=dev(cs["texts"]), text_lengths=dev(tlen), durations=dev(cs["durations"]))
    torch.cuda.synchronize()
    print(f"\n  duration {log['duration'].item():.5f} vs {l_dur.item():.5f}  ce {log['duration_ce'].item():.5f} vs "
          f"{l_ce.item():.5f}  generator {log['generator'].item():.5f} vs {l_gen.item():.5f}")
    for k in ("duration", "duration_ce", "generator"):
        assert abs(log[k].item() - olog[k].item()) <= 1e-4 * olog[k].item()
        ref = fx["duration.log." + k].item()  # the reference's own LossLog
        assert abs(log[k].item() - ref) <= 1e-4 * ref, (k, log[k].item(), ref)
    rep = Report()
    rep.add("duration", tr.duration, duration.detach(), 1e-4)
    rep.add("duration (reference)", tr.duration, fx["duration.pred_duration"], 1e-4)
    ndp, nse = dict(tr.dp.named_parameters()), dict(tr.se.named_parameters())
    for k in dp_keys:
        rep.add("d dp." + k[-40:], ndp[k].grad, Pdp[k].grad, 3e-4)
        rep.add("d dp(ref)." + k[-36:], stage_sub(ndp[k].grad), fx["duration.grad.dp." + k], 4e-4)
    for k in se_keys:
        rep.add("d se." + k[-40:], nse[k].grad, Pse[k].grad, 3e-4)
        rep.add("d se(ref)." + k[-36:], stage_sub(nse[k].grad), fx["duration.grad.se." + k], 4e-4)
    rep.done()


@pytest.mark.parametrize("which", ["duration", "pitch_energy"])
def test_second_stage_predictors_train_mode_dropouts_vs_oracle(env, which):
    """Training-mode dropouts of the two trainable predictors -- text encoder sites, attention probabilities (0.5 / 0.2),
    DropPath(0.5) per AdaptiveConvNeXt block and Dropout1d(0.5) after it (duration_predictor.py:25-40, 79), the prosody
    encoder's three sites per layer (prosody_encoder.py:72-78), Dropout(0.2) in front of both convs of the pitch / energy
    stacks' AdaptiveDecoderBlocks (ada_norm.py:172-179) -- with the counter-based hash masks: outputs, d_style and
    parameter gradients against (1) the REFERENCE run in .train() with its random masks patched to the same function
    (tests/golden/n3_dropout_small, tools/gen_golden_dropout.py) and (2) autograd on the float64 oracle."""
    import stylish_tts_amd as S
    from safetensors.torch import load_file
    from tests.test_oracle_golden import DROPOUT_KEYS, oracle_predictor_with_dropout, stage_sub
    gold = load_file(os.path.join(G, "n3_small.safetensors"))
    fx = load_file(os.path.join(G, "n3_dropout_small.safetensors"))
    cs = env["cs"]
    _, _, Pd, Pp = _n3_models()
    P = Pd if which == "duration" else Pp
    keys = DROPOUT_KEYS[which]
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    for k in keys:
        P64[k].requires_grad_(True)
    style_name = "duration_style" if which == "duration" else "pe_style"
    style64 = gold[style_name].double().requires_grad_(True)
    outs, nsites = oracle_predictor_with_dropout(which, P64, cs, gold, style64, torch.float64)
    assert nsites == int(fx[which + ".nsites"].item())
    g = torch.Generator().manual_seed(8)
    seeds = [torch.randn(o.shape, generator=g) for o in outs]
    sum((o * s_.double()).sum() for o, s_ in zip(outs, seeds)).backward()
    m = (S.DurationPredictor() if which == "duration" else S.PitchEnergyPredictor())
    m.load_state_dict(P)
    m = m.to(DEV).enable_training()
    m.set_train_opts(dropout_seed=4321, text_dropout=0.2, block_dropout=0.2)
    if which == "duration":
        got = (m.forward_train(dev(cs["texts"]), dev(cs["text_lengths"]), dev(gold["duration_style"])),)
        d_style = m.backward(dev(seeds[0]))
    else:
        got = m.forward_train(dev(cs["texts"]), dev(cs["text_lengths"]), dev(gold["alignment"]), dev(gold["pe_style"]))
        d_style = m.backward(dev(seeds[0]), dev(seeds[1]))
    torch.cuda.synchronize()
    rep = Report()
    tol_o, tol_g = (1e-4, 1e-3) if which == "duration" else (2e-3, 8e-2)
    for i, (a_, b_) in enumerate(zip(got, outs)):
        rep.add(f"out{i}", a_, b_.detach().float(), tol_o)
        rep.add(f"out{i} (reference)", a_, fx[f"{which}.out{i}"], tol_o)
    rep.add("d_style", d_style, style64.grad.float(), tol_g)
    rep.add("d_style (reference)", d_style, fx[which + ".d_style"], tol_g)
    nm = dict(m.named_parameters())
    for k in keys:
        rep.add("d " + k[-44:], nm[k].grad, P64[k].grad.float(), tol_g)
        rep.add("d(ref) " + k[-40:], stage_sub(nm[k].grad), fx[f"{which}.grad.{k}"], tol_g)
    rep.done()


def test_textual_train_step_vs_oracle(env):
    """train_textual (stage_type.py:415-450) assembled: trainable pe_style_encoder + pitch_energy_predictor, the frozen
    speech predictor / style encoder carrying d loss / d (pitch, energy) back from the mel loss, pitch / energy losses, the
    pitch_disc generator term, LossLog normalisation; lr = 0 so that parameters stay put.  Losses, predictions and
    gradients of the two trained models against (1) the REFERENCE's own train_textual (tests/golden/stages_small,
    tools/gen_golden_stages.py) and (2) autograd on the oracle's assembly (oracle/stages.py, pinned to the same fixture).
    Gradient gates as in the acoustic step: fp32 conditioning behind ~20 normalisation layers (the fp32 oracle itself
    sits up to 1.4e-2 from the fp32 reference; the deepest tensor listed, the prosody encoder's conv_q, measured 8.4e-2
    against the oracle and 6.5e-2 against the reference's own gradient in round 5 -- two fp32 evaluations of one graph, a
    summation order apart: gate 1e-1)."""
    import stylish_tts_amd as S
    from oracle import stages
    from stylish_tts_amd.discriminators import PitchDiscriminator
    from stylish_tts_amd.textual import TextualTrainer
    from tests.test_oracle_golden import STAGE_KEYS, stage_inputs, stage_sub
    fx, P, cs = stage_inputs()
    Ppep, Ppse, Psp, Pse, Pd = P["pep"], P["pse"], P["sp"], P["se"], P["pitch_disc"]
    pep_keys, pse_keys = STAGE_KEYS["pep"], STAGE_KEYS["pse"]
    for k in pep_keys:
        Ppep[k].requires_grad_(True)
    for k in pse_keys:
        Ppse[k].requires_grad_(True)
    audio_gt, pitch = fx["audio_gt"], cs["pitch"]
    want = {}
    olog, total, cat_p = stages.train_textual(Ppep, Ppse, Psp, Pse, Pd, audio_gt, cs["texts"], cs["text_lengths"], pitch,
                                              cs["durations"], cs["noise"], want)
    total.backward()
    # ---- HIP ----
    def shell(cls, P, **kw):
        m = cls(**kw)
        m.load_state_dict({k: v.detach() for k, v in P.items()}, strict=False)
        return m.to(DEV)

    pd = PitchDiscriminator(dim_in=2, kernel=21)
    pd.load_state_dict(Pd)
    tr = TextualTrainer(shell(S.PitchEnergyPredictor, Ppep), shell(S.PitchStyleEncoder, Ppse), shell(S.SpeechPredictor, Psp),
                        shell(S.MelStyleEncoder, Pse), pd.to(DEV), lr=0.0, train_mode=False)
    log = tr.train_batch(audio_gt=dev(audio_gt), texts=dev(cs["texts"]), text_lengths=dev(cs["text_lengths"]),
                         pitch=dev(pitch), durations=dev(cs["durations"]), noise=dev(cs["noise"]),
                         prior_override=dev(want["prior"]))
    torch.cuda.synchronize()
    print("\n  " + "  ".join(f"{k} {log[k].item():.5f} vs {olog[k].item():.5f} (reference {fx['textual.log.' + k].item():.5f})"
                            for k in ("mel", "pitch", "energy", "generator")))
    for k in ("mel", "pitch", "energy", "generator"):
        assert abs(log[k].item() - olog[k].item()) <= 2e-3 * olog[k].item()
        ref = fx["textual.log." + k].item()  # the reference's own LossLog
        assert abs(log[k].item() - ref) <= 2e-3 * ref, (k, log[k].item(), ref)
    rep = Report()
    got_cat = torch.stack([tr.pred_pitch * (dev(pitch) > 10).float(), tr.pred_energy], 1)
    rep.add("pred pitchcat", got_cat, cat_p.detach(), 2e-3)
    rep.add("pred pitchcat (reference)", got_cat, fx["textual.pred_pitchcat"], 2e-3)
    npep, npse = dict(tr.pep.named_parameters()), dict(tr.pse.named_parameters())
    for k in pep_keys:
        rep.add("d pep." + k[-40:], npep[k].grad, Ppep[k].grad, 1e-1)
        rep.add("d pep(ref)." + k[-36:], stage_sub(npep[k].grad), fx["textual.grad.pep." + k], 1e-1)
    for k in pse_keys:
        rep.add("d pse." + k[-40:], npse[k].grad, Ppse[k].grad, 8e-2)
        rep.add("d pse(ref)." + k[-36:], stage_sub(npse[k].grad), fx["textual.grad.pse." + k], 8e-2)
    rep.done()


def test_pitch_style_encoder_training_graph_vs_oracle_autograd():
    """PitchStyleEncoder forward_train + backward (the trainable `pe_style_encoder` of train_textual, stage_type.py:
    119-121): style and parameter gradients -- preconv g / v / bias through the weight_norm chain and trunk parameters --
    against autograd on the oracle (itself pinned to the reference's forward by pse_small)."""
    import stylish_tts_amd as S
    from safetensors.torch import load_file
    from oracle import predictors as OP
    from oracle.manifest import pitch_style_encoder_manifest
    from oracle.weights import fill_state_dict
    g = load_file(os.path.join(G, "pse_small.safetensors"))
    P = fill_state_dict(pitch_style_encoder_manifest(), 5)
    keys = ["preconv.parametrizations.weight.original0", "preconv.parametrizations.weight.original1", "preconv.bias",
            "shared.0.weight_orig", "shared.2.conv1.weight_orig", "shared.4.conv2.bias", "unshared.weight", "unshared.bias"]
    Pr = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in P.items()}
    s_ref = OP.pitch_style_encoder(Pr, g["mel"], g["pitch"], g["energy"])
    seed = torch.randn(s_ref.shape, generator=torch.Generator().manual_seed(2))
    (s_ref * seed).sum().backward()
    m = S.PitchStyleEncoder()
    m.load_state_dict(P)
    m = m.to(DEV).enable_training()
    s = m.forward_train(dev(g["mel"]), dev(g["pitch"]), dev(g["energy"]))
    m.backward(dev(seed))
    torch.cuda.synchronize()
    rep = Report()
    rep.add("style", s, s_ref.detach(), 2e-5)
    nm = dict(m.named_parameters())
    for k in keys:
        rep.add("d " + k[-40:], nm[k].grad, Pr[k].grad, 5e-5)
    rep.done()


def test_pitch_style_encoder_vs_reference_golden():
    """PitchStyleEncoder (the pe_style_encoder of build_model) on the HIP path vs the reference's output
    (tests/golden/pse_small.safetensors): 1x1 weight-normed preconv with padding 1, then the MelStyleEncoder plan."""
    import stylish_tts_amd as S
    from safetensors.torch import load_file
    from oracle.manifest import pitch_style_encoder_manifest
    from oracle.weights import fill_state_dict
    g = load_file(os.path.join(G, "pse_small.safetensors"))
    m = S.PitchStyleEncoder()
    m.load_state_dict(fill_state_dict(pitch_style_encoder_manifest(), 5))
    m = m.to(DEV)
    with torch.no_grad():
        s = m(dev(g["mel"]), dev(g["pitch"]), dev(g["energy"]))
    torch.cuda.synchronize()
    rep = Report()
    rep.add("style vs reference", s, g["style"], 2e-5)
    rep.done()
