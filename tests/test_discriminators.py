"""Spectrogram discriminators + adversarial loss helpers.

CPU: oracle/discriminator.py against the fixture written by the reference's own classes (tools/gen_golden_disc.py).
GPU: the HIP path (through the C ABI) against the same fixture and against the oracle on other shapes.
"""
import json
import os

import pytest
import torch
from safetensors.torch import load_file

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fixture():
    fx = load_file(os.path.join(G, "disc_small.safetensors"))
    params = {k[2:]: v for k, v in fx.items() if k.startswith("w.")}
    return fx, params


def test_spec_discriminator_manifest_equals_reference_state_dict():
    from stylish_tts_amd.discriminators import spec_discriminator_manifest
    with open(os.path.join(G, "manifest_spec_discriminator.json")) as f:
        want = json.load(f)
    assert spec_discriminator_manifest() == want


def test_discriminator_oracle_matches_reference_classes():
    """oracle/discriminator.py vs SpecDiscriminator / GeneratorLossHelper / DiscriminatorLossHelper of the reference:
    score maps, both losses, d loss / d pred, d loss / d parameters."""
    from oracle import discriminator as od
    fx, params = _fixture()
    for case in range(2):
        t = fx[f"c{case}.target"]
        q = fx[f"c{case}.pred"].clone().requires_grad_(True)
        rs, gs = od.spec_discriminator(params, t), od.spec_discriminator(params, q)
        for i in range(5):
            assert (rs[i] - fx[f"c{case}.real_score{i}"]).abs().max().item() <= 2e-6
            assert (gs[i] - fx[f"c{case}.gen_score{i}"]).abs().max().item() <= 2e-6
        gl = od.generator_loss_helper(rs, gs)
        gl.backward()
        assert abs(gl.item() - fx[f"c{case}.gen_loss"].item()) <= 1e-5
        ref = fx[f"c{case}.d_pred"]
        assert (q.grad - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
        pp = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        dl = od.discriminator_loss_helper(od.spec_discriminator(pp, t), od.spec_discriminator(pp, fx[f"c{case}.pred"]))
        dl.backward()
        assert abs(dl.item() - fx[f"c{case}.disc_loss"].item()) <= 1e-5
        for k in pp:
            ref = fx[f"c{case}.grad." + k]
            assert (pp[k].grad - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1e-3), k


def _hip_model(params, dev):
    from stylish_tts_amd.discriminators import SpecDiscriminator
    m = SpecDiscriminator().to(dev)
    missing = m.load_state_dict(params, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m


@pytest.mark.gpu
def test_spec_discriminator_hip_vs_reference_fixture():
    """HIP SpecDiscriminator.forward, GeneratorLossHelper and DiscriminatorLossHelper against the values the reference's
    classes produced (fp32 mode): score maps, both loss values, the tracked loss; gradients through the pinned oracle."""
    from stylish_tts_amd.discriminators import DiscriminatorLossHelper, GeneratorLossHelper
    dev = torch.device("cuda:0")
    fx, params = _fixture()
    m = _hip_model(params, dev)
    for case in range(2):
        t, q = fx[f"c{case}.target"].to(dev), fx[f"c{case}.pred"].to(dev)
        rs, fm = m(t)
        gs, _ = m(q)
        assert fm == []
        for i in range(5):
            for got, name in ((rs[i], "real"), (gs[i], "gen")):
                ref = fx[f"c{case}.{name}_score{i}"]
                assert got.shape == ref.shape
                err = (got.cpu() - ref).abs().max().item()
                assert err <= 2e-5 * max(ref.abs().max().item(), 0.1), (case, i, name, err)
        gl = GeneratorLossHelper(m)(target=t, pred=q)
        assert abs(gl.item() - fx[f"c{case}.gen_loss"].item()) <= 2e-5 * fx[f"c{case}.gen_loss"].item()
        for p in m.parameters():
            p.grad = None
        helper = DiscriminatorLossHelper(m, 5)
        dl = helper(target=t, pred=q)
        assert abs(dl.item() - fx[f"c{case}.disc_loss"].item()) <= 2e-5 * fx[f"c{case}.disc_loss"].item()
        assert abs(helper.last_loss - fx[f"c{case}.last_loss"].item()) <= 1e-5
        # gradients: the oracle (pinned to the reference's gradients on this very fixture by the CPU test above) fed with
        # the HIP score maps -- see _check_against_oracle for why not the fixture's gradient tensors directly
        _check_against_oracle(m, params, fx[f"c{case}.target"], fx[f"c{case}.pred"], dev, 2e-5, 2e-4)


def _check_against_oracle(m, params, t, q0, dev, tol_x, tol_w, tol_loss=2e-5):
    """The relativistic term puts a large share of its gradient on ONE score element (the median of real - gen, as
    torch.median's backward does), so score maps that differ in the last bit can move that spike to another element.
    The comparison is therefore made in two exact halves: (1) the loss kernels against the oracle's loss functions ON
    THE HIP SCORE MAPS (same inputs bit for bit -> same median element): values here, gradients through (2);
    (2) the network backward against the oracle network's backward fed with those score gradients."""
    from oracle import discriminator as od
    B, _, H, W = t.shape
    td, qd = t.to(dev), q0.to(dev)
    rs_h = [x.cpu().clone().requires_grad_(True) for x in m(td)[0]]
    gs_h = [x.cpu().clone().requires_grad_(True) for x in m(qd)[0]]
    gl = od.generator_loss_helper(rs_h, gs_h)
    g_gen = torch.autograd.grad(gl, gs_h)
    dl = od.discriminator_loss_helper(rs_h, gs_h)
    g_dr = torch.autograd.grad(dl, rs_h, retain_graph=True)
    g_dg = torch.autograd.grad(dl, gs_h)
    # oracle network, score gradients injected
    q = q0.clone().requires_grad_(True)
    pp = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    so_t, so_q = od.spec_discriminator(pp, t), od.spec_discriminator(pp, q)
    for a_, b_ in zip(so_t + so_q, rs_h + gs_h):  # the HIP score maps themselves
        assert (a_ - b_).abs().max().item() <= tol_x * max(a_.abs().max().item(), 0.1)
    dq, = torch.autograd.grad(so_q, q, grad_outputs=list(g_gen), retain_graph=True)
    keys = sorted(pp)
    dw = torch.autograd.grad(so_t + so_q, [pp[k] for k in keys], grad_outputs=list(g_dr) + list(g_dg))
    for p_ in m.parameters():
        p_.grad = None
    d_pred = torch.zeros(B, H, W, device=dev)
    gen, disc = m.losses(td, qd, gen_scale=2.0, d_pred=d_pred, disc_scale=3.0)
    assert abs(gen[0].item() - gl.item()) <= tol_loss * gl.item(), (gen[0].item(), gl.item())
    assert abs(disc[0].item() - dl.item()) <= tol_loss * dl.item(), (disc[0].item(), dl.item())
    err = (d_pred.cpu() - 2.0 * dq[:, 0]).abs().max().item()
    assert err <= tol_w * 2.0 * dq.abs().max().item(), ("d_pred", err, dq.abs().max().item())
    got = dict(m.named_parameters())
    for k, ref in zip(keys, dw):
        ref = 3.0 * ref
        err = (got[k].grad.cpu() - ref).abs().max().item()
        lim = tol_w * (5.0 if k.endswith("original0") else 1.0) * max(ref.abs().max().item(), 1e-3)
        assert err <= lim, (k, err, ref.abs().max().item())
    return gen, disc, d_pred


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(3, 65, 203), (2, 129, 77), (1, 17, 9)])
def test_spec_discriminator_hip_vs_oracle_other_shapes(B, H, W):
    """Widths that are odd at every halving, a single-tile image, and the combined call (both helpers from one forward
    pass) against the oracle; the combined call must equal the two separate calls."""
    dev = torch.device("cuda:0")
    _, params = _fixture()
    m = _hip_model(params, dev)
    g = torch.Generator().manual_seed(B * 1000 + W)
    t = torch.rand(B, 1, H, W, generator=g) ** 2 * 3
    q0 = (t + 0.4 * torch.randn(B, 1, H, W, generator=g)).abs()
    gen, disc, d_pred = _check_against_oracle(m, params, t, q0, dev, 2e-5, 2e-4)
    for p in m.parameters():
        p.grad = None
    d2 = torch.zeros(B, H, W, device=dev)
    gen2, _ = m.losses(t.to(dev), q0.to(dev), gen_scale=2.0, d_pred=d2)
    _, disc2 = m.losses(t.to(dev), q0.to(dev), disc_scale=3.0)
    assert torch.equal(gen2, gen) and torch.equal(d2, d_pred)
    assert abs(disc2[0].item() - disc[0].item()) <= 1e-6 * abs(disc[0].item())


@pytest.mark.gpu
def test_spec_discriminator_bf16_mode():
    """compute_bf16: the 32 -> 32 convs and their gradient convs multiply bf16-rounded operands (convp16 / wgrad kernels
    of the acoustic path); same two-part check as in fp32 mode, bf16-sized tolerances."""
    dev = torch.device("cuda:0")
    _, params = _fixture()
    m = _hip_model(params, dev)
    m.compute_bf16 = True
    g = torch.Generator().manual_seed(5)
    t = torch.rand(4, 1, 129, 160, generator=g) ** 2 * 3
    q0 = (t + 0.4 * torch.randn(t.shape, generator=g)).abs()
    _check_against_oracle(m, params, t, q0, dev, 2e-2, 5e-2)


@pytest.mark.gpu
def test_acoustic_loss_with_adversarial_term_composes_the_pinned_pieces():
    """sty_acoustic_gan_loss_fwd_bwd = the acoustic losses + the "mrd" generator term + the discriminator side, all from
    one pass.  Checked against the separately pinned pieces: mel / phase losses and their seed (sty_acoustic_loss_fwd_bwd),
    the three discriminators on MultiSpectrogram's |X| (sty_specdisc_losses), and the oracle's STFT (torch.stft) for the
    vector-Jacobian product that carries d loss / d |X| back to the waveform."""
    from oracle import frontend as of
    from stylish_tts_amd.frontend import MultiSpectrogram
    from stylish_tts_amd.losses import acoustic_gan_loss, acoustic_loss
    dev = torch.device("cuda:0")
    _, params = _fixture()
    g = torch.Generator().manual_seed(11)
    mrd = []
    for r in range(3):
        p = {k: (v * (1.0 + 0.1 * torch.randn(v.shape, generator=g)) if k.endswith("original1") else v.clone())
             for k, v in params.items()}
        mrd.append(_hip_model(p, dev))
    B, N = 2, 9600
    gt = 0.3 * torch.randn(B, N, generator=g)
    pr = gt + 0.1 * torch.randn(B, N, generator=g)
    gtd, prd = gt.to(dev), pr.to(dev)
    w_gen, dscale = 0.7, 1.5
    losses0, d0 = acoustic_loss(gtd, prd, 5.0, 8.0)
    for m in mrd:
        for p_ in m.parameters():
            p_.grad = None
    losses, gan, d = acoustic_gan_loss(gtd, prd, mrd, w_mel=5.0, w_phase=8.0, w_gen=w_gen, disc_scale=dscale, step=(1,))
    assert torch.allclose(losses, losses0, rtol=1e-6, atol=0)
    got_grads = {k: v.grad.clone() for k, v in mrd[1].named_parameters()}
    assert all(p_.grad is None for p_ in mrd[0].parameters()) and all(p_.grad is None for p_ in mrd[2].parameters())
    # the pieces
    _, _, _, _, t_fft, p_fft = MultiSpectrogram()(target=gtd, pred=prd)
    gen_sum, vjp = 0.0, torch.zeros(B, N)
    prc = pr.clone().requires_grad_(True)
    for r, (fft, hop, win) in enumerate(of.RESOLUTIONS):
        for p_ in mrd[r].parameters():
            p_.grad = None
        dx = torch.zeros_like(p_fft[r][:, 0])
        gen, disc = mrd[r].losses(t_fft[r], p_fft[r], gen_scale=w_gen, d_pred=dx, disc_scale=dscale)
        gen_sum += gen[0].item()
        assert abs(gan[1 + 2 * r].item() - disc[0].item()) <= 1e-5 * abs(disc[0].item())
        assert abs(gan[2 + 2 * r].item() - disc[1].item()) <= 1e-5 * abs(disc[1].item())
        if r == 1:
            for k, v in mrd[1].named_parameters():
                err = (v.grad - got_grads[k]).abs().max().item()
                assert err <= 1e-4 * max(v.grad.abs().max().item(), 1e-3), (k, err)
        _, _, fm = of.multi_spectrogram_single(prc, fft, hop, win)
        vjp += torch.autograd.grad(fm, prc, grad_outputs=dx.cpu().unsqueeze(1))[0]
    assert abs(gan[0].item() - gen_sum) <= 1e-5 * abs(gen_sum)
    diff = (d - d0).cpu()
    err = (diff - vjp).abs().max().item()
    assert err <= 2e-4 * vjp.abs().max().item() + 1e-6 * d0.abs().max().item(), (err, vjp.abs().max().item())


def _cf_fixture():
    fx = load_file(os.path.join(G, "cfdisc_small.safetensors"))
    params = {k[2:]: v for k, v in fx.items() if k.startswith("w.")}
    return fx, params


def test_context_free_discriminator_oracle_and_manifest_match_reference():
    """oracle.context_free_discriminator vs the reference's ContextFreeDiscriminator in training mode: score maps, both
    helpers' losses and gradients; the shell's key table equals the reference state_dict."""
    from oracle import discriminator as od
    from stylish_tts_amd.discriminators import context_free_discriminator_manifest
    with open(os.path.join(G, "manifest_context_free_discriminator.json")) as f:
        assert context_free_discriminator_manifest() == json.load(f)
    fx, p = _cf_fixture()
    t, q = fx["target"], fx["pred"].clone().requires_grad_(True)
    rs, gs = od.context_free_discriminator(p, t), od.context_free_discriminator(p, q)
    assert (rs - fx["real_score"]).abs().max().item() <= 1e-6 and (gs - fx["gen_score"]).abs().max().item() <= 1e-6
    gl = od.generator_loss_helper([rs], [gs])
    gl.backward()
    assert abs(gl.item() - fx["gen_loss"].item()) <= 1e-6
    assert (q.grad - fx["d_pred"]).abs().max().item() <= 1e-5 * fx["d_pred"].abs().max().item()
    pp = {k: (v.clone().requires_grad_(True) if ("grad." + k) in fx else v) for k, v in p.items()}
    dl = od.discriminator_loss_helper([od.context_free_discriminator(pp, t)], [od.context_free_discriminator(pp, fx["pred"])])
    dl.backward()
    assert abs(dl.item() - fx["disc_loss"].item()) <= 1e-6
    for k in pp:
        if ("grad." + k) in fx:
            ref = fx["grad." + k]
            # (a bias in front of a BatchNorm has a structurally zero gradient: |ref| ~ 1e-8 of rounding noise on both sides,
            # whose size depends on the host's conv kernels -- the floor keeps such a tensor from being compared to itself)
            assert (pp[k].grad - ref).abs().max().item() <= 1e-4 * max(ref.abs().max().item(), 1e-3), k


def _cf_model(params, dev):
    from stylish_tts_amd.discriminators import ContextFreeDiscriminator
    m = ContextFreeDiscriminator()
    m.load_state_dict(params, strict=True)
    return m.to(dev)


def _cf_check(m, params, t, q0, dev, tol_x, tol_w):
    """same two-part scheme as _check_against_oracle (the relativistic term's median spike)"""
    from oracle import discriminator as od
    B, N = t.shape
    td, qd = t.to(dev), q0.to(dev)
    rs_h = m(td)[0][0].cpu().clone().requires_grad_(True)
    gs_h = m(qd)[0][0].cpu().clone().requires_grad_(True)
    gl = od.generator_loss_helper([rs_h], [gs_h])
    g_gen, = torch.autograd.grad(gl, gs_h)
    dl = od.discriminator_loss_helper([rs_h], [gs_h])
    g_dr, = torch.autograd.grad(dl, rs_h, retain_graph=True)
    g_dg, = torch.autograd.grad(dl, gs_h)
    q = q0.clone().requires_grad_(True)
    keys = sorted(k for k, v in params.items() if v.is_floating_point() and "running" not in k)
    pp = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in params.items()}
    so_t, so_q = od.context_free_discriminator(pp, t), od.context_free_discriminator(pp, q)
    for a_, b_ in ((so_t, rs_h), (so_q, gs_h)):
        assert (a_ - b_).abs().max().item() <= tol_x * max(a_.abs().max().item(), 0.1), (a_ - b_).abs().max().item()
    dq, = torch.autograd.grad(so_q, q, grad_outputs=g_gen, retain_graph=True)
    dw = torch.autograd.grad([so_t, so_q], [pp[k] for k in keys], grad_outputs=[g_dr, g_dg])
    for p_ in m.parameters():
        p_.grad = None
    d_pred = torch.zeros(B, N, device=dev)
    gen, disc = m.losses(td, qd, gen_scale=2.0, d_pred=d_pred, disc_scale=3.0)
    assert abs(gen[0].item() - gl.item()) <= 5e-5 * gl.item(), (gen[0].item(), gl.item())
    assert abs(disc[0].item() - dl.item()) <= 5e-5 * dl.item(), (disc[0].item(), dl.item())
    err = (d_pred.cpu() - 2.0 * dq).abs().max().item()
    assert err <= tol_w * 2.0 * dq.abs().max().item(), ("d_pred", err, dq.abs().max().item())
    got = dict(m.named_parameters())
    for k, ref in zip(keys, dw):
        ref = 3.0 * ref
        err = (got[k].grad.cpu() - ref).abs().max().item()
        assert err <= tol_w * max(ref.abs().max().item(), 1e-3), (k, err, ref.abs().max().item())


@pytest.mark.gpu
def test_context_free_discriminator_hip_vs_reference_fixture():
    """HIP ContextFreeDiscriminator (training mode) against the reference's values: score maps, both loss values, BatchNorm
    running statistics after two forwards; gradients through the pinned oracle."""
    dev = torch.device("cuda:0")
    fx, params = _cf_fixture()
    m = _cf_model(params, dev)
    t, q = fx["target"], fx["pred"]
    rs = m(t.to(dev))[0][0].cpu()
    gs = m(q.to(dev))[0][0].cpu()
    for got, ref in ((rs, fx["real_score"]), (gs, fx["gen_score"])):
        assert got.shape == ref.shape
        assert (got - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 0.1), (got - ref).abs().max().item()
    sd = m.state_dict()
    for k, ref in fx.items():
        if k.startswith("after2."):
            got = sd[k[7:]].cpu()
            assert (got.float() - ref.float()).abs().max().item() <= 1e-5 * max(ref.float().abs().max().item(), 1.0), k
    m = _cf_model(params, dev)
    gen, _ = m.losses(t.to(dev), q.to(dev), gen_scale=1.0)
    assert abs(gen[0].item() - fx["gen_loss"].item()) <= 5e-5 * fx["gen_loss"].item()
    _, disc = m.losses(t.to(dev), q.to(dev), disc_scale=1.0)
    assert abs(disc[0].item() - fx["disc_loss"].item()) <= 5e-5 * fx["disc_loss"].item()
    _cf_check(_cf_model(params, dev), params, t, q, dev, 2e-5, 5e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(3, 5000), (1, 1024)])
def test_context_free_discriminator_hip_vs_oracle_other_shapes(B, N):
    dev = torch.device("cuda:0")
    _, params = _cf_fixture()
    g = torch.Generator().manual_seed(N)
    t = 0.3 * torch.randn(B, N, generator=g)
    q = t + 0.1 * torch.randn(B, N, generator=g)
    _cf_check(_cf_model(params, dev), params, t, q, dev, 2e-5, 5e-4)


def _pd_fixture(name):
    fx = load_file(os.path.join(G, "pdisc_small.safetensors"))
    params = {k[len(name) + 3:]: v for k, v in fx.items() if k.startswith(name + ".w.")}
    return {k[len(name) + 1:]: v for k, v in fx.items() if k.startswith(name + ".") and ".w." not in k[:len(name) + 3]}, params


@pytest.mark.parametrize("name", ["pitch", "dur"])
def test_pitch_discriminator_oracle_matches_reference(name):
    from oracle import discriminator as od
    fx, p = _pd_fixture(name)
    t, q = fx["target"], fx["pred"].clone().requires_grad_(True)
    rs, gs = od.pitch_discriminator(p, t), od.pitch_discriminator(p, q)
    for i in range(5):
        assert (rs[i] - fx[f"real_score{i}"]).abs().max().item() <= 2e-6 and (gs[i] - fx[f"gen_score{i}"]).abs().max().item() <= 2e-6
    gl = od.generator_loss_helper(rs, gs)
    gl.backward()
    assert abs(gl.item() - fx["gen_loss"].item()) <= 1e-5
    assert (q.grad - fx["d_pred"]).abs().max().item() <= 1e-5 * fx["d_pred"].abs().max().item()
    pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    dl = od.discriminator_loss_helper(od.pitch_discriminator(pp, t), od.pitch_discriminator(pp, fx["pred"]))
    dl.backward()
    assert abs(dl.item() - fx["disc_loss"].item()) <= 1e-5
    for k in pp:
        ref = fx["grad." + k]
        assert (pp[k].grad - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1e-3), k


@pytest.mark.gpu
@pytest.mark.parametrize("name,dim_in,kernel", [("pitch", 2, 21), ("dur", 1, 5)])
def test_pitch_discriminator_hip(name, dim_in, kernel):
    """HIP PitchDiscriminator (pitch_disc / dur_disc configurations): score maps and both loss values against the
    reference fixture, gradients through the pinned oracle (two-part scheme), then a longer odd-length sequence."""
    from oracle import discriminator as od
    from stylish_tts_amd.discriminators import PitchDiscriminator
    dev = torch.device("cuda:0")
    fx, params = _pd_fixture(name)
    m = PitchDiscriminator(dim_in=dim_in, kernel=kernel)
    m.load_state_dict(params, strict=True)
    m = m.to(dev)
    cases = [(fx["target"], fx["pred"], True)]
    g = torch.Generator().manual_seed(9)
    tl = torch.randn(4, dim_in, 517, generator=g) * 2
    cases.append((tl, tl + 0.5 * torch.randn(tl.shape, generator=g), False))
    for t, q0, pinned in cases:
        td, qd = t.to(dev), q0.to(dev)
        rs_h = [x.cpu().clone().requires_grad_(True) for x in m(td)[0]]
        gs_h = [x.cpu().clone().requires_grad_(True) for x in m(qd)[0]]
        if pinned:
            for i in range(5):
                assert (rs_h[i] - fx[f"real_score{i}"]).abs().max().item() <= 2e-5 * max(fx[f"real_score{i}"].abs().max().item(), 0.1)
                assert (gs_h[i] - fx[f"gen_score{i}"]).abs().max().item() <= 2e-5 * max(fx[f"gen_score{i}"].abs().max().item(), 0.1)
        gl = od.generator_loss_helper(rs_h, gs_h)
        g_gen = torch.autograd.grad(gl, gs_h)
        dl = od.discriminator_loss_helper(rs_h, gs_h)
        g_dr = torch.autograd.grad(dl, rs_h, retain_graph=True)
        g_dg = torch.autograd.grad(dl, gs_h)
        q = q0.clone().requires_grad_(True)
        keys = sorted(params)
        pp = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        so_t, so_q = od.pitch_discriminator(pp, t), od.pitch_discriminator(pp, q)
        dq, = torch.autograd.grad(so_q, q, grad_outputs=list(g_gen), retain_graph=True)
        dw = torch.autograd.grad(so_t + so_q, [pp[k] for k in keys], grad_outputs=list(g_dr) + list(g_dg))
        for p_ in m.parameters():
            p_.grad = None
        d_pred = torch.zeros_like(qd)
        gen, disc = m.losses(td, qd, gen_scale=2.0, d_pred=d_pred, disc_scale=3.0)
        assert abs(gen[0].item() - gl.item()) <= 2e-5 * gl.item() and abs(disc[0].item() - dl.item()) <= 2e-5 * dl.item()
        if pinned:
            assert abs(gen[0].item() - fx["gen_loss"].item()) <= 5e-5 * fx["gen_loss"].item()
            assert abs(disc[0].item() - fx["disc_loss"].item()) <= 5e-5 * fx["disc_loss"].item()
        err = (d_pred.cpu() - 2.0 * dq).abs().max().item()
        assert err <= 2e-4 * 2.0 * dq.abs().max().item(), ("d_pred", err)
        got = dict(m.named_parameters())
        for k, ref in zip(keys, dw):
            ref = 3.0 * ref
            err = (got[k].grad.cpu() - ref).abs().max().item()
            assert err <= (1e-3 if k.endswith("original0") else 2e-4) * max(ref.abs().max().item(), 1e-3), (k, err)


def test_discriminator_shells_load_reference_state_dicts_and_refuse_the_cpu():
    """The three shells take the reference modules' state_dicts key for key (strict), and there is no CPU path."""
    from stylish_tts_amd import lib as L
    from stylish_tts_amd.discriminators import ContextFreeDiscriminator, PitchDiscriminator, SpecDiscriminator
    _, p = _fixture()
    m = SpecDiscriminator()
    m.load_state_dict(p, strict=True)
    with pytest.raises(L.StyError):
        m(torch.zeros(1, 1, 9, 9))
    _, pc = _cf_fixture()
    c = ContextFreeDiscriminator()
    c.load_state_dict(pc, strict=True)
    with pytest.raises(L.StyError):
        c(torch.zeros(1, 2048))
    for name, dim_in, k in (("pitch", 2, 21), ("dur", 1, 5)):
        _, pp = _pd_fixture(name)
        d = PitchDiscriminator(dim_in=dim_in, kernel=k)
        d.load_state_dict(pp, strict=True)
        with pytest.raises(L.StyError):
            d(torch.zeros(1, dim_in, 16))
