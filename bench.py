#!/usr/bin/env python
"""Benchmark of the stylish-tts acoustic hot path on MI355X (see DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W [--workload c3|c3-fp32|c3-gan|c3-textual|c3-duration|c2|c2-gan|c2-fwd|c5|c5-bf16|tts] [--no-extra]

A step is one pass of the hot path over one synthetic batch already resident in HBM:
  c2 (BASELINE.json configs[1]): sample_dataset shape, B=16 utterances of T=160 mel frames (2.0 s),
      L=37 phoneme tokens, fp32 -- one train_acoustic step (train/stage_type.py:346-373 + stage.py:124):
      AcousticStep forward (mel x2 + energy, alignment, style encoder, text encoder, alignment expand, decoder,
      vocoder), mel + multi-phase losses through the 3-resolution STFT features, backward through the predictor and
      the style encoder, gradient all-reduce (N > 1) and the AdamW step of both models.  GAN / WavLM loss terms are
      off (third-party models); module.train() behaviour (BatchNorm batch statistics, spectral-norm
      power iteration, random Decoder smoothing, TextEncoder dropout).
  c2-fwd: the forward half only (AcousticStep forward + the six multi-spectrogram lists).
  c3 (DEFAULT; BASELINE.json configs[2], the configuration the metric is quoted on): LJSpeech shape, B=32, T=520,
      L=100, the same training step with bf16 operands on the dense convs / Linears
      (fp32 accumulation, storage, norms, attention and losses: SURVEY.md 8(d) "bf16 autocast for conv/GEMM").
  c3-fp32: the same shape entirely in fp32.
  c3-gan / c2-gan: c3 / c2 plus the adversarial terms (three spectrogram discriminators + the waveform discriminator) and
           the discriminator step (train/stage.py:124-146; the WavLM term stays off).
  c3-textual / c3-duration: the second- and third-stage training steps (train_textual / train_duration) at c3's shape.
  c5: vocoder only, B=8, T=800 (10 s utterances), the roofline workload of SURVEY.md 8(d).
  c5-bf16: the same with bf16 operands on the dense convs (outside the fp32 parity gates; reported beside c5).
  tts: the export graph (ExportModel.forward): B=8 token strings of L=100 -> duration predictor ->
      alignment -> pitch / energy predictor -> speech predictor -> audio, fp32 inference; frames = B x predicted frames.
N > 1: one process per GPU (torch.distributed, RCCL), utterances sharded across ranks (weak scaling, no data-path
collective in the forward); time = max over ranks between two barriers; value = frames of all ranks / time.
The default run (c3 at N = 1) also carries c5 / c5-bf16 (vocoder only) as `extra` summaries and the c3 step under RCCL at world
size 1; `--extra` adds c2, c3-fp32 and c3-gan.  Prints ONE JSON line (< 8 000 bytes: headline, roofline, cpu_baseline) on
rank 0; the per-family kernel tables and the extras' full records go to `bench_detail.json`.
"""
import argparse
import json
import os
import sys
import time

# The training step runs on four HIP streams and RCCL adds its own.  With the runtime's default of
# four hardware queues per process, a fifth concurrently active stream sends the step from 38 to 59-70 ms (measured
# with dummy streams); with two hardware queues it stays at 38-42 ms however many streams exist.  Must be set before
# the HIP runtime is loaded, i.e. before `import torch`.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "c2": dict(B=16, T=160, L=37, what="train"),
    "c2-fwd": dict(B=16, T=160, L=37, what="forward"),
    "c3": dict(B=32, T=520, L=100, what="train", compute="bf16"),
    "c3-fp32": dict(B=32, T=520, L=100, what="train"),
    "c3-gan": dict(B=32, T=520, L=100, what="train", compute="bf16", gan=True),
    "c2-gan": dict(B=16, T=160, L=37, what="train", gan=True),
    "c3-textual": dict(B=32, T=520, L=100, what="textual", compute="bf16"),
    "c3-duration": dict(B=32, T=520, L=100, what="duration", compute="bf16"),
    "c5": dict(B=8, T=800, L=0, what="vocoder"),
    "c5-bf16": dict(B=8, T=800, L=0, what="vocoder", compute="bf16"),
    "tts": dict(B=8, T=0, L=100, what="synth"),
}
PASS = {
    "train": "forward + backward + AdamW (mel + multi-phase losses; GAN/WavLM terms off; train mode)",
    "textual": "train_textual step (pitch / energy predictor + pitch style encoder trained through the frozen speech "
               "predictor: mel, pitch, energy, pitch_disc generator losses; pitch_disc step)",
    "duration": "train_duration step (duration predictor + duration style encoder: duration, duration_ce, dur_disc "
                "generator losses; dur_disc step)",
    "forward": "forward only (AcousticStep forward + multi-spectrogram features)",
    "vocoder": "vocoder forward only (inference)",
    "synth": "text -> audio inference (duration, pitch/energy and speech predictors)",
}
PEAK_FP32_TFLOPS = 157.3   # MI355X fp32 vector = fp32-input MFMA peak (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0


def make_inputs(w, seed, device):
    g = torch.Generator().manual_seed(seed)
    B, T, L = w["B"], w["T"], w["L"]
    pitch = torch.rand(B, T, generator=g) * 200 + 80
    unv = (torch.rand(B, (T + 9) // 10, generator=g) < 0.3).repeat_interleave(10, dim=1)[:, :T]
    pitch[unv] = 0
    d = dict(pitch=pitch, voiced=(pitch > 20).float(), energy=torch.randn(B, T, generator=g),
             style=torch.randn(B, 64, generator=g))
    if w["what"] == "synth":
        return {k: v.to(device) for k, v in dict(
            texts=torch.randint(1, 178, (B, L), generator=g), text_lengths=torch.full((B,), L, dtype=torch.int64),
            speech_style=torch.randn(B, 64, generator=g), pe_style=torch.randn(B, 64, generator=g),
            duration_style=torch.randn(B, 64, generator=g)).items()}
    if w["what"] == "vocoder":
        d["mel"] = torch.randn(B, 128, T, generator=g)
    else:
        # synthetic sample_dataset-like audio: harmonics of a slowly varying f0 + noise
        t = torch.arange(300 * T) / 24000.0
        f0 = 110.0 + 100.0 * torch.rand(B, 1, generator=g)
        d["audio_gt"] = sum(torch.sin(2 * torch.pi * f0 * (h + 1) * t) / (h + 1) for h in range(8)) * 0.15 \
            + 0.01 * torch.randn(B, 300 * T, generator=g)
        d["texts"] = torch.randint(1, 178, (B, L), generator=g)
        d["text_lengths"] = torch.full((B,), L, dtype=torch.int64)
        # integer durations >= 1 summing to T (multinomial split), as the alignment cache holds them
        dur = torch.ones(B, L)
        extra = torch.multinomial(torch.ones(L), T - L, replacement=True, generator=g) if T > L else torch.zeros(0)
        for b in range(B):
            idx = torch.multinomial(torch.ones(L), T - L, replacement=True, generator=g)
            dur[b] += torch.bincount(idx, minlength=L).float()
        d["durations"] = dur
    return {k: v.to(device) for k, v in d.items()}


def build_model(device):
    import stylish_tts_amd as S
    from stylish_tts_amd.manifest import speech_predictor_manifest, style_encoder_manifest
    from stylish_tts_amd.synthetic_weights import fill_state_dict   # deterministic random-init weights
    P = fill_state_dict(speech_predictor_manifest(), 0)
    m = S.SpeechPredictor()
    m.load_state_dict(P, strict=False)
    se = S.MelStyleEncoder()
    se.load_state_dict(fill_state_dict(style_encoder_manifest(), 0))
    return m.to(device), se.to(device), P


def norm_kernel_name(name):
    """A kernel's name reduced to what identifies the instantiation: no return type, no `sty::` / anonymous namespace, no
    argument list, no blanks -- the form in which the library's `sty_prof_row.inst`, the rows of a committed
    `profiles/*_kernel_stats.txt` and the keys of `profiles/*_pmc_traffic.json` are compared.  Mangled names
    (`_ZN3sty...`, as the tracer leaves a few of them) are demangled first when c++filt is there."""
    import re
    n = name.strip()
    if n.startswith("_Z"):
        try:
            import subprocess
            n = subprocess.run(["c++filt", n], stdout=subprocess.PIPE, timeout=10).stdout.decode().strip() or n
        except Exception:
            pass
    trunc = n.endswith("...")  # tools/rocpd_summary.py of rounds 1-5 cut names at 107 characters
    if trunc:
        n = n[:-3]
    n = re.sub(r"^void\s+", "", n)
    n = n.replace("(anonymous namespace)::", "").replace("sty::", "")
    depth, cut = 0, len(n)
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    n = re.sub(r"\s+", "", n[:cut])
    return n, trunc and cut == len(n)


def _same_kernel(inst, other):
    """`inst`: an instantiation name from the library; `other`: a name from a profile file (possibly cut short)."""
    a, _ = norm_kernel_name(inst)
    b, cut = norm_kernel_name(other)
    return a == b or (cut and len(b) > 20 and a.startswith(b))


def _profile_files(workload, suffix):
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{workload}_{suffix}")), reverse=True)


def rocprof_rows(workload):
    """(rows, source) of the newest committed rocprofv3 --kernel-trace --stats summary of this same command
    (profiles/*_<workload>_kernel_stats.txt, tools/profile_round.sh): rows = [(calls, total_us, name)] in the file's order,
    i.e. largest total first."""
    for f in _profile_files(workload, "kernel_stats.txt"):
        rows = []
        try:
            for ln in open(f).read().splitlines():
                parts = ln.split(None, 4)
                if len(parts) == 5 and parts[0].isdigit():
                    rows.append((int(parts[0]), float(parts[1]), parts[4]))
        except Exception:
            continue
        if rows:
            return rows, os.path.relpath(f, ROOT)
    return [], None


def rocprof_avg_us(insts, workload):
    """Average launch duration (us) of the instantiation(s) `insts` in the committed rocprofv3 summary of this same command --
    the figure `roofline.avg_launch_us` (HIP events, live) has to agree with."""
    rows, src = rocprof_rows(workload)
    tot = n = 0.0
    for calls, total, name in rows:
        if any(_same_kernel(i, name) for i in insts):
            tot += total
            n += calls
    return (tot / n, src) if n else (None, src)


def pmc_traffic(insts, workload):
    """HBM bytes per launch of a kernel from the committed PMC passes of this same command (tools/profile_round.sh ->
    profiles/*_pmc_traffic.json; FETCH_SIZE and WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md prescribes).
    Counters cannot be read from inside the process, so this is the recorded figure of the SAME workload, launch-weighted over
    `insts`; None when there is no such file."""
    for f in _profile_files(workload, "pmc_traffic.json"):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        tot = n = 0.0
        for k, v in d.items():
            if k.startswith("_") or not isinstance(v, dict) or "launches" not in v:
                continue
            if any(_same_kernel(i, k) for i in insts):
                tot += (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"]
                n += v["launches"]
        if n:
            return tot / n, os.path.relpath(f, ROOT)
    return None, None


def pick_dominant(rows, workload):
    """Which kernel the `roofline` object is about -- decided the same way on every box: the top row of the committed
    rocprofv3 summary of this command (profiles/*_<workload>_kernel_stats.txt) if the library still has that instantiation
    (`rows` = this run's warm-up table, one row per (family, instantiation)); otherwise -- a library newer than its profiles --
    the instantiation with the largest summed time in `rows`.  Returns (family, inst, how)."""
    rows = [r for r in rows if r["launches"]]
    if not rows:
        return None, None, None
    prows, src = rocprof_rows(workload)
    for calls, total, name in prows[:1]:
        for r in rows:
            if r["inst"] and _same_kernel(r["inst"], name):
                return r["name"], r["inst"], f"top row of {src}"
    r = max(rows, key=lambda r: r["ms"])
    return r["name"], r["inst"], ("largest summed event time of the warm-up step (no committed profile names a kernel of this "
                                  "library as its top row)")


def _usable_cpus():
    """cores this process may actually use: affinity mask intersected with the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def _cpu_baseline_worker(w, q, budget_s):
    """Runs in a child process: the CPU oracle (plain PyTorch restatement, kind 'port'), bounded sample."""
    from oracle import frontend, speech_predictor as osp, vocoder as ov
    from oracle.manifest import speech_predictor_manifest
    from oracle.weights import fill_state_dict
    from oracle.manifest import style_encoder_manifest
    P = fill_state_dict(speech_predictor_manifest(), 0)
    Pse = fill_state_dict(style_encoder_manifest(), 0)
    Bs, T = 2, w["T"]
    inp = make_inputs(dict(w, B=Bs), 99, "cpu")
    if w["what"] == "synth":
        from oracle import predictors as OP
        from oracle.manifest import duration_predictor_manifest, pitch_energy_predictor_manifest
        Pd = fill_state_dict(duration_predictor_manifest(), 3)
        Pp = fill_state_dict(pitch_energy_predictor_manifest(), 4)
        with torch.no_grad():
            pred = OP.duration_predictor(Pd, inp["texts"], inp["text_lengths"], inp["duration_style"])
            T = OP.duration_to_alignment(OP.prediction_to_duration(pred, inp["text_lengths"])).shape[2]
    noise = torch.randn(Bs, 300 * T, 9)

    if w["what"] == "train":
        from oracle import losses as ol
        for d in (P, Pse):
            for v in d.values():
                if v.is_floating_point():
                    v.requires_grad_(True)
        params = [v for d in (P, Pse) for v in d.values() if v.requires_grad]
        opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4, betas=(0.85, 0.99), eps=1e-9)

    def once():
        t0 = time.perf_counter()
        if w["what"] == "synth":
            with torch.no_grad():
                pred = OP.duration_predictor(Pd, inp["texts"], inp["text_lengths"], inp["duration_style"])
                ali = OP.duration_to_alignment(OP.prediction_to_duration(pred, inp["text_lengths"]))
                f0, en = OP.pitch_energy_predictor(Pp, inp["texts"], inp["text_lengths"], ali, inp["pe_style"])
                osp.speech_predictor(P, inp["texts"], inp["text_lengths"], ali, f0, en, (f0 > 20).float(),
                                     inp["speech_style"], f0, noise)
        elif w["what"] == "vocoder":
            with torch.no_grad():
                ov.multi_generator(P, "generator", inp["mel"], inp["style"], inp["pitch"], inp["voiced"], noise)
        elif w["what"] == "train":
            opt.zero_grad()
            a = osp.acoustic_forward(P, Pse, inp["audio_gt"], inp["texts"], inp["text_lengths"], inp["pitch"],
                                     inp["durations"], noise)
            ol.acoustic_losses(inp["audio_gt"], a.squeeze(1))[2].backward()
            opt.step()
        else:
            with torch.no_grad():
                a = osp.acoustic_forward(P, Pse, inp["audio_gt"], inp["texts"], inp["text_lengths"], inp["pitch"],
                                         inp["durations"], noise)
                for fft, hop, win in frontend.RESOLUTIONS:
                    frontend.multi_spectrogram_single(a.squeeze(1), fft, hop, win)
                    frontend.multi_spectrogram_single(inp["audio_gt"], fft, hop, win)
        return time.perf_counter() - t0

    usable = _usable_cpus()
    t_all = time.perf_counter()
    best, best_thr, n_timed = None, None, 0
    if True:
        # ATen's CPU kernels on this op mix (tiny-channel convs, elementwise) get slower when the OpenMP team is
        # larger than the cores actually granted; try a few team sizes and report the one used.
        for thr in sorted({min(usable, t) for t in ((16, 32) if w["what"] == "train" else (8, 16, 32, 64))}):
            torch.set_num_threads(thr)
            once()
            for _ in range(2):
                dt = once()
                n_timed += 1
                if best is None or dt < best:
                    best, best_thr = dt, thr
                q.put(dict(value=Bs * T / best, unit="frames/s", cores=best_thr, kind="port",
                           host_cores=os.cpu_count(), usable_cores=usable,
                           sample=f"oracle {PASS[w['what']].split(' (')[0]}, B={Bs}, T={T}; best of {n_timed} timed iterations "
                                  f"(1 warm-up per OpenMP team size), {time.perf_counter() - t_all:.1f} s of CPU work"))
            if time.perf_counter() - t_all > budget_s:
                break


def cpu_baseline(w, budget_s=15.0, hard_timeout_s=120.0):
    """CPU baseline in a child process with a hard wall-clock limit, so the bench can never hang on it."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_cpu_baseline_worker, args=(w, q, budget_s), daemon=True)
    p.start()
    p.join(hard_timeout_s)
    if p.is_alive():
        p.kill()
        p.join()
    last = None
    while not q.empty():
        last = q.get()
    if last is None:
        return dict(value=None, unit="frames/s", cores=_usable_cpus(), kind="port",
                    sample=f"oracle forward did not finish one timed iteration within {hard_timeout_s:.0f} s")
    return last


def run_workload(name, steps, warmup, rank, world, device, lib, L, D, share, serial_pass=True):
    """W untimed + 1 profiled step, K timed steps between two barriers, max over ranks; returns the record (rank 0)
    or None.  Every model / trainer / workspace of the workload is released before returning."""
    w = WORKLOADS[name]
    model, style_enc, P = build_model(device)
    from stylish_tts_amd.acoustic import AcousticTrainer, acoustic_forward
    from stylish_tts_amd.frontend import MultiSpectrogram
    mspec = MultiSpectrogram(sample_rate=24000)
    inp = make_inputs(w, 1000 + rank, device)
    B, T = w["B"], w["T"]
    bf16 = w.get("compute") == "bf16"
    mrd, wave_disc = None, None
    if w.get("gan"):  # the three spectrogram discriminators (random init: the reference's Conv2d / weight_norm defaults)
        from stylish_tts_amd.discriminators import SpecDiscriminator
        torch.manual_seed(7)
        from stylish_tts_amd.discriminators import ContextFreeDiscriminator
        mrd = [SpecDiscriminator().to(device) for _ in range(3)]
        wave_disc = ContextFreeDiscriminator().to(device)
    trainer = (AcousticTrainer(model, style_enc, lr=1e-4, compute=w.get("compute", "fp32"), seed=rank, mrd=mrd,
                               disc=wave_disc if mrd is not None else None)
               if w["what"] == "train" else None)
    if bf16 and trainer is None:
        model.set_train_opts(compute_bf16=True)

    stage_trainer = None
    if w["what"] in ("textual", "duration"):
        import stylish_tts_amd as S
        from stylish_tts_amd.discriminators import PitchDiscriminator
        from stylish_tts_amd.manifest import (duration_predictor_manifest, pitch_energy_predictor_manifest,
                                              pitch_style_encoder_manifest, style_encoder_manifest)
        from stylish_tts_amd.synthetic_weights import fill_state_dict
        torch.manual_seed(11)
        if w["what"] == "textual":
            from stylish_tts_amd.textual import TextualTrainer
            pem, pse = S.PitchEnergyPredictor(), S.PitchStyleEncoder()
            pem.load_state_dict(fill_state_dict(pitch_energy_predictor_manifest(), 4))
            pse.load_state_dict(fill_state_dict(pitch_style_encoder_manifest(), 5))
            stage_trainer = TextualTrainer(pem.to(device), pse.to(device), model, style_enc,
                                           PitchDiscriminator(dim_in=2, kernel=21).to(device), lr=1e-4, seed=rank,
                                           compute=w.get("compute", "fp32"))
        else:
            from stylish_tts_amd.duration import DurationTrainer
            dpm, dse = S.DurationPredictor(), S.MelStyleEncoder()
            dpm.load_state_dict(fill_state_dict(duration_predictor_manifest(), 3))
            dse.load_state_dict(fill_state_dict(style_encoder_manifest(), 7))
            stage_trainer = DurationTrainer(dpm.to(device), dse.to(device), PitchDiscriminator(dim_in=1, kernel=5).to(device),
                                            torch.ones(16), lr=1e-4, seed=rank, compute=w.get("compute", "fp32"))
    synth = None
    if w["what"] == "synth":
        import stylish_tts_amd as S
        from stylish_tts_amd.manifest import duration_predictor_manifest, pitch_energy_predictor_manifest
        from stylish_tts_amd.synthetic_weights import fill_state_dict
        dpm, pem = S.DurationPredictor(), S.PitchEnergyPredictor()
        dpm.load_state_dict(fill_state_dict(duration_predictor_manifest(), 3))
        pem.load_state_dict(fill_state_dict(pitch_energy_predictor_manifest(), 4))
        synth = S.ExportModel(speech_predictor=model, pitch_energy_predictor=pem.to(device),
                              duration_predictor=dpm.to(device))

    def step(i):
        if synth is not None:
            return synth(inp["texts"], inp["text_lengths"], inp["speech_style"], inp["pe_style"],
                         inp["duration_style"], seed=i)
        if w["what"] == "textual":
            log = stage_trainer.train_batch(audio_gt=inp["audio_gt"], texts=inp["texts"], text_lengths=inp["text_lengths"],
                                            pitch=inp["pitch"], durations=inp["durations"], seed=i)
            return torch.stack([log["mel"], log["pitch"], log["energy"], log["generator"]])
        if w["what"] == "duration":
            log = stage_trainer.train_batch(audio_gt=inp["audio_gt"], texts=inp["texts"], text_lengths=inp["text_lengths"],
                                            durations=inp["durations"])
            return torch.stack([log["duration"], log["duration_ce"], log["generator"]])
        if w["what"] == "train":
            # one train_acoustic step incl. gradient all-reduce and optimizer; returns the loss values
            return trainer.train_batch(audio_gt=inp["audio_gt"], texts=inp["texts"], text_lengths=inp["text_lengths"],
                                       pitch=inp["pitch"], durations=inp["durations"], seed=i)
        with torch.no_grad():
            if w["what"] == "vocoder":
                return model.vocoder_forward(mel=inp["mel"], style=inp["style"], pitch=inp["pitch"],
                                             voiced=inp["voiced"], seed=i).audio
            o = acoustic_forward(model, style_enc, audio_gt=inp["audio_gt"], texts=inp["texts"],
                                 text_lengths=inp["text_lengths"], pitch=inp["pitch"], durations=inp["durations"],
                                 T=T, seed=i, multi_spectrogram=mspec)
            return o.pred.audio

    # tuning aid: STY_BENCH_DUMMY_STREAMS=k keeps k more streams busy with a tiny kernel per step (stands in for the
    # queues a communication library adds), to see how the step reacts to more active hardware queues
    dummies = [torch.cuda.Stream(device=device) for _ in range(int(os.environ.get("STY_BENCH_DUMMY_STREAMS", "0")))]
    dbuf = torch.zeros(1024, device=device)
    if dummies:
        inner = step

        def step(i):  # noqa: F811
            for s_ in dummies:
                with torch.cuda.stream(s_):
                    dbuf.add_(1.0)
            return inner(i)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Warm-up: W untimed steps, then one more untimed step timed per kernel family with HIP events on the launch
    # streams; the per-family table `kernels` comes from there.  Timed region: only the dominant family keeps its
    # events (two events per launch on every instrumented launch cost ~7 % of a c2 step), and `roofline` is computed
    # from those -- live, inside the timed region, as the contract asks.
    for i in range(warmup):
        out = step(i)
    nprof = 1
    torch.cuda.synchronize()
    lib.sty_prof_enable(1)
    out = step(warmup)
    barrier()
    lib.sty_prof_enable(0)
    warm_raw = L.prof_report(4096, by_inst=True)   # one row per (family, instantiation)
    warm_prof = L.group_families(warm_raw)
    dom_name, dom_inst, dom_how = pick_dominant(warm_raw, name)
    if dom_name:
        lib.sty_prof_only(dom_name.split(" ")[0].encode())  # STY_PROF_SHAPES appends the shape to the family name
        lib.sty_prof_enable(1)
    if trainer is not None:  # how long each step's AdamW stands still for the gradient exchange (events around GradBuckets.finish)
        for o in trainer.opt.values():
            o.grads.exposed = []
            o.grads.measure_exposed = True
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(warmup + 1 + i)
    # the host's time to issue the steps (no sync yet).  It contains the runtime's back-pressure -- the host is kept a
    # bounded number of commands ahead of the GPU (c3: 63 of 75 ms, c2: 28 of 34 ms for the same ~1 700 launches of
    # 3 us each) -- so it is a lower bound on how far ahead the host runs, not the cost of the launches
    host_dt = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    lib.sty_prof_enable(0)
    lib.sty_prof_only(None)
    prof = L.prof_report(4096, by_inst=True) if rank == 0 else []
    assert bool(torch.isfinite(out).all())
    # After the timed region (training workloads, rank 0's view): the same step with the library's side streams off, all
    # families timed -- the dominant kernel's duration free of the stretch from sharing the chip with the
    # weight-gradient / style-encoder streams.  Reported beside `roofline`, never instead of it.
    serial_prof, serial_raw = [], []
    if trainer is not None and serial_pass:  # every rank: the steps contain the gradient all-reduce
        lib.sty_set_single_stream(1)
        trainer.single_stream = True
        step(warmup + steps + 1)
        torch.cuda.synchronize()
        lib.sty_prof_enable(1)
        for i in range(2):
            step(warmup + steps + 2 + i)
        torch.cuda.synchronize()
        lib.sty_prof_enable(0)
        serial_raw = L.prof_report(4096, by_inst=True)
        serial_prof = L.group_families(serial_raw)
        ts = time.perf_counter()  # and the single-stream step itself, without the per-launch events
        for i in range(2):
            step(warmup + steps + 4 + i)
        torch.cuda.synchronize()
        serial_ms = 1e3 * (time.perf_counter() - ts) / 2
        lib.sty_set_single_stream(0)
        trainer.single_stream = False
    barrier()
    exposed_ms = 0.0
    if trainer is not None:
        for o in trainer.opt.values():
            o.grads.measure_exposed = False
            exposed_ms += o.grads.exposed_ms()
    rank_view = None
    if world > 1 or D.collectives_on():
        # every rank's own step time and exposed exchange time (the driver's first 8-GPU run should say WHICH rank is the
        # straggler and whether the wire or the host is what it waits for)
        mine = torch.tensor([1e3 * dt / steps, exposed_ms / steps, 1e3 * host_dt / steps], dtype=torch.float64, device=device)
        if world > 1:
            allr = [torch.zeros_like(mine) for _ in range(world)]
            torch.distributed.all_gather(allr, mine)
        else:
            allr = [mine]
        rank_view = {"per_rank_ms": [float(t[0]) for t in allr], "allreduce_exposed_ms": [float(t[1]) for t in allr],
                     "host_issue_ms": [float(t[2]) for t in allr],
                     "exchange": ("library communicator (sty_comm_*: ncclReduceScatter + ncclAllGather on the library's stream)"
                                  if D.native_comm() is not None else "torch.distributed all_reduce"),
                     "env": {k: v for k, v in os.environ.items()
                             if k in ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY", "OMP_NUM_THREADS", "STY_NATIVE_COMM", "STY_COMM_PRIORITY")
                             or k.startswith(("NCCL_", "RCCL_"))}}
    dt = D.max_over_ranks(dt, device)
    if synth is not None:
        T = out.shape[-1] // 300  # frames the duration predictor asked for (same inputs every step)
    phases = None
    if trainer is not None and getattr(trainer, "_probe_on", False):
        # STY_STEP_PROBE=1 (a tuning aid, not part of the default line): device time stamps of the phases of one more step
        if getattr(trainer, "single_stream", False):
            lib.sty_set_single_stream(0)
            trainer.single_stream = False
        step(warmup + steps + 50)
        step(warmup + steps + 51)
        step(warmup + steps + 52)
        torch.cuda.synchronize()
        phases = [[n, round(t, 3)] for n, t in trainer.probe_report()]
    del trainer, model, style_enc, synth, inp, out, step
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    frames = world * B * T * steps
    rec = {
        "metric": "audio frames/sec/GPU (24 kHz) forward+backward; DDP scaling 1/2/4/8 MI355X",
        "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 GEMM operands, f32 accumulation/storage" if bf16 else "f32", "data": "synthetic",
        "config": {"workload": f"{name}: B={B}/GPU T={T} frames ({T / 80:.1f} s) L={w['L']}",
                   "pass": (PASS[w["what"]] if not w.get("gan") else
                            "forward + backward + AdamW (mel + multi-phase + generator loss of the three spectrogram discriminators and "
                            "the waveform discriminator; discriminator losses + AdamW step of mrd{i} and disc; WavLM off)"),
                   "x_realtime": frames / dt / 80.0},
        "host_issue_ms_per_step": 1e3 * host_dt / steps,
    }
    if rank_view:
        rec["rank_view"] = rank_view
    if phases:
        rec["phases_ms"] = phases
    if w["what"] == "vocoder":
        # achieved HBM rate of the vocoder stack from the committed PMC passes of this same workload (FETCH_SIZE x2 + WRITE_SIZE
        # summed over every kernel of a forward, tools/profile_round.sh) over THIS run's time per forward
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{name}_pmc_traffic.json")), reverse=True):
            try:
                tot = json.load(open(f)).get("_total")
            except Exception:
                tot = None
            if tot:
                rec["hbm_counter_GBps"] = (tot["fetch_bytes_per_pass"] + tot["write_bytes_per_pass"]) / (dt / steps) / 1e9
                rec["hbm_counter_bytes_per_pass"] = tot["fetch_bytes_per_pass"] + tot["write_bytes_per_pass"]
                rec["hbm_counter_source"] = os.path.relpath(f, ROOT)
                break
    if prof:
        # the timed region timed ONE family (sty_prof_only); the roofline is about ONE instantiation of it -- the one
        # pick_dominant named, the same on every box
        cand = [r for r in prof if r["inst"] == dom_inst] or [max(prof, key=lambda r: r["ms"])]
        dom = cand[0]
        insts = [dom["inst"]] if dom["inst"] else []
        traffic, traffic_src = pmc_traffic(insts, name)
        rp_us, rp_src = rocprof_avg_us(insts, name)
        per = dom["ms"] / dom["launches"] * 1e-3
        tf = dom["flops"] / dom["launches"] / per / 1e12
        gbs = dom["bytes"] / dom["launches"] / per / 1e9
        # a kernel of the bf16 compute mode (",true>" family) is priced against the bf16 MFMA peak; the roof
        # quoted as `bound` is the one the kernel sits closer to (both fractions are in the record)
        peak = PEAK_BF16_TFLOPS if dom["name"].split(" ")[0].endswith("true>") or ",true," in dom["name"] else PEAK_FP32_TFLOPS
        f_mfma, f_hbm = tf / peak, gbs / PEAK_HBM_GBS
        hbm_bound = f_hbm > f_mfma
        rec["roofline"] = {"kernel": dom["inst"] or dom["name"], "family": dom["name"], "chosen_by": dom_how,
                           "bound": "hbm" if hbm_bound else "mfma",
                           "achieved": gbs if hbm_bound else tf, "peak": PEAK_HBM_GBS if hbm_bound else peak,
                           "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": f_hbm if hbm_bound else f_mfma,
                           "traffic": traffic, "traffic_source": traffic_src,
                           "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
                           "avg_launch_us": per * 1e6, "launches": dom["launches"],
                           "avg_launch_us_is": "HIP events on the launch stream inside the timed region: the kernel shares the "
                                               "chip with up to three other streams, so this includes queueing behind them; "
                                               "rocprof_avg_launch_us is the committed rocprofv3 --kernel-trace --stats average "
                                               "of the same command, single_stream.avg_launch_us the kernel alone on the chip",
                           "rocprof_avg_launch_us": rp_us, "rocprof_source": rp_src,
                           "mfma_TFLOPs": tf, "mfma_peak": peak, "mfma_frac": f_mfma,
                           "hbm_GBps_algorithmic": gbs, "hbm_frac": f_hbm,
                           "share_of_step_time": dom["ms"] / (1e3 * dt)}
        sp = [r for r in serial_raw if r["name"] == dom["name"] and r["inst"] == dom["inst"]]
        if sp:
            per1 = sp[0]["ms"] / sp[0]["launches"] * 1e-3
            tf1 = sp[0]["flops"] / sp[0]["launches"] / per1 / 1e12
            gb1 = sp[0]["bytes"] / sp[0]["launches"] / per1 / 1e9
            rec["roofline"]["single_stream"] = {
                "what": "same kernel, two extra steps after the timed region with the side streams off",
                "avg_launch_us": per1 * 1e6, "mfma_TFLOPs": tf1, "mfma_frac": tf1 / peak,
                "hbm_GBps_algorithmic": gb1, "hbm_frac": gb1 / PEAK_HBM_GBS}
        if serial_prof:  # every family alone on the chip: what each costs, free of the stretch from sharing it
            rec["single_stream_step_ms"] = serial_ms
            rec["single_stream_kernels"] = [
                {"name": r["name"], "launches": r["launches"] // 2, "ms_per_step": r["ms"] / 2,
                 "TFLOPs": r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] else 0.0,
                 "GBps": r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] else 0.0}
                for r in sorted(serial_prof, key=lambda r: -r["ms"])[:40]]
        # the same table grouped by kernel NAME (all instantiations of a template together): the legacy tiled conv kernel
        # is the largest family of the step by name and no single instantiation of it shows that
        by_name = {}
        for r in warm_prof:
            b = by_name.setdefault(r["name"].split("<")[0], dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            b["launches"] += r["launches"]
            b["ms"] += r["ms"]
            b["flops"] += r["flops"]
            b["bytes"] += r["bytes"]
        rec["kernels_by_name"] = [
            {"name": k, "launches": v["launches"], "ms_per_step": v["ms"] / nprof,
             "TFLOPs": v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0.0,
             "GBps": v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] else 0.0}
            for k, v in sorted(by_name.items(), key=lambda kv: -kv[1]["ms"])]
        # step level: algorithmic flops and bytes of every instrumented family of ONE step (the GEMM-like families and the
        # large element-wise ones carry a model; the small element-wise kernels do not and are not counted) against the
        # step time -- the honest whole-step position under both roofs
        sflops = sum(r["flops"] for r in warm_prof) / nprof
        sbytes = sum(r["bytes"] for r in warm_prof) / nprof
        step_s = dt / steps
        peak_step = PEAK_BF16_TFLOPS if bf16 else PEAK_FP32_TFLOPS
        rec["roofline_step"] = {
            "what": "sum over the instrumented kernel families of one step / ms_per_step",
            "algorithmic_TFLOP_per_step": sflops / 1e12, "algorithmic_GB_per_step": sbytes / 1e9,
            "TFLOPs": sflops / step_s / 1e12, "mfma_peak": peak_step, "mfma_frac": sflops / step_s / 1e12 / peak_step,
            "GBps": sbytes / step_s / 1e9, "hbm_peak": PEAK_HBM_GBS, "hbm_frac": sbytes / step_s / 1e9 / PEAK_HBM_GBS,
            "launches_instrumented": sum(r["launches"] for r in warm_prof) // nprof}
        rec["kernels_source"] = "HIP events over one untimed step after the warm-up; roofline: over the timed region"
        rec["kernels"] = [{"name": r["name"], "insts": r.get("insts"), "launches": r["launches"], "ms_per_step": r["ms"] / nprof,
                           "TFLOPs": r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] else 0.0,
                           "GBps": r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] else 0.0}
                          for r in sorted(warm_prof, key=lambda r: -r["ms"])]
    return rec


# carried as sub-records of the default (c3) line at N = 1: the other single-GPU configurations of BASELINE.json and the
# c3 step with the reference's adversarial terms on (what `train_acoustic` + the discriminator step cost in full)
EXTRA_DEFAULT = ("c5-bf16", "c5")               # the vocoder-only configuration of the north-star (HBM GB/s on the conv stack)
EXTRA_WORKLOADS = ("c5-bf16", "c5", "c2", "c3-fp32", "c3-gan")   # with --extra


LINE_LIMIT = 8000   # the driver's record keeps the last 8 000 characters of stdout: the ONE line must fit in it whole


def _r(x, sig=5):
    """floats to `sig` significant digits (the line is a record, not a lab notebook)"""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


_ROOF_KEYS = ("kernel", "family", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "rocprof_source",
              "algorithmic_bytes_per_launch", "avg_launch_us", "rocprof_avg_launch_us", "launches", "mfma_frac", "hbm_frac",
              "share_of_step_time")


def _short_roofline(r):
    if not r:
        return None
    out = {k: r[k] for k in _ROOF_KEYS if k in r}
    ss = r.get("single_stream")
    if ss:  # the same kernel alone on the chip (side streams off), after the timed region
        out["single_stream"] = {k: ss[k] for k in ("avg_launch_us", "mfma_frac", "hbm_frac") if k in ss}
    return out


def format_line(rec, detail_path=None):
    """The ONE stdout line: headline, roofline, step-level roofline, cpu_baseline and a four-field summary of every
    extra workload -- always under LINE_LIMIT bytes.  Everything else of `rec` (per-family kernel tables, the extras' full
    records) is the sidecar `bench_detail.json`.  tests/test_boundary.py formats a recorded `rec` through this function."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
           "vs_baseline", "dtype", "data", "config", "host_issue_ms_per_step", "launches_per_step",
           "single_stream_step_ms")
    line = {k: rec[k] for k in top if k in rec}
    if "roofline" in rec:
        line["roofline"] = _short_roofline(rec["roofline"])
    if "roofline_step" in rec:
        line["roofline_step"] = {k: rec["roofline_step"][k] for k in
                                 ("algorithmic_TFLOP_per_step", "algorithmic_GB_per_step", "TFLOPs", "GBps", "mfma_frac",
                                  "hbm_frac") if k in rec["roofline_step"]}
    if "cpu_baseline" in rec:
        line["cpu_baseline"] = rec["cpu_baseline"]
    if "library" in rec:
        line["library"] = {k: rec["library"][k] for k in ("sha1", "rebuilt_here") if k in rec["library"]}
    if "ranks" in rec:
        rk = dict(rec["ranks"])
        r1 = rk.get("rccl_world1")
        if isinstance(r1, dict) and "error" not in r1:
            rk["rccl_world1"] = {k: r1[k] for k in ("ms_per_step", "vs_no_process_group", "GPU_MAX_HW_QUEUES", "allreduce_exposed_ms",
                                                    "exchange") if k in r1}
        if isinstance(rk.get("exchange"), str):
            rk["exchange"] = rk["exchange"].split(" (")[0]
        line["ranks"] = rk
    if rec.get("extra"):
        line["extra"] = {}
        for name, r in rec["extra"].items():
            rf = r.get("roofline") or {}
            e = {"ms_per_step": r.get("ms_per_step"), "value": r.get("value"),
                 "roofline": {k: rf[k] for k in ("kernel", "bound", "frac", "achieved", "unit", "traffic", "avg_launch_us",
                                                   "rocprof_avg_launch_us") if k in rf}}
            for k in ("hbm_counter_GBps", "hbm_counter_source"):
                if k in r:
                    e[k] = r[k]
            line["extra"][name] = e
    if "phases_ms" in rec:
        line["phases_ms"] = rec["phases_ms"]
    if detail_path:
        line["detail"] = detail_path
    line = _r(line)
    s = json.dumps(line, separators=(",", ":"))
    for drop in ("phases_ms", "extra", "roofline_step", "library"):  # cannot happen with the fields above; a guard, not a plan
        if len(s) < LINE_LIMIT:
            break
        line.pop(drop, None)
        s = json.dumps(line, separators=(",", ":"))
    assert len(s) < LINE_LIMIT, len(s)
    return s


def _launch_ranks(n):
    """Re-run this command as n ranks under torch.distributed.run (127.0.0.1 rendezvous on a free port) and pass its
    output through; returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    # default = the configuration BASELINE.json's metric is quoted on: configs[2], LJSpeech shape, B=32, bf16
    ap.add_argument("--workload", default="c3", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the c5 sub-records and the RCCL world-1 record of the default run")
    ap.add_argument("--extra", action="store_true", help="also run c2 / c3-fp32 / c3-gan as sub-records (the default run carries c5 only)")
    ap.add_argument("--rccl1", action="store_true", help="tuning aid: the main record IS the step under RCCL at world size 1 with every "
                    "gradient bucket all-reduced (what `ranks.rccl_world1` measures), e.g. under rocprofv3 (tools/prof_rccl1.sh)")
    ap.add_argument("--detail", default=None, help="where the full record goes (default: bench_detail.json beside bench.py, "
                                                   "and gpurun_out/ when that directory exists)")
    args = ap.parse_args()
    # ONE JSON line on stdout: anything a library prints there (RCCL's version banner when the process group comes up)
    # goes to stderr instead -- file descriptor 1 is pointed at stderr for the whole run, the record is written to the
    # saved descriptor at the end
    sys.stdout.flush()
    out_fd = os.dup(1)
    os.dup2(2, 1)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` called directly: become the launcher.  One process per GPU, as the reference gets
        # its ranks from accelerate (train/train_context.py:94-104, train/train.py:188,208-211); the ranks run this
        # same file with the torchrun environment set and rank 0 prints the one JSON line.
        os.dup2(out_fd, 1)  # (the ranks inherit the real stdout; rank 0 does the same redirection for itself)
        sys.exit(_launch_ranks(args.gpus))

    from stylish_tts_amd import dist as D
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device: there is no CPU product path"
    # STY_BENCH_SHARE_DEVICE=1 (test aid): all ranks share device 0 and exchange gradients over gloo, so the N > 1 code
    # path can be exercised on a 1-GPU box; the numbers of such a run mean nothing
    share = os.environ.get("STY_BENCH_SHARE_DEVICE") == "1"
    if share:
        # two processes time-slicing ONE GPU: cross-queue event waits then stall for whole time slices (measured:
        # 161 ms per step single-stream, 16-28 s with the side streams).  Not a configuration anyone trains in; the
        # test aid runs single-stream.  One process per GPU keeps its streams.
        os.environ.setdefault("STY_NO_SIDE_STREAM", "1")
        os.environ.setdefault("STY_NO_SE_STREAM", "1")
    local = 0 if share else local
    torch.cuda.set_device(local)
    if not share and torch.cuda.device_count() < int(os.environ.get("WORLD_SIZE", "1")):
        raise SystemExit(f"bench.py: {os.environ['WORLD_SIZE']} ranks asked for, {torch.cuda.device_count()} HIP devices "
                         "visible (one process per GPU; STY_BENCH_SHARE_DEVICE=1 is the 1-GPU test aid)")
    if args.rccl1 and "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        os.environ.update(STY_DIST_FORCE_COLLECTIVE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world = D.init("gloo" if share else "nccl")  # "nccl" = RCCL; one process per GPU (torchrun environment)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} ranks")
    device = torch.device("cuda", local)

    import __graft_entry__ as ge
    from stylish_tts_amd import build as sbuild
    before = os.path.getmtime(sbuild.LIB) if os.path.exists(sbuild.LIB) else None
    ge.build()
    from stylish_tts_amd import lib as L
    lib = L.load()
    # which native library this run measured: the in-tree .so as shipped, or rebuilt here because a source was newer
    import hashlib
    with open(sbuild.LIB, "rb") as f:
        lib_info = {"file": os.path.relpath(sbuild.LIB, ROOT), "sha1": hashlib.sha1(f.read()).hexdigest()[:12],
                    "rebuilt_here": before is None or os.path.getmtime(sbuild.LIB) != before,
                    "hipcc": sbuild.hipcc_version()}
    rec = run_workload(args.workload, args.steps, args.warmup, rank, world, device, lib, L, D, share)
    extras = {}
    if world == 1 and args.workload == "c3" and not args.no_extra:
        # the other single-GPU configurations of BASELINE.json, same process, fewer steps; each is a full record of its
        # own (value, ms_per_step, roofline of ITS dominant kernel) without the per-family table
        for name in (EXTRA_WORKLOADS if args.extra else EXTRA_DEFAULT):
            heavy = bool(WORKLOADS[name].get("gan"))
            r = run_workload(name, min(args.steps, 5 if heavy else 10), min(args.warmup, 2 if heavy else 3), rank, world,
                             device, lib, L, D, share, serial_pass=False)
            if r is not None:
                r.pop("kernels", None)
                r.pop("kernels_source", None)
                extras[name] = r
    backend = torch.distributed.get_backend() if world > 1 else None
    if world > 1:
        torch.distributed.barrier()
        D.destroy_native_comm()
        torch.distributed.destroy_process_group()
    rccl1 = None
    if world == 1 and args.workload == "c3" and not args.no_extra and not share and not args.rccl1:
        # RCCL under the step on a 1-GPU box: backend "nccl" with ONE rank and the gradient exchange forced
        # (STY_DIST_FORCE_COLLECTIVE=1, dist.force_collective): every bucket goes through all_reduce(async_op=True) from the
        # library's gradient hooks, RCCL's streams run beside the trainer's four under GPU_MAX_HW_QUEUES=2.
        # Measured as a run of its own (`bench.py --rccl1` in a child process, the headline's K / W): as a second trainer inside
        # this process -- behind the headline and the extras -- the same step read 1.06-1.08 with or without RCCL in it
        # (profiles/r05_ab_env.txt block 8), which is this process's history, not the collective path.
        import subprocess
        import tempfile
        try:
            with tempfile.TemporaryDirectory() as td:
                cmd = [sys.executable, os.path.abspath(__file__), "--rccl1", "--steps", str(args.steps), "--warmup", str(args.warmup),
                       "--no-extra", "--no-cpu-baseline", "--detail", os.path.join(td, "detail.json")]
                env = dict(os.environ)
                for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                    env.pop(k, None)
                out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, check=True)
                r = json.loads(out.stdout.decode().strip().splitlines()[-1])
            rccl1 = {"what": "the c3 step with backend nccl (RCCL), world size 1, every gradient bucket all-reduced "
                             "(STY_DIST_FORCE_COLLECTIVE=1) -- the collective path of an N-GPU run minus the wire; a run of its own "
                             "(bench.py --rccl1 in a child process)",
                     "ms_per_step": r["ms_per_step"], "vs_no_process_group": r["ms_per_step"] / rec["ms_per_step"],
                     "host_issue_ms_per_step": r["host_issue_ms_per_step"], "backend": "nccl (RCCL)",
                     "allreduce_exposed_ms": (r.get("ranks") or {}).get("allreduce_exposed_ms"),
                     "exchange": (r.get("ranks") or {}).get("exchange"),
                     "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
        except Exception as e:  # never fail the bench line on this
            rccl1 = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank != 0:
        return
    rec["library"] = lib_info
    rec["ranks"] = {"world_size": world, "launcher": "torch.distributed.run, one process per GPU" if world > 1 else "single process",
                    "backend": (backend + (" (RCCL)" if not share else " (share-device test aid)")) if world > 1 else None,
                    "devices": 1 if share else world}
    rec["ranks"].update(rec.pop("rank_view", None) or {})
    if rccl1 is not None:
        rec["ranks"]["rccl_world1"] = rccl1
    if extras:
        rec["extra"] = extras
    if not args.no_cpu_baseline and world == 1:
        rec["cpu_baseline"] = cpu_baseline(WORKLOADS[args.workload])
    # the full record (per-family kernel tables, the extras' whole records) goes to the sidecar; stdout gets the ONE line
    detail = args.detail or os.path.join(ROOT, "bench_detail.json")
    paths = [detail] + ([os.path.join(ROOT, "gpurun_out", "bench_detail.json")]
                        if not args.detail and os.path.isdir(os.path.join(ROOT, "gpurun_out")) else [])
    written = None
    for pth in paths:
        try:
            with open(pth, "w") as f:
                json.dump(rec, f, indent=1)
            written = written or os.path.relpath(pth, ROOT)
        except OSError:
            pass
    sys.stdout.flush()
    os.write(out_fd, (format_line(rec, written) + "\n").encode())


if __name__ == "__main__":
    main()
