"""Golden fixture of the reference's own loader (FilePathDataset / Collater / DynamicBatchSampler, train/dataloader.py) on
the synthetic dataset of tools/make_sample_dataset.py (build container only; soundfile / librosa / tqdm are absent and
stubbed: soundfile.info / read through the standard `wave` module with soundfile's float64 / 32768 scaling).

    python tools/gen_golden_data.py
Writes tests/golden/data_small.safetensors (first two batches) + data_small.json (length bins, batch order).
"""
import importlib.machinery
import json
import os
import sys
import tempfile
import types
import wave

import numpy as np
import torch
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import  # noqa: E402
from make_sample_dataset import make  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
N, SEED, BATCH = 20, 1234, 4


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def main():
    ref_import.install()

    class Info:
        def __init__(self, path):
            with wave.open(path, "rb") as f:
                self.frames, self.samplerate = f.getnframes(), f.getframerate()

    def sf_read(path):
        with wave.open(path, "rb") as f:
            x = np.frombuffer(f.readframes(f.getnframes()), dtype="<i2").astype(np.float64) / 32768.0
            return x, f.getframerate()
    stub("soundfile", info=Info, read=sf_read)
    lib = stub("librosa", resample=None)
    lib.filters = stub("librosa.filters", mel=None)

    class Tq:
        def __init__(self, iterable=None, **k):
            self.it = iterable

        def __iter__(self):
            return iter(self.it)

        def clear(self):
            pass

        def close(self):
            pass
    stub("tqdm", tqdm=Tq)
    from stylish_tts.train import dataloader as RD
    from stylish_tts.lib.text_utils import TextCleaner
    from stylish_tts.train.utils import DurationProcessor
    mc = ref_import.model_config()
    root = tempfile.mkdtemp()
    make(root, N, SEED)
    lines = open(os.path.join(root, "training-list.txt"), encoding="utf-8").read().splitlines()
    ds = RD.FilePathDataset(data_list=lines, root_path=os.path.join(root, "wav-dir"), text_cleaner=TextCleaner(mc.symbol),
                            model_config=mc, pitch_path=os.path.join(root, "pitch.safetensors"),
                            alignment_path=os.path.join(root, "alignment.safetensors"),
                            duration_processor=DurationProcessor(16, 50))
    bins, _ = ds.time_bins()

    class Stage:
        def get_batch_size(self, key):
            return BATCH

        def load_batch_sizes(self):
            pass

        def get_steps_per_epoch(self):
            return 0

    class Train:
        stage = Stage()
    sampler = RD.DynamicBatchSampler(bins, shuffle=True, seed=0, drop_last=False, epoch=1, train=Train())
    order = [list(map(int, b)) for b in sampler]
    coll = RD.Collater(stage="acoustic", hop_length=mc.hop_length)
    out = {}
    for bi in range(2):
        waves, texts, text_lengths, paths, pitches, alignments = coll([ds[i] for i in order[bi]])
        out.update({f"b{bi}.texts": texts, f"b{bi}.text_lengths": text_lengths,
                    f"b{bi}.pitches": pitches, f"b{bi}.alignments": alignments})
        out[f"b{bi}.waves"] = waves[:, ::7].contiguous()  # strided: keeps the fixture small, still pins padding/offsets
    save_file(out, os.path.join(OUT, "data_small.safetensors"))
    json.dump({"n": N, "seed": SEED, "batch": BATCH, "bins": {str(k): v for k, v in bins.items()}, "order": order},
              open(os.path.join(OUT, "data_small.json"), "w"))
    print("bins", bins)
    print("order", order)
    print(os.path.getsize(os.path.join(OUT, "data_small.safetensors")) / 1024, "KiB")


if __name__ == "__main__":
    main()
