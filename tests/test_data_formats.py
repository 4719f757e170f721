"""N2 (batch / wire formats): stylish_tts_amd.data vs the reference's own loader classes on the same synthetic dataset
(fixture written by tools/gen_golden_data.py from the reference's FilePathDataset / Collater / DynamicBatchSampler)."""
import json
import os
import sys

import torch
from safetensors.torch import load_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
G = os.path.join(os.path.dirname(__file__), "golden")


def test_loader_matches_reference(tmp_path):
    from make_sample_dataset import make
    from stylish_tts_amd import data as D
    meta = json.load(open(os.path.join(G, "data_small.json")))
    gold = load_file(os.path.join(G, "data_small.safetensors"))
    root = str(tmp_path)
    make(root, meta["n"], meta["seed"])
    lines = open(os.path.join(root, "training-list.txt"), encoding="utf-8").read().splitlines()
    ds = D.SampleDataset(data_list=lines, root_path=os.path.join(root, "wav-dir"),
                         pitch_path=os.path.join(root, "pitch.safetensors"),
                         alignment_path=os.path.join(root, "alignment.safetensors"))
    bins, _ = ds.time_bins()
    assert {str(k): v for k, v in bins.items()} == meta["bins"]
    order = [list(map(int, b)) for b in D.LengthBinSampler(bins, lambda k: meta["batch"], shuffle=True, seed=0, epoch=1)]
    assert order == meta["order"]  # same torch.Generator draws as DynamicBatchSampler
    coll = D.Collater(stage="acoustic", hop_length=300)
    for bi in range(2):
        waves, texts, text_lengths, paths, pitches, alignments = coll([ds[i] for i in order[bi]])
        assert torch.equal(waves[:, ::7], gold[f"b{bi}.waves"])
        assert torch.equal(texts, gold[f"b{bi}.texts"])
        assert torch.equal(text_lengths, gold[f"b{bi}.text_lengths"])
        assert torch.equal(pitches, gold[f"b{bi}.pitches"])
        assert torch.equal(alignments, gold[f"b{bi}.alignments"])
        kw = D.to_step_inputs((waves, texts, text_lengths, paths, pitches, alignments), "cpu")
        assert kw["durations"].shape == texts.shape and kw["audio_gt"].shape[1] == 300 * pitches.shape[1]
        assert torch.allclose(kw["durations"].sum(1), torch.full((len(order[bi]),), float(pitches.shape[1])))


def test_text_cleaner_and_bins():
    from stylish_tts_amd import data as D
    tc = D.TextCleaner()
    assert len(tc.index) <= 178 and tc("a b")[0] == 0 and tc("a b")[-1] == 0 and len(tc("a b")) == 5
    assert tc("a@b") == tc("ab")  # symbols outside the table are dropped
    assert D.get_time_bin(5999, 300) == -1 and D.get_time_bin(6000, 300) == 0 and D.get_time_bin(24000, 300) == 3
    assert D.get_frame_count(3) == 120
