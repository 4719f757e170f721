"""Batch / wire formats of the reference's training loader, so that an existing dataset feeds the
HIP training step unchanged.  Stock PyTorch on the host; nothing here touches the GPU.

Counterparts (reference file:line, src/stylish_tts/...):
  TextCleaner          lib/text_utils.py:8-43        phoneme string -> token ids, pad symbol added on both sides
  get_time_bin / get_frame_count   train/dataloader.py:421-430   0.25 s length bins: frames = 20*bin + 60
  SampleDataset        train/dataloader.py:20-181 (FilePathDataset)   list line `wav|phonemes|speaker|text`; wav centre-padded
                       to its bin's frame count; pitch.safetensors / alignment.safetensors keyed by the wav path
  Collater             train/dataloader.py:184-261   (waves, texts, text_lengths, paths, pitches, alignments)
  LengthBinSampler     train/dataloader.py:303-418 (DynamicBatchSampler)   one length bin per batch, the same torch.Generator
                       draws (seed + epoch; randperm per bin, randint over the remaining-batches weights)
Differences, stated: wav files are read with the standard library (`wave`, PCM16) because soundfile / librosa are not part
of this image -- files must already be 24 kHz mono PCM16, as the reference's README asks; the duration-class weights
(dataloader.py:33-49) belong to the duration stage and are not computed.
"""
import os.path as osp
import wave as _wave

import numpy as np
import torch
from safetensors import safe_open

# train/config/model.yml:80-84 (symbol table: 178 tokens)
SYMBOLS = dict(
    pad="$",
    punctuation=";:,.!?¡¿—…\"()“” ",
    letters="ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz",
    letters_ipa="ɑɐɒæɓʙβɔɕçɗɖðʤəɘɚɛɜɝɞɟʄɡɠɢʛɦɧħɥʜɨɪʝɭɬɫɮʟɱɯɰŋɳɲɴøɵɸθœɶʘɹɺɾɻʀʁɽʂʃʈʧʉʊʋⱱʌɣɤʍχʎʏʑʐʒʔʡʕʢǀǁᵊǃˈˌːˑʼʴʰʱʲʷˠˤ˞↓↑→↗↘'̩'ᵻ",
)


class TextCleaner:
    def __init__(self, symbols=None):
        s = symbols or SYMBOLS
        self.pad = s["pad"]
        table = [s["pad"]] + list(s["punctuation"]) + list(s["letters"]) + list(s["letters_ipa"])
        self.index = {}
        for i, ch in enumerate(table):
            self.index[ch] = i  # later duplicates overwrite earlier ones, as the reference's dict build does

    def __call__(self, text):
        return [self.index[ch] for ch in self.pad + text + self.pad if ch in self.index]  # unknown symbols are skipped


def get_frame_count(bin_num):
    return bin_num * 20 + 20 + 40


def get_time_bin(sample_count, coarse_hop_length):
    frames = sample_count // coarse_hop_length
    return (frames - 20) // 20 if frames >= 20 else -1


def read_wav(path):
    """PCM16 wav -> (float64 samples in [-1, 1) as soundfile.read returns them, sample rate); first channel of stereo"""
    with _wave.open(path, "rb") as f:
        if f.getsampwidth() != 2:
            raise ValueError(f"{path}: only 16-bit PCM wav files are supported (got {8 * f.getsampwidth()}-bit)")
        n, ch, sr = f.getnframes(), f.getnchannels(), f.getframerate()
        x = np.frombuffer(f.readframes(n), dtype="<i2").astype(np.float64) / 32768.0
    if ch > 1:
        x = x.reshape(-1, ch)[:, 0]
    return x, sr


def wav_frames(path):
    with _wave.open(path, "rb") as f:
        return f.getnframes(), f.getframerate()


class SampleDataset(torch.utils.data.Dataset):
    def __init__(self, *, data_list, root_path, pitch_path, alignment_path, text_cleaner=None, sample_rate=24000,
                 hop_length=300, coarse_multiplier=1):
        self.pitch, self.alignment = {}, {}
        with safe_open(pitch_path, framework="pt", device="cpu") as f:
            for k in f.keys():
                self.pitch[k] = f.get_tensor(k)
        if alignment_path and osp.isfile(alignment_path):
            with safe_open(alignment_path, framework="pt", device="cpu") as f:
                for k in f.keys():
                    self.alignment[k] = f.get_tensor(k)
        self.data_list = []
        for line in data_list:
            fields = line.strip().split("|")
            if len(fields) != 4:
                raise ValueError("Dataset lines must have 4 |-delimited fields: " + line)
            self.data_list.append(fields)
        self.text_cleaner = text_cleaner or TextCleaner()
        self.root_path, self.sample_rate = root_path, sample_rate
        self.coarse_hop_length = hop_length * coarse_multiplier

    def time_bins(self):
        bins, seconds = {}, {}
        for i, (wav, phonemes, _, _) in enumerate(self.data_list):
            frames, sr = wav_frames(osp.join(self.root_path, wav))
            if sr != self.sample_rate:
                raise ValueError(f"{wav}: sample rate {sr}, expected {self.sample_rate} (resample the dataset first)")
            b = get_time_bin(frames, self.coarse_hop_length)
            if b == -1:
                raise ValueError(f"Segment Length Too Short. Must be at least 0.25 seconds: {wav}")
            if get_frame_count(b) < len(phonemes) or not 1 <= len(phonemes) <= 510:
                raise ValueError(f"Segment audio / phoneme count mismatch (dataloader.py:100-111): {wav}")
            bins.setdefault(b, []).append(i)
            seconds[b] = seconds.get(b, 0) + frames / self.sample_rate
        return bins, seconds

    def __len__(self):
        return len(self.data_list)

    def __getitem__(self, idx):
        wav_path, text, speaker_id, _ = self.data_list[idx]
        x, sr = read_wav(osp.join(self.root_path, wav_path))
        if sr != self.sample_rate:
            raise ValueError(f"{wav_path}: sample rate {sr}, expected {self.sample_rate}")
        pad_start = pad_end = 5000
        b = get_time_bin(x.shape[0], self.coarse_hop_length)
        if b != -1:
            total = get_frame_count(b) * self.coarse_hop_length
            pad_start = (total - x.shape[0]) // 2
            pad_end = total - x.shape[0] - pad_start
        x = torch.from_numpy(np.concatenate([np.zeros([pad_start]), x, np.zeros([pad_end])], axis=0)).float()
        tokens = torch.LongTensor(self.text_cleaner(text))
        pitch = torch.nan_to_num(self.pitch[wav_path].detach().clone()) if wav_path in self.pitch else None
        if wav_path in self.alignment:
            alignment = self.alignment[wav_path].detach()
        else:
            alignment = torch.zeros((3, tokens.shape[0]), dtype=torch.float32)
        return int(speaker_id), tokens, wav_path, x, pitch, alignment


class Collater:
    def __init__(self, *, stage, hop_length):
        self.stage, self.hop_length = stage, hop_length

    def __call__(self, batch):
        n = len(batch)
        max_text = max(b[1].shape[0] for b in batch)
        mel_length = batch[0][3].shape[-1] // self.hop_length
        texts = torch.zeros((n, max_text)).long()
        text_lengths = torch.zeros(n).long()
        paths = ["" for _ in range(n)]
        waves = torch.zeros((n, batch[0][3].shape[-1])).float()
        pitches = torch.zeros((n, mel_length)).float()
        alignments = torch.zeros((n, 1, max_text))
        for i, (_, text, path, wave, pitch, duration) in enumerate(batch):
            size = text.size(0)
            texts[i, :size] = text
            text_lengths[i] = size
            paths[i] = path
            waves[i] = wave
            if self.stage != "alignment":
                if pitch is None:
                    raise ValueError(f"Pitch not found for segment {path}")
                pitches[i] = pitch
            alignments[i, :1, :size] = duration[:1]
        return waves, texts, text_lengths, paths, pitches, alignments


class LengthBinSampler(torch.utils.data.Sampler):
    """DynamicBatchSampler: every batch comes from ONE length bin, so it is a dense equal-length tensor."""

    def __init__(self, time_bins, batch_size_of, shuffle=True, seed=0, drop_last=False, epoch=1):
        self.time_bins, self.batch_size_of = time_bins, batch_size_of
        self.shuffle, self.seed, self.drop_last, self.epoch = shuffle, seed, drop_last, epoch

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)
        samples = {}
        for key in self.time_bins.keys():
            if self.batch_size_of(key) <= 0:
                continue
            if self.shuffle:
                order = torch.randperm(len(self.time_bins[key]), generator=g)
                samples[key] = [self.time_bins[key][i] for i in order]
            else:
                samples[key] = self.time_bins[key]
        keys = list(samples.keys())
        while keys:
            index = 0
            if self.shuffle:
                total = sum(len(samples[k]) // self.batch_size_of(k) + 1 for k in keys)
                weight = torch.randint(0, total, [1], generator=g)[0]
                for i, k in enumerate(keys):
                    weight -= len(samples[k]) // self.batch_size_of(k) + 1
                    if weight <= 0:
                        index = i
                        break
            key = keys[index]
            cur = samples[key]
            bs = min(len(cur), self.batch_size_of(key))
            batch, rest = cur[:bs], cur[bs:]
            if len(rest) == 0 or (self.drop_last and len(rest) < bs):
                del samples[key]
            else:
                samples[key] = rest
            yield batch
            keys = list(samples.keys())

    def __len__(self):
        return sum(len(v) for v in self.time_bins.values())


def to_step_inputs(batch, device):
    """Collater tuple -> keyword arguments of AcousticTrainer.train_batch / acoustic_forward (stage_type.py:76-100:
    durations are row 0 of the alignment tensor)."""
    waves, texts, text_lengths, _, pitches, alignments = batch
    return dict(audio_gt=waves.to(device), texts=texts.to(device), text_lengths=text_lengths.to(device),
                pitch=pitches.to(device), durations=alignments[:, 0, :].to(device))
