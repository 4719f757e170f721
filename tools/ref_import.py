"""Import the read-only reference (this container only) behind stub modules.

Test/fixture infrastructure: used by tools/gen_golden.py to run the reference's own
PyTorch CPU path.  Nothing here travels to the GPU box as a dependency: /root/reference
does not exist there.  Stub list follows SURVEY.md Appendix D.
"""
import importlib.machinery
import sys
import types

REF_SRC = "/root/reference/src"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Munch(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def install():
    import torch

    if "stylish_tts" in sys.modules:
        return
    _stub("munch", Munch=_Munch)
    _stub("pynvml", nvmlInit=None, nvmlDeviceGetHandleByIndex=None, nvmlDeviceGetMemoryInfo=None)
    ta = _stub("torchaudio")
    ta.models = _stub("torchaudio.models", Conformer=object)

    class _MelScale(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    ta.transforms = _stub("torchaudio.transforms", MelScale=_MelScale)
    ta.functional = _stub("torchaudio.functional")
    for n in ("onnxruntime", "onnx", "pyloudnorm"):
        _stub(n)
    sys.path.insert(0, REF_SRC)


def model_config():
    install()
    from stylish_tts.lib.config_loader import load_model_config_yaml

    with open(REF_SRC + "/stylish_tts/train/config/model.yml", "r", encoding="utf-8") as f:
        return load_model_config_yaml(f)
