"""Import harness for the REAL reference (xiaosu-zhu/McQuic @ /root/reference), build container only.

TEST INFRASTRUCTURE ONLY.  Used by `tests/golden/make_golden.py` (fixture capture) and by
`tests/test_oracle_vs_reference.py` (skipped when /root/reference is absent, i.e. on the GPU box).
The reference sources are imported unmodified from the read-only tree; nothing of them is copied.

Why a harness is needed (SURVEY.md "Five facts", §8(c)):
  * `import mcquic` pulls marshmallow / vlutils / torchvision / fairscale, none of which is installed;
  * `EntropyCoder.__init__` starts with `raise NotImplementedError` (mcquic/modules/entropyCoder.py:17),
    so `Compressor(...)` cannot be constructed as shipped.
The harness pre-seeds `sys.modules` with package shells (so the package `__init__`s never run), stubs
the missing third-party modules, stubs `mcquic.rans` (only the tensor path is pinned here), and swaps in
an EntropyCoder subclass whose ctor performs the attribute set-up of the dead ctor body.
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import types

REF = os.environ.get("MCQUIC_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "mcquic", "modules"))


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
    m.__spec__.submodule_search_locations = [path]
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


class _Registry:
    """Stand-in for vlutils.base.Registry: generic alias + @register / @register("key")."""

    def __class_getitem__(cls, item):
        return cls

    def __init_subclass__(cls, **kw):
        cls._map = {}

    @classmethod
    def register(cls, key=None):
        if key is None or isinstance(key, str):
            return lambda t: cls._map.setdefault(key or t.__name__, t)
        cls._map[key.__name__] = key
        return key

    @classmethod
    def get(cls, key, logger=None):
        return cls._map[key]


class _Restorable:
    def __init__(self):
        pass


class _Field:
    def __init__(self, *a, **k):
        pass


_loaded = None


def load():
    """Returns the reference's `mcquic.modules.compressor` module with a constructible `Compressor`."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference tree not found at {REF}")
    import torch
    from torch import nn

    _stub("vlutils")
    _stub("vlutils.base", Registry=_Registry, Restorable=_Restorable)
    _stub("vlutils.logger", readableSize=lambda n: f"{n}B")
    _stub("marshmallow", Schema=_Field, post_load=lambda f: f, ValidationError=Exception, RAISE="raise",
          fields=types.SimpleNamespace(Field=_Field, Int=_Field, Str=_Field, List=_Field, Nested=_Field, Dict=_Field,
                                       Bool=_Field, Float=_Field))
    _stub("torchvision")
    _stub("torchvision.transforms")
    _stub("torchvision.transforms.functional")
    _stub("fairscale")
    _stub("fairscale.nn")
    _stub("fairscale.nn.checkpoint")
    _stub("fairscale.nn.checkpoint.checkpoint_activations", checkpoint_wrapper=lambda m: m)

    m = _pkg("mcquic", REF + "/mcquic")
    m.__version__ = "0.1.40"
    from mcquic.consts import Consts  # noqa: E402  (resolved through the shell's __path__)
    m.Consts = Consts
    _pkg("mcquic.data", REF + "/mcquic/data")
    _pkg("mcquic.modules", REF + "/mcquic/modules")
    # the native rANS coder sits beside the tensor path; byte streams are not pinned here
    _stub("mcquic.rans", pmfToQuantizedCDF=lambda pmf, prec: [], RansEncoder=lambda: None, RansDecoder=lambda: None)

    import mcquic.modules.entropyCoder as EC
    import mcquic.modules.quantizer as Q

    class RevivedEntropyCoder(EC.EntropyCoder):
        def __init__(self, m, k, ema=0.9):  # attribute set-up of the dead body, entropyCoder.py:18-26
            nn.Module.__init__(self)
            self.encoder, self.decoder = None, None
            self._freqEMA = nn.ParameterList(nn.Parameter(torch.ones(m, ki) / ki, requires_grad=False) for ki in k)
            self._k, self._ema, self._cdfs, self._normalizedFreq = k, ema, None, None

    Q.EntropyCoder = RevivedEntropyCoder
    import mcquic.modules.compressor as C
    _loaded = C
    return C


def load_validate_handlers():
    """The reference's validation handlers (mcquic/validate/handlers.py: MsSSIM, PSNR, BPP, IdealBPP) and the Decibel
    formatter (mcquic/validate/utils.py), imported unmodified.  handlers.py pulls torchvision and
    vlutils.metrics.meter.Handler at import time: both are stubbed (Handler = the three members the handlers use)."""
    load()

    class _Handler:                                   # vlutils.metrics.meter.Handler as used by handlers.py
        def __init__(self, format: str = r"%.2f"):
            self._format = format
            self.length = 0

        def reset(self):
            self.length = 0

        def to(self, device):
            return self

    _stub("vlutils.metrics")
    _stub("vlutils.metrics.meter", Handler=_Handler)
    tf = sys.modules["torchvision.transforms.functional"]
    tf.resize = tf.center_crop = tf.convert_image_dtype = None       # only InceptionScore (not used here) calls them
    _stub("torchvision.models", inception_v3=None)
    _pkg("mcquic.validate", REF + "/mcquic/validate")
    import mcquic.validate.handlers as H
    import mcquic.validate.utils as U
    return {"MsSSIM": H.MsSSIM, "PSNR": H.PSNR, "BPP": H.BPP, "IdealBPP": H.IdealBPP, "Decibel": U.Decibel}


def reference_compressor(channel, m, k, state_dict=None):
    """Construct the reference's Compressor (eval mode) and optionally load a state_dict into it."""
    C = load()
    model = C.Compressor(channel, m, k).eval()
    if state_dict is not None:
        missing, unexpected = model.load_state_dict(state_dict, strict=True), None
    return model
