"""MI355X-native implementation of Spiking-FullSubNet's recurrent inference hot path.

Drop-in modules (same constructor keywords, state-dict names and forward() tuples as the reference):

* ``spiking_fullsubnet_amd.modeling_spiking_fullsubnet.SpikingFullSubNet``  (live recipes)
* ``spiking_fullsubnet_amd.model_low_freq.Separator``                        (frozen recipe / model_zoo checkpoints)

All compute between ``stft`` and ``istft`` runs in hand-written gfx950 kernels behind the C ABI of
``include/sfsn.h`` (``csrc/libsfsn_hip.so``).  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from .engine import Engine, PathSpec, SpikeSummary  # noqa: F401
from . import checkpoint, metric  # noqa: F401
from .model_low_freq import Separator  # noqa: F401
from .modeling_spiking_fullsubnet import SpikingFullSubNet  # noqa: F401
from .streaming import StreamingSession  # noqa: F401

__all__ = ["SpikingFullSubNet", "Separator", "Engine", "PathSpec", "StreamingSession"]
