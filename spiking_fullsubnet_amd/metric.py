"""The two energy proxies the recipes log next to the audio metrics, for the outputs of this package's modules.

Drop-ins for ``audiozen.metric.compute_synops`` / ``compute_neuronops`` (audiozen/metric.py:303-340; called at
recipes/intel_ndns/spiking_fullsubnet_freeze_phase/trainer.py:130-135 with the module's ``all_layer_outputs`` lists).
Each list is ``[layer input, spikes of layer 1, ..., spikes of layer L, projection]``; the reference reads, per spike
entry, its firing rate and its last dimension.  Entries may be the fp32 spike tensors (``layer_outputs="tensors"``,
the module default) or ``SpikeSummary`` objects (``layer_outputs="counts"``): exact device-side counts of the int8 spikes
the scan already writes, so the 4 B/spike tensors never exist.
"""
from __future__ import annotations

import torch

from .engine import SpikeSummary


def _rate(x) -> torch.Tensor:
    if isinstance(x, SpikeSummary):
        return x.rate()
    return torch.gt(x, 0).float().mean()  # audiozen/metric.py:306


def compute_synops(fb_all_layer_outputs, sb_all_layer_outputs, shared_weights=True) -> float:
    """sum over spike layers of rate * H * (fan_out + H), fp32 arithmetic as in the reference; doubled for unshared gates."""
    synops = 0.0
    for outs in [fb_all_layer_outputs] + list(sb_all_layer_outputs):
        for i in range(1, len(outs) - 1):
            synops = synops + _rate(outs[i]) * outs[i].size(-1) * (outs[i + 1].size(-1) + outs[i].size(-1))
    synops = float(synops.item()) if isinstance(synops, torch.Tensor) else float(synops)
    return synops if shared_weights else 2 * synops


def compute_neuronops(fb_all_layer_outputs, sb_all_layer_outputs) -> float:
    """audiozen/metric.py:330-340: the sum of the last dimensions of every entry."""
    n = 0.0
    for outs in [fb_all_layer_outputs] + list(sb_all_layer_outputs):
        for o in outs:
            n += o.size(-1)
    return n
