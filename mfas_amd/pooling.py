"""Feature-table construction from raw backbone taps.

``global_pool`` is ``GlobalPooling2D`` (/root/reference/models/auxiliary/aux_models.py:54-64, applied to every selected
tap at models/search/ntu_searchable.py:224-225) as a HIP reduction kernel; ``build_feature_table`` pools a dict of raw
taps — e.g. the 6-tuple / list that ``Visual.forward`` / ``Skeleton.forward`` return (models/central/ntu.py:35-50,
129-183), sliced like ntu_searchable.py:212-217 — into the HBM-resident table the engine trains on.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib
from .engine import FeatureTable


def global_pool(x: torch.Tensor, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """(B, C, *trailing) -> (B, C): mean over all trailing dims, f32 accumulation, on the tensor's HIP device."""
    if x.device.type != "cuda":
        raise RuntimeError("global_pool runs on a HIP device only")
    if x.dim() < 2:
        raise ValueError("expected (B, C, ...)")
    out_dtype = out_dtype or x.dtype
    if x.dim() == 2:
        return x.to(out_dtype)
    x = x.contiguous()
    B, Cc = x.shape[0], x.shape[1]
    inner = x[0, 0].numel()
    out = torch.empty((B, Cc), dtype=out_dtype, device=x.device)
    dt = _lib.MFAS_DT[str(x.dtype).replace("torch.", "")]
    odt = _lib.MFAS_DT[str(out_dtype).replace("torch.", "")]
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().mfas_global_pool(C.c_void_p(x.data_ptr()), dt, B * Cc, inner, C.c_void_p(out.data_ptr()), odt,
                                              C.c_void_p(stream)))
    return out


def build_feature_table(raw_taps: Dict[str, torch.Tensor], label: torch.Tensor, dtype=torch.bfloat16,
                        vlogit: Optional[torch.Tensor] = None, slogit: Optional[torch.Tensor] = None) -> FeatureTable:
    """raw_taps: {'s0'..'s3', 'v0'..'v3'} -> (N, C, ...) backbone maps (any trailing shape).  Returns the pooled table
    stored as `dtype`."""
    taps = {k: global_pool(v, dtype) for k, v in raw_taps.items()}
    return FeatureTable(taps, label, vlogit, slogit)
