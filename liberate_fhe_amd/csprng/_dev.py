"""Pointer / stream plumbing shared by the sampler shims (no compute here)."""
from __future__ import annotations

import torch


def dev_stream(t: torch.Tensor, who: str):
    if t.device.type != "cuda":
        raise RuntimeError(f"{who}: tensor on {t.device}; the HIP kernels need device memory (no CPU fallback)")
    idx = t.device.index if t.device.index is not None else torch.cuda.current_device()
    return idx, torch.cuda.current_stream(idx).cuda_stream


def ptr(t: torch.Tensor, who: str, dtype=torch.int64):
    # Same checks as the reference's CHECK_INPUT (chacha20.cpp:8-11).
    if t.dtype != dtype:
        raise TypeError(f"{who}: expected a {dtype} tensor, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{who}: tensor must be contiguous")
    return t.data_ptr()
