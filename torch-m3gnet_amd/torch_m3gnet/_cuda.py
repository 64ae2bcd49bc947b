"""Cheap forms of two torch.cuda helpers the wrappers call on every C-ABI call.  A caller that waits for the device every step
(molecular dynamics: the skin test's verdict) pays the host time between its launches in full, and `torch.cuda.current_stream()`
(~9 us: device-index resolution + a Stream object) and `with torch.cuda.device(dev)` (~8 us when `dev` is already current) were a
third of it on the refill path (profiles/r05_md_host_profile.txt)."""
from __future__ import annotations

import ctypes as C

import torch

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr() -> C.c_void_p:
    """The current stream of the current device, as the `void* stream` argument of the C ABI."""
    if _raw_stream is not None and _get_device is not None:
        return C.c_void_p(_raw_stream(_get_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class on_device:
    """`with torch.cuda.device(dev)`, free when `dev` is the current device already."""

    __slots__ = ("idx", "prev")

    def __init__(self, dev):
        self.idx = dev.index if isinstance(dev, torch.device) else dev
        self.prev = -1

    def __enter__(self):
        if self.idx is None:
            return self
        cur = _get_device() if _get_device is not None else torch.cuda.current_device()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False
