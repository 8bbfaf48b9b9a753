"""HIP streams the torch API cannot make: a LOW-priority stream (torch.cuda.Stream offers normal and high only), created through the
HIP runtime torch has loaded and wrapped as a torch ExternalStream. Plumbing for the Trainer's side stream (vdn_hip/train.py)."""
import ctypes

import torch

_hip = None


def _runtime():
    global _hip
    if _hip is None:
        torch.cuda.init()
        for line in open("/proc/self/maps"):            # the copy of libamdhip64 this process already runs on
            if "libamdhip64" in line:
                _hip = ctypes.CDLL(line.split()[-1])
                break
        if _hip is None:
            raise RuntimeError("libamdhip64 is not loaded in this process")
        _hip.hipStreamCreateWithPriority.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint, ctypes.c_int]
        _hip.hipDeviceGetStreamPriorityRange.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    return _hip


_low = {}


def low_priority_stream(device):
    """One low-priority stream per device and process (streams share a few hardware queues: the fewer, the fewer collisions;
    the Trainers of a process never step concurrently). Falls back to an ordinary torch stream if the runtime offers no range."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx in _low:
        return _low[idx]
    hip = _runtime()
    least, greatest = ctypes.c_int(0), ctypes.c_int(0)
    with torch.cuda.device(idx):
        rc = hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest))
        if rc != 0 or least.value <= 0:                 # (numerically larger = lower priority; 0 = normal)
            s = torch.cuda.Stream(device=dev)
        else:
            h = ctypes.c_void_p()
            rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, least.value)          # 1 = hipStreamNonBlocking
            if rc != 0:
                raise RuntimeError("hipStreamCreateWithPriority failed with %d" % rc)
            s = torch.cuda.ExternalStream(h.value, device=dev)
    _low[idx] = s
    return s
