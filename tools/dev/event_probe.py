"""What a cross-stream dependency costs (dev): side stream runs a kernel, records an event; the main stream waits for it and runs a
stamp kernel (tools/dev/stamp.hip). Latency = main's stamp - the stamp right behind the side stream's kernel, for events created
with different flags through the HIP runtime directly, and for an already-completed event.  usage: event_probe.py"""
import os, sys, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
hip = None
for line in open("/proc/self/maps"):
    if "libamdhip64" in line:
        hip = ctypes.CDLL(line.split()[-1]); break
assert hip is not None
st = ctypes.CDLL(os.path.join(ROOT, "tools", "dev", "_build", "libstamp.so"))
st.dev_stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
dev = torch.device("cuda:0")
main, side = torch.cuda.Stream(), torch.cuda.Stream()
slots = torch.zeros(4096, dtype=torch.int64, device=dev)
big = torch.zeros(64 << 20, dtype=torch.float32, device=dev)       # 256 MB
small = torch.zeros(1024, dtype=torch.float32, device=dev)
FLAGS = {"default(0)": 0x0, "disable_timing": 0x2, "disable_timing|no_system_fence": 0x2 | 0x20000000,
         "disable_timing|release_to_device": 0x2 | 0x40000000, "disable_timing|release_to_system": 0x2 | 0x80000000}
def P(i):
    return slots.data_ptr() + 8 * i
def run(flags, work, completed, reps=30):
    ev = ctypes.c_void_p()
    assert hip.hipEventCreateWithFlags(ctypes.byref(ev), flags) == 0
    lat, after = [], []
    for r in range(reps):
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            work.mul_(1.0001)
        st.dev_stamp(P(0), side.cuda_stream)
        assert hip.hipEventRecord(ev, side.cuda_stream) == 0
        if completed:
            torch.cuda.synchronize()
            st.dev_stamp(P(0), main.cuda_stream)
        assert hip.hipStreamWaitEvent(main.cuda_stream, ev, 0) == 0
        st.dev_stamp(P(1), main.cuda_stream)
        st.dev_stamp(P(2), main.cuda_stream)
        torch.cuda.synchronize()
        t = slots[:3].cpu().numpy()
        lat.append((t[1] - t[0]) / 100.0); after.append((t[2] - t[1]) / 100.0)
    return np.median(lat), np.min(lat), np.median(after)
print("%-40s %-8s %-10s %10s %8s %12s" % ("event flags", "work", "event", "median us", "min", "next stamp"))
for name, fl in FLAGS.items():
    for wname, work in (("256MB", big), ("4KB", small)):
        for completed in (False, True):
            m, mn, a = run(fl, work, completed)
            print("%-40s %-8s %-10s %10.1f %8.1f %12.1f" % (name, wname, "complete" if completed else "pending", m, mn, a))
# in-queue: stamp, record, stamp on ONE stream
for name, fl in FLAGS.items():
    ev = ctypes.c_void_p(); hip.hipEventCreateWithFlags(ctypes.byref(ev), fl)
    v = []
    for r in range(30):
        torch.cuda.synchronize()
        with torch.cuda.stream(main):
            big.mul_(1.0001)
        st.dev_stamp(P(0), main.cuda_stream)
        hip.hipEventRecord(ev, main.cuda_stream)
        st.dev_stamp(P(1), main.cuda_stream)
        torch.cuda.synchronize()
        t = slots[:2].cpu().numpy(); v.append((t[1] - t[0]) / 100.0)
    print("in-queue record between two stamps, %-40s median %.1f us" % (name, np.median(v)))
v = []
for r in range(30):
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        big.mul_(1.0001)
    st.dev_stamp(P(0), main.cuda_stream); st.dev_stamp(P(1), main.cuda_stream)
    torch.cuda.synchronize()
    t = slots[:2].cpu().numpy(); v.append((t[1] - t[0]) / 100.0)
print("two stamps back to back: median %.1f us" % np.median(v))
