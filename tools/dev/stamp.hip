// Development only: a one-thread kernel that writes the 100-MHz constant clock into a slot - launched in front of and behind
// every library call by tools/dev/gap_probe.py, to see the gaps between the kernels of a training step WITHOUT a profiler
// (rocprofv3 intercepts the queues and changes how barriers and events are processed).
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/dev/_build/libstamp.so tools/dev/stamp.hip
#include <hip/hip_runtime.h>
__global__ void stamp_kernel(unsigned long long* slot) { *slot = __builtin_amdgcn_s_memrealtime(); }
extern "C" int dev_stamp(void* slot, void* stream) {
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)slot);
    return (int)hipGetLastError();
}
