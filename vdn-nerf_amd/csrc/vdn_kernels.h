// Internal: maps the C-ABI structs of include/vdn_render.h into namespace vdn.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/vdn_render.h"

namespace vdn {
using SdfArgs = ::VdnSdfArgs;
using WeightNormDesc = ::VdnWeightNormDesc;
using ChunkDesc = ::VdnChunkDesc;
using RenderNetArgs = ::VdnRenderNetArgs;
using NerfArgs = ::VdnNerfArgs;
using CoarseArgs = ::VdnCoarseArgs;
using UpsampleArgs = ::VdnUpsampleArgs;
using MergeArgs = ::VdnMergeArgs;
using SectionArgs = ::VdnSectionArgs;
using CompositeArgs = ::VdnCompositeArgs;
using RenderNetBwdArgs = ::VdnRenderNetBwdArgs;
using NerfBwdArgs = ::VdnNerfBwdArgs;
using SdfRbarArgs = ::VdnSdfRbarArgs;
using SdfFbarArgs = ::VdnSdfFbarArgs;
using DwDesc = ::VdnDwDesc;
using DwFinalizeDesc = ::VdnDwFinalizeDesc;
using WeightNormBwdDesc = ::VdnWeightNormBwdDesc;
using CompositeBwdArgs = ::VdnCompositeBwdArgs;
using LossArgs = ::VdnLossArgs;
using RayAdjointArgs = ::VdnRayAdjointArgs;
using TrainPrepArgs = ::VdnTrainPrepArgs;
using GenRaysArgs = ::VdnGenRaysArgs;

// Kernels needing more than 64 KiB of dynamic LDS opt in once per process.
template <class K>
inline void allow_big_lds(K kernel, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
}  // namespace vdn
