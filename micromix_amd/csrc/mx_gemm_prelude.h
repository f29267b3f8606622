// What a translation unit that includes mx_gemm_tile.inc needs in front of it (mx_gemm256.hip; developer probes include it too).
#pragma once
#include <hip/hip_ext.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "mx_acc_regs.h"
#ifndef MM_FP4_KD256
#define MM_FP4_KD256 1  // fp4 x fp4 segment on 256-deep slabs (whole cache lines per row); 0 = 128-deep slabs like the other segments
#endif
#ifndef MM_PRIO
#define MM_PRIO 1        // s_setprio for waves 4-7 of the 8-wave tiles (mx_gemm_tile.inc, tile_body); 0 = none
#endif
#include "mx_instrument.h"   // MM_DBG ablation switches and MM_CLOCKS: constant 0 unless built with -DMM_INSTRUMENT
#ifndef MM_CHAIN
#define MM_CHAIN 1  // chained segment hand-over on the 256-row tile (mx_gemm_tile.inc); 0 = every segment's own prologue (A/B builds)
#endif
#include "mx_common.h"
#include "mx_direct_convert.h"
#include "mx_kernels.h"

namespace mm {

// hipcc parses __device__ bodies in its host pass as well; gfx950 inline asm and target builtins only exist in
// the device pass, so those few bodies are compiled for the device only.
#if defined(__HIP_DEVICE_COMPILE__)
#define MM_DEVICE_ONLY(...) __VA_ARGS__
#else
#define MM_DEVICE_ONLY(...)
#endif

// in-kernel split-K (split_tile_reduce): scope of the ticket atomics and cache-policy bits of the partial-sum traffic
#define MM_SPLIT_SCOPE __HIP_MEMORY_SCOPE_AGENT
#define MM_SPLIT_AUX 16   // sc1 = device scope
#ifndef MM_SPLIT_FENCES
// 1 = spell the hand-over of the in-kernel split-K with agent-scope release / acquire fences and an acq_rel ticket (split_tile_reduce).
// Measured (round 4, tools/time_cases.py, k/v at M = 128, back-to-back launches through the Python shim, alternating processes):
// 23.3 / 23.6 us with the fences against 16.1 / 16.4 us without -- every wave's buffer_wbl2 sc1 walks the L2 although nothing of
// this kernel is dirty there -- so the default is 0: the same ordering from the instructions that are already needed (see there).
#define MM_SPLIT_FENCES 0
#endif

template <class KernelT>
static hipError_t launch_tile(KernelT kern, DynamicLdsOnce &attr, int lds_bytes, int tiles, int threads, const GemmArgs &a,
                              hipStream_t stream) {
    if (hipError_t e = attr.ensure(reinterpret_cast<const void *>(kern), lds_bytes); e != hipSuccess) return e;
    if (a.ev_start != nullptr && a.ev_stop != nullptr)
        hipExtLaunchKernelGGL(kern, dim3(tiles), dim3(threads), lds_bytes, stream, a.ev_start, a.ev_stop, 0, a);
    else
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(threads), lds_bytes, stream, a);
    return hipGetLastError();
}

// the tiles for launches with few tiles live in mx_gemm_tiles_small.hip (compiled beside mx_gemm256.hip)
hipError_t launch_small_tile(int kind, bool w4, bool splitk, int wgs, const GemmArgs &a, hipStream_t stream);
hipError_t launch_small_tile_grouped(int kind, bool w4, int total, const GroupedTileArgs &ga, hipStream_t stream);
constexpr size_t SMALL_PART_BYTES_64x64 = 16 * 256 * 4, SMALL_PART_BYTES_64x128 = 32 * 256 * 4;   // fp32 accumulators of one workgroup (in-kernel split-K)

}  // namespace mm
