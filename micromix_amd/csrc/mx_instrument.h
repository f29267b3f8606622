// Kernel-developer instrumentation of the tiled GEMM.  NOT part of the default build: libmicromix_hip.so is compiled without
// MM_INSTRUMENT, so every switch below is the constant 0 there, the in-kernel clock stores do not exist and
// mm_diag_set_clock_buffer is not exported.  The instrumented variant is built by
//     tools/build_variant.sh instr -DMM_INSTRUMENT [-DMM_DBG=<bits>]      -> micromix_amd/lib/dbg/lib_instr.so
// and selected with MICROMIX_HIP_LIB=<path> (tools/gemm_clock.py picks lib_instr.so up by itself).
//   MM_DBG bits (ablations, results are garbage): 1 = no MFMA, 2 = no DMA, 512 = no fragment reads in the loop;
//   1024 = no workgroup barriers, 2048 = no waits for the DMA, 4096 = every workgroup loads tile (0,0)'s operands (no L2 misses),
//   8192 = every workgroup loads its XCD's first tile's operands, 16384 = 128-byte operand rows fetched as half lines; 64-row tiles: 4 = no global loads, 8 = no LDS writes, 32 = no fragment reads.
#pragma once
#ifdef MM_INSTRUMENT
#ifndef MM_DBG
#define MM_DBG 0
#endif
#define MM_CLOCKS 1   // per-workgroup s_memtime / s_memrealtime stamps into GemmArgs::clock_out
#else
#ifdef MM_DBG
#error "MM_DBG ablation switches need -DMM_INSTRUMENT (tools/build_variant.sh)"
#endif
#define MM_DBG 0
#define MM_CLOCKS 0
#endif
