// Shared helpers for libmmft_hip.so (gfx950 / CDNA4 only; wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mmft.h"

// Kernels whose packed-fp32 code would contain v_pk_fma_f32 with a LOW-half source select (`op_sel`: the low result takes the
// high register of a 64-bit operand pair) are compiled without packed fp32 instructions.  Measured on MI355X (DESIGN §3.7,
// "packed fma"): fc_prefix_kernel, replayed inside the whole-step HIP graph next to the sweep stream's MFMA kernels, lost the
// f * w term of one such instruction's low result in 16 lanes in ~5 % of its launches (inputs verified before and after the
// launch; 0 of 85 runs with plain v_fma_f32, 14 of 68 with the packed form).  tests/test_host_cpu.py checks that no kernel of
// the built library contains the form.  (The attribute exists on the device pass only.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MMFT_ALLOW_PACKED_OPSEL)      // make EXTRA=-DMMFT_ALLOW_PACKED_OPSEL: the reproducer's build
#define MMFT_NO_PACKED_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define MMFT_NO_PACKED_F32
#endif

namespace mmft {

void set_error(const char* fmt, ...);
int math_mode();            // MMFT_MATH_F32 / MMFT_MATH_BF16 (linear.hip)

struct DeviceGuard {
  int prev;
  bool switched;
  explicit DeviceGuard(int dev) : prev(-1), switched(false) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
      switched = (hipSetDevice(dev) == hipSuccess);
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return MMFT_ERR_LAUNCH;
  }
  return MMFT_OK;
}

// Kernels that need more dynamic LDS than the 64 KB default get hipFuncAttributeMaxDynamicSharedMemorySize raised once per
// (launch site, device): the attribute belongs to the device's code object, entry points run on the main thread and on
// autograd's worker thread, and a refused request must not surface later as an unrelated launch failure.
struct DynLdsOnce {
  unsigned long long done = 0;       // bit d: set for device d (devices >= 64 simply repeat the idempotent call)
};
inline int ensure_dyn_lds(DynLdsOnce& once, const void* kernel, int bytes, const char* what) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = dev < 64 ? (1ull << dev) : 0ull;
  if (__atomic_load_n(&once.done, __ATOMIC_ACQUIRE) & bit) return MMFT_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d): %s", what, bytes, hipGetErrorString(e));
    return MMFT_ERR_LAUNCH;
  }
  __atomic_fetch_or(&once.done, bit, __ATOMIC_RELEASE);
  return MMFT_OK;
}

#define MMFT_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      mmft::set_error(__VA_ARGS__);        \
      return MMFT_ERR_BAD_ARG;             \
    }                                      \
  } while (0)

// Optional launch profiling (mmft_prof_enable): HIP events recorded on the launch stream around every
// instrumented kernel, aggregated per kernel name by mmft_prof_report.  Off by default (zero overhead
// beyond one branch per launch).
bool prof_on();
// launches that are being recorded into a HIP graph cannot carry the profiler's events
inline bool stream_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}
void prof_begin(const char* name, double flops, double bytes, hipStream_t st);
void prof_end(hipStream_t st);
struct ProfScope {
  hipStream_t st;
  bool on;
  ProfScope(const char* name, double flops, double bytes, hipStream_t s) : st(s), on(prof_on() && !stream_capturing(s)) {
    if (on) prof_begin(name, flops, bytes, st);
  }
  ~ProfScope() {
    if (on) prof_end(st);
  }
};

// Single-kernel launch sites use hipExtLaunchKernelGGL while profiling: its start/stop events carry the kernel's own
// begin/end timestamps (what rocprofv3 reports), not the stream position of an event recorded from the host, so
// 8-20 us kernels are not inflated by launch gaps.
void prof_events(const char* name, double flops, double bytes, hipEvent_t* e0, hipEvent_t* e1);
void prof_commit();
#define MMFT_LAUNCH_LDS(name, flops, bytes, kernel, grid, block, lds, st, ...)                               \
  do {                                                                                                       \
    if (mmft::prof_on() && !mmft::stream_capturing(st)) {                                                    \
      hipEvent_t mmft_e0_, mmft_e1_;                                                                         \
      mmft::prof_events(name, flops, bytes, &mmft_e0_, &mmft_e1_);                                           \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, st, mmft_e0_, mmft_e1_, 0, __VA_ARGS__);               \
      mmft::prof_commit();                                                                                   \
    } else {                                                                                                 \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                         \
    }                                                                                                        \
  } while (0)
#define MMFT_LAUNCH(name, flops, bytes, kernel, grid, block, st, ...) \
  MMFT_LAUNCH_LDS(name, flops, bytes, kernel, grid, block, 0, st, __VA_ARGS__)

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// grid for memory-bound elementwise work: cap at ~8 blocks/CU and grid-stride the rest
inline int ew_grid(long long work_items, int block = 256) {
  long long g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 256 * 8) g = 256 * 8;
  return (int)g;
}

// slab reductions (linear.hip): partial results written by the workgroups of a split kernel, added in a fixed order
struct SlabSeg {
  const float* slabs;
  float* out;
  long long stride;      // floats between slabs
  int splits, elems, fold, accumulate;
};
int launch_slab_reduce_batch(const SlabSeg* segs, int n, hipStream_t st);

}  // namespace mmft
