// HIP streams restricted to a subset of the compute units (hipExtStreamCreateWithCUMask).
//
// The training step has two independent branches - the netlist sweep (a chain of ~100 small dependent launches) and the
// U-Net (a few dozen chip-filling ones).  On two ordinary streams they overlap only at kernel tails: whichever kernel is
// running holds every CU's workgroup slots and the other queue waits (rocprofv3 kernel trace: 6.2 of 7.0 ms with ONE
// kernel in flight).  Giving each branch its own share of every XCD's CUs lets both run all the time; neither is
// compute bound, and HBM bandwidth is shared by demand.
#include "common.h"
#include <vector>

using namespace mmft;

extern "C" int mmft_stream_create_cu_mask(int device, const unsigned int* mask, int nwords, long long* stream_out) {
  MMFT_REQUIRE(mask && nwords > 0 && stream_out, "stream_create_cu_mask: null pointer / empty mask");
  bool any = false;
  for (int i = 0; i < nwords; ++i) any = any || mask[i] != 0;
  MMFT_REQUIRE(any, "stream_create_cu_mask: the mask selects no compute unit");
  DeviceGuard dg(device);
  hipStream_t st = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, mask);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("stream_create_cu_mask: %s", hipGetErrorString(e));
    return MMFT_ERR_LAUNCH;
  }
  *stream_out = (long long)(uintptr_t)st;
  return MMFT_OK;
}

extern "C" int mmft_stream_destroy(long long stream) {
  if (!stream) return MMFT_OK;
  hipError_t e = hipStreamDestroy((hipStream_t)(uintptr_t)stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("stream_destroy: %s", hipGetErrorString(e));
    return MMFT_ERR_LAUNCH;
  }
  return MMFT_OK;
}

// one record per workgroup: (XCC id << 16) | (HW_ID & 0xffff)  (HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13)
__global__ void cu_census_kernel(unsigned int* __restrict__ out, int spin) {
  unsigned int xcc = __builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 20) & 0xf;      // HW_REG_XCC_ID = 20, bits 3:0
  unsigned int hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);           // HW_REG_HW_ID = 4, bits 15:0
  // hold the CU for a while so that a grid of one workgroup per slot spreads over every CU the queue may use
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {
  }
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | (hw & 0xffff);
}

/* Diagnostic: launches n_wg workgroups of 1024 threads (each spinning ~spin_ticks of the 100 MHz wall clock) and records
 * where each ran.  out[i] = (XCC id << 16) | (HW_ID & 0xffff). */
extern "C" int mmft_debug_cu_census(unsigned int* out, int n_wg, int spin_ticks, int device, void* stream) {
  MMFT_REQUIRE(out && n_wg > 0 && spin_ticks >= 0 && spin_ticks <= 100000, "debug_cu_census: bad arguments");
  DeviceGuard dg(device);
  hipLaunchKernelGGL(cu_census_kernel, dim3(n_wg), dim3(1024), 0, (hipStream_t)stream, out, spin_ticks);
  return check_launch("debug_cu_census");
}

extern "C" int mmft_device_cu_count(int device) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return p.multiProcessorCount;
}
