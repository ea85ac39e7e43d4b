// Stream / event plumbing the data-parallel step needs and PyTorch-ROCm does not expose: an event recorded by an
// event-record NODE inside a captured HIP graph (hipEventRecordWithFlags + hipEventRecordExternal), which a stream
// outside the graph can wait for after every replay.  mmft.dist.GradReducer uses it to start a gradient bucket's RCCL
// all-reduce while the rest of the captured backward pass is still running (torch.cuda.Event(external=True) raises
// "External events are disallowed in rocm" on this build).
#include "common.h"

using namespace mmft;

extern "C" {

int mmft_event_create(void** event) {
  MMFT_REQUIRE(event, "event_create: null pointer");
  hipEvent_t e;
  hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
  if (rc != hipSuccess) {
    set_error("event_create: %s", hipGetErrorString(rc));
    return MMFT_ERR_LAUNCH;
  }
  *event = (void*)e;
  return MMFT_OK;
}

int mmft_event_destroy(void* event) {
  if (!event) return MMFT_OK;
  hipError_t rc = hipEventDestroy((hipEvent_t)event);
  if (rc != hipSuccess) {
    set_error("event_destroy: %s", hipGetErrorString(rc));
    return MMFT_ERR_LAUNCH;
  }
  return MMFT_OK;
}

int mmft_event_record(void* event, int external, int device, void* stream) {
  MMFT_REQUIRE(event, "event_record: null event");
  DeviceGuard dg(device);
  hipError_t rc = external ? hipEventRecordWithFlags((hipEvent_t)event, (hipStream_t)stream, hipEventRecordExternal)
                           : hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
  if (rc != hipSuccess) {
    set_error("event_record: %s", hipGetErrorString(rc));
    return MMFT_ERR_LAUNCH;
  }
  return MMFT_OK;
}

int mmft_stream_wait_event(void* stream, void* event, int device) {
  MMFT_REQUIRE(event, "stream_wait_event: null event");
  DeviceGuard dg(device);
  hipError_t rc = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
  if (rc != hipSuccess) {
    set_error("stream_wait_event: %s", hipGetErrorString(rc));
    return MMFT_ERR_LAUNCH;
  }
  return MMFT_OK;
}

int mmft_stream_is_capturing(void* stream, int device) {
  DeviceGuard dg(device);
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &st) != hipSuccess) return 0;
  return st == hipStreamCaptureStatusActive ? 1 : 0;
}

}  // extern "C"
