// hipFree / hipHostFree / hipDeviceSynchronize wait for EVERYTHING enqueued on the device -- with a persistent latency-path workgroup resident (srn_index_serve_start,
// srn_serve.hip) that is "until it leaves".  Every such call of this library therefore goes through these wrappers: the resident workgroups are told to leave first
// (they come back at the next srn_predict that wants them).  Include this header LAST in a translation unit that frees or synchronises.
#pragma once
#include <hip/hip_runtime.h>

namespace srn {
void serve_quiesce_all();   // every resident workgroup of this process leaves; returns when they have (no-op when none is resident: one atomic load)
void serve_hold_begin();    // ... and none is started again until serve_hold_end(): "leave, then free" is not enough -- a caller of srn_predict may start the launch again
void serve_hold_end();      //     between the two, and the free then waits for THAT one (seen: an 8-second call, 3 s of traffic + the 5 s idle timeout)
struct ServeHold { ServeHold() { serve_hold_begin(); } ~ServeHold() { serve_hold_end(); } ServeHold(const ServeHold&) = delete; ServeHold& operator=(const ServeHold&) = delete; };
inline hipError_t quiesced_free(void* p) { ServeHold h; return hipFree(p); }
inline hipError_t quiesced_host_free(void* p) { ServeHold h; return hipHostFree(p); }
inline hipError_t quiesced_device_sync() { ServeHold h; return hipDeviceSynchronize(); }
}  // namespace srn
#define hipFree(p) ::srn::quiesced_free(p)
#define hipHostFree(p) ::srn::quiesced_host_free(p)
#define hipDeviceSynchronize() ::srn::quiesced_device_sync()
