// Device side of the call path (call.hip).
#pragma once
#include "call_host.h"
#include <memory>

namespace pantas {

struct DeviceCall;
struct DeviceCallDeleter { void operator()(DeviceCall*) const; };
using DeviceCallPtr = std::unique_ptr<DeviceCall, DeviceCallDeleter>;

struct CallTimes {
  float annotated_ms = 0, novel_ms = 0, around_ms = 0;   // count + scan + write kernels of each pass (HIP events)
  float passes_ms = 0;                                   // all passes, first launch to last end: they run side by side (round 5)
  uint64_t junctions = 0;                                // junction evaluations launched
};

struct CallOutput {
  RawVector<EventRec> recs;      // tens of millions at whole-genome scale: resized once, never value-initialised
  RawVector<uint32_t> side;      // record payloads
  std::vector<TaskOut> tasks;    // outcome of the deferred intron walks (indexed by EventRec.j1 of IR records)
  std::vector<uint32_t> paths;   // walked paths
};

DeviceCallPtr upload_call(const Graph& g, const CallTables& t, int device);
// Runs the three passes; fills records + payload (host).  Throws where the reference raises.
void run_call(DeviceCall& d, const Graph& g, const CallTables& t, const CallOptions& o, CallOutput& out, CallTimes& times);

}  // namespace pantas
