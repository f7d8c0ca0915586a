// Cross-stream ordering used inside the library (no counterpart in the reference, which launches everything on one stream:
// structural_loss.cpp:39,54,71,101,126): "what is enqueued on `to` from now on starts after everything enqueued on `from` so
// far".  One hipEvent_t per (device, recording stream), created on first use under a mutex and never destroyed — round 3 kept
// ONE function-local static event per call site, created on whichever device was current at the first call: two host threads,
// two engines or a second device in the process could record/wait on each other's event.  Re-recording an event
// is legal; a wait binds to the record that precedes it in host order, and the mutex keeps record+wait pairs of different
// threads apart.  Capturable: under hipStreamBeginCapture the pair becomes a graph edge.
#include "hp_common.h"
#include <map>
#include <mutex>
#include <utility>

int hp_order_streams(hipStream_t from, hipStream_t to) {
    if (from == to) return 0;
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, hipEvent_t> events;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    std::lock_guard<std::mutex> lock(mu);
    hipEvent_t& ev = events[std::make_pair(dev, from)];
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
        ev = nullptr;
        return (int)hipGetLastError();
    }
    if (hipEventRecord(ev, from) != hipSuccess) return (int)hipGetLastError();
    if (hipStreamWaitEvent(to, ev, 0) != hipSuccess) return (int)hipGetLastError();
    return 0;
}
