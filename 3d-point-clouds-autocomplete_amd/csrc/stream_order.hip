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
    // The event lives on the device `from` belongs to — not on whatever device the calling thread has current (an engine may
    // be driven from a thread whose current device is another one: ADVICE r4).  The null stream has no device of its own:
    // it is the current device's.
    int cur = 0, dev = 0;
    if (hipGetDevice(&cur) != hipSuccess) return (int)hipGetLastError();
    dev = cur;
    if (from && hipStreamGetDevice(from, &dev) != hipSuccess) {
        (void)hipGetLastError();      // (a runtime without the query: fall back to the current device)
        dev = cur;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (events.size() > 4096) {       // streams come and go (a test suite creates hundreds): do not grow without bound
        for (auto& kv : events)
            if (kv.second) (void)hipEventDestroy(kv.second);      // legal with waits pending: released when they complete
        events.clear();
    }
    hipEvent_t& ev = events[std::make_pair(dev, from)];
    int rc = 0;
    if (dev != cur && hipSetDevice(dev) != hipSuccess) return (int)hipGetLastError();
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
        ev = nullptr;
        rc = (int)hipGetLastError();
    }
    if (!rc && hipEventRecord(ev, from) != hipSuccess) rc = (int)hipGetLastError();
    if (!rc && hipStreamWaitEvent(to, ev, 0) != hipSuccess) rc = (int)hipGetLastError();
    if (dev != cur) (void)hipSetDevice(cur);
    return rc;
}
