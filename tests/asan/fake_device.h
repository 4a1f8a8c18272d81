// Controls of the fake device (tests/asan/fake_device.cpp) for the harness.
#ifndef CALS_FAKE_DEVICE_H
#define CALS_FAKE_DEVICE_H
#include <cstddef>
#include <vector>
// true: stream operations run only when the host waits for the device (the latest schedule the API allows);
// false: inside the call that issued them (the earliest).  Switching drains the queue.
void fake_set_deferred(bool on);
// replay of a recorded run: the model admitted k-th is evicted once its iteration count reaches schedule[k]
// (empty vector: back to the built-in pseudo-convergence rule)
void fake_set_schedule(const std::vector<long long> &iters_at_eviction);
size_t fake_queue_depth();
// device allocations above this size fail with hipErrorOutOfMemory (0: no limit)
void fake_set_malloc_limit(size_t bytes);
#endif
