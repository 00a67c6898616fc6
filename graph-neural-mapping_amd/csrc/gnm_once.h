// Per-device "done once" flags and the read-once environment helper (no HIP types: also used by host.cpp).
#pragma once
#include <atomic>
#include <stdlib.h>
static constexpr int kGnmMaxDevices = 64;
struct GnmDeviceOnce {
    std::atomic<unsigned char> done[kGnmMaxDevices];
    // true exactly when `dev` has not been marked yet (ordinals outside the table are never cached)
    bool first_use(int dev) const {
        return dev < 0 || dev >= kGnmMaxDevices || done[dev].load(std::memory_order_acquire) == 0;
    }
    void mark(int dev) {
        if (dev >= 0 && dev < kGnmMaxDevices) done[dev].store(1, std::memory_order_release);
    }
};
