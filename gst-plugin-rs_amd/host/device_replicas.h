// One replica of a device-side resource per device ordinal, created on first use on that device and kept until the owner dies
// (round 6: an `mvfx_cube_lut` used from streaming threads on several GPUs keeps a copy on each instead of freeing and re-uploading on every
// switch; the model is the reference's d3d12colorlut, which rebuilds its context for the device of the incoming memory,
// video/colorlut/src/d3d12colorlut/imp.rs:494-542).  Host-only, no HIP types: tests/replica_table_test.cpp exercises it with fake ordinals.
#pragma once

#include <atomic>
#include <memory>
#include <mutex>

namespace mvfx {

template <typename T, int MAX_DEVICES = 64>
class DeviceReplicas {
public:
    DeviceReplicas() { for (auto &s : slot_) s.store(nullptr, std::memory_order_relaxed); }
    ~DeviceReplicas() { for (auto &s : slot_) delete s.load(std::memory_order_relaxed); }
    DeviceReplicas(const DeviceReplicas &) = delete;
    DeviceReplicas &operator=(const DeviceReplicas &) = delete;

    static constexpr int capacity() { return MAX_DEVICES; }

    // the replica of `device`, or nullptr when none has been made (or the ordinal is out of range).  Lock-free: a published replica never moves.
    T *find(int device) const { return in_range(device) ? slot_[device].load(std::memory_order_acquire) : nullptr; }

    // the replica of `device`, made by `make()` (-> T *, nullptr on failure) if this is the device's first use; *created says which.  Two threads
    // asking for the same new device: one `make()` runs, both get its result.  An ordinal outside 0 .. MAX_DEVICES - 1 yields nullptr.
    template <typename MAKE>
    T *get_or_create(int device, MAKE &&make, bool *created = nullptr)
    {
        if (created) *created = false;
        if (!in_range(device)) return nullptr;
        if (T *have = slot_[device].load(std::memory_order_acquire)) return have;
        std::lock_guard<std::mutex> g(mu_);
        if (T *have = slot_[device].load(std::memory_order_acquire)) return have;
        T *fresh = make();
        if (!fresh) return nullptr;
        slot_[device].store(fresh, std::memory_order_release);
        if (created) *created = true;
        return fresh;
    }

    // f(device, T &) for every replica made so far, in ordinal order
    template <typename F>
    void for_each(F &&f) const
    {
        for (int d = 0; d < MAX_DEVICES; d++)
            if (T *r = slot_[d].load(std::memory_order_acquire)) f(d, *r);
    }

    int count() const
    {
        int n = 0;
        for_each([&](int, T &) { n++; });
        return n;
    }

private:
    static bool in_range(int device) { return device >= 0 && device < MAX_DEVICES; }
    std::atomic<T *> slot_[MAX_DEVICES];
    std::mutex mu_;
};

} // namespace mvfx
