// agx_device_guard.h - HOST side: "make device d current for this call, put the caller's device back afterwards".
// Every C-ABI entry point that touches the device opens one (agx_api.hip), so a context created on cuda:3 works from a
// thread whose current device is cuda:0 - the case an 8-GPU node has and a one-GPU test box cannot run.  The device calls
// come through a policy class so that tests/device_guard_harness.cpp can drive the logic with a mocked runtime.
#pragma once

namespace agx {

template <class Api>      // Api::get(int *dev) / Api::set(int dev) return 0 on success
struct DeviceGuardT {
    int prev = -1;        // the caller's device, -1 when it could not be read (then nothing is restored)
    bool switched = false;
    bool ok = true;       // false: the context's device could not be made current (the launch that follows will fail loudly)
    explicit DeviceGuardT(int dev) {
        if (Api::get(&prev) != 0) prev = -1;
        if (prev != dev) {
            ok = Api::set(dev) == 0;
            switched = ok;
        }
    }
    ~DeviceGuardT() {
        if (switched && prev >= 0) (void)Api::set(prev);
    }
    DeviceGuardT(const DeviceGuardT &) = delete;
    DeviceGuardT &operator=(const DeviceGuardT &) = delete;
};

}  // namespace agx
