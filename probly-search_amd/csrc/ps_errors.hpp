// ps_errors.hpp — exception types the C ABI maps onto ps_status codes (ps_capi.cpp: guard()).
//   std::invalid_argument -> PS_EINVAL        std::length_error -> PS_EUNSUPPORTED
//   std::bad_alloc        -> PS_ENOMEM        NoDeviceError     -> PS_ENODEVICE
//   RcclError             -> PS_ERCCL         anything else     -> PS_EHIP
#pragma once
#include <stdexcept>
#include <string>

namespace ps {

struct NoDeviceError : std::runtime_error {
  explicit NoDeviceError(const std::string& m) : std::runtime_error(m) {}
};
struct RcclError : std::runtime_error {
  explicit RcclError(const std::string& m) : std::runtime_error(m) {}
};
// hipErrorOutOfMemory from an allocation (still PS_EHIP at the ABI): the one device error ps_snapshot_update answers
// by releasing the old planes and trying again
struct DeviceOom : std::runtime_error {
  explicit DeviceOom(const std::string& m) : std::runtime_error(m) {}
};

}  // namespace ps
