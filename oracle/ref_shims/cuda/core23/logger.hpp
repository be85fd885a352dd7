// <core23/logger.hpp> stand-in -- TEST INFRASTRUCTURE ONLY (see ../cuda_runtime_api.h): the one name
// the reference's managed_allocator.cuh pulls in (`using HugeCTR::Logger;`); the logging macros it
// uses are those of ref_shims/common.hpp.
#pragma once
#include <common.hpp>
namespace HugeCTR {
class Logger {};
}  // namespace HugeCTR
