// alloc.h -- every device / pinned allocation of libbfhip.so's engines goes through these two
// (defined in bfhip.hip), so that the tests can make the n-th one fail
// (bfhip_selftest_fail_alloc, include/bfhip.h) and walk every out-of-memory path.
#pragma once
#include <hip/hip_runtime.h>

extern "C" {
hipError_t bfhip_internal_dev_alloc(void **p, size_t bytes);
hipError_t bfhip_internal_pin_alloc(void **p, size_t bytes, unsigned int flags);
}
