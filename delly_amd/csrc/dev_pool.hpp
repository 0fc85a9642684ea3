// Device memory of the single-call helpers and the chromosome table: blocks come from, and go back to, the library's pool of
// parked blocks (DevPool in dellyhip.hip) -- nothing in the library calls hipFree while the process lives, see the memory
// policy in include/dellyhip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace dh {
hipError_t dev_alloc(void** p, size_t bytes);   // hipMalloc's contract
void dev_free(void* p);                         // nullptr is fine
}  // namespace dh
