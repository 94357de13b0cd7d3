// Internal helpers shared by the libmdx translation units: error reporting,
// per-device stream, scoped device buffers.  Not part of the public C-ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <mutex>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <new>
#include <thread>
#include <vector>

#include "mdx.h"

namespace mdx {

// Thread-local text of the last error, returned by mdx_last_error().
char *error_buffer();
int fail(int code, const char *fmt, ...);

// The library stream of the current device (created on first use).
int get_stream(hipStream_t *stream);
// Second per-device stream, for host-to-device copies that overlap with kernels.
int get_copy_stream(hipStream_t *stream);

#define MDX_HIP(call)                                                          \
  do {                                                                         \
    hipError_t mdx_err_ = (call);                                              \
    if (mdx_err_ != hipSuccess)                                                \
      return ::mdx::fail(mdx_err_ == hipErrorOutOfMemory ? MDX_ERR_OUT_OF_MEMORY \
                                                        : MDX_ERR_DEVICE,      \
                         "%s failed: %s (%s:%d)", #call,                       \
                         hipGetErrorString(mdx_err_), __FILE__, __LINE__);     \
  } while (0)

#define MDX_TRY(call)                                                          \
  do {                                                                         \
    int mdx_rc_ = (call);                                                      \
    if (mdx_rc_ != MDX_OK) return mdx_rc_;                                     \
  } while (0)

#define MDX_REQUIRE(cond, ...)                                                 \
  do {                                                                         \
    if (!(cond)) return ::mdx::fail(MDX_ERR_INVALID_ARGUMENT, __VA_ARGS__);    \
  } while (0)

// Device allocation freed when it leaves scope (after the caller synchronised).
// Device memory for the library's own buffers comes from a per-device cache of
// freed blocks: every kernel and copy of the library runs on one stream per
// device, so a block released in program order can be handed out again at once,
// and repeated calls stop paying for hipMalloc / hipFree (which synchronise the
// device).  pool_trim returns the cached blocks of every device to the driver.
int pool_acquire(void **ptr, size_t bytes, size_t *capacity, int *device);
void pool_release(void *ptr, size_t capacity, int device);
void pool_trim();
// Counts one call to hipMalloc / hipHostMalloc (mdx_allocation_counters).
void note_driver_allocation();

struct DeviceBuffer {
  void *ptr = nullptr;
  size_t bytes = 0;     // requested size
  size_t capacity = 0;  // size of the underlying block
  int device = -1;
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
  ~DeviceBuffer() { release(); }
  int alloc(size_t n) {
    release();
    if (n == 0) n = 16;
    const int rc = pool_acquire(&ptr, n, &capacity, &device);
    if (rc != MDX_OK) {
      ptr = nullptr;
      return rc;
    }
    bytes = n;
    return MDX_OK;
  }
  // Grow-only reuse: keeps the allocation when it is already large enough.
  int reserve(size_t n) {
    if (ptr != nullptr && capacity >= n) {
      if (bytes < n) bytes = n;
      return MDX_OK;
    }
    return alloc(n);
  }
  void release() {
    if (ptr) pool_release(ptr, capacity, device);
    ptr = nullptr;
    bytes = 0;
    capacity = 0;
  }
  template <typename T>
  T *as() const {
    return static_cast<T *>(ptr);
  }
};

// Returns a device pointer for `src`: the pointer itself when it already lives on
// the device, else a staged copy held by `staging` (async copy on `stream`).
int to_device(const void *src, int location, size_t bytes, DeviceBuffer &staging,
              hipStream_t stream, const void **dptr);

// Host-to-device copy in stream order; large ranges are pipelined through page-locked
// memory.  The host range has been read when the call returns.
int upload(void *dst, const void *src, size_t bytes, hipStream_t stream);
// The same for n_rows rows of row_bytes lying `pitch` bytes apart on the host.
int upload_rows(void *dst, const void *src, int64_t n_rows, size_t row_bytes, size_t pitch,
                hipStream_t stream);

// Brackets the kernel launches issued during its lifetime with HIP events when
// profiling is enabled (mdx_profile_enable); free otherwise.
struct KernelTimer {
  KernelTimer(int slot, hipStream_t stream);
  ~KernelTimer();
  KernelTimer(const KernelTimer &) = delete;
  KernelTimer &operator=(const KernelTimer &) = delete;

 private:
  int slot_;
  hipStream_t stream_;
  hipEvent_t start_ = nullptr;
  hipEvent_t stop_ = nullptr;
};

// In-place all-reduce of device memory on `stream` (asynchronous).
int comm_allreduce_device(mdx_comm_s *comm, void *dptr, int64_t count, int dtype, int op,
                          hipStream_t stream);

inline size_t dtype_size(int dtype) { return dtype == MDX_F64 ? 8 : 4; }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// rint(r / box) in float32, exactly as the IEEE divide would give it: the product
// with the rounded reciprocal is within a few ulp of the true quotient, so its
// nearest integer is the right one unless the quotient lies that close to a
// half-integer; only then the divide is executed.
__device__ __forceinline__ float exact_rint_quotient(float r, float box, float inv_box) {
  const float q = r * inv_box;
  float n = rintf(q);
  const float dist = fabsf(fabsf(q - n) - 0.5f);
  if (__builtin_expect(dist <= fabsf(q) * 0x1p-20f + 0x1p-30f, 0)) {
    n = rintf(r / box);
  }
  return n;
}

}  // namespace mdx
