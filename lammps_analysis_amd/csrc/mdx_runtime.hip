// Runtime part of libmdx: error text, device selection, the per-device library
// stream, raw device memory and event timing.
#include <atomic>

#include "mdx_common.h"

namespace mdx {

static thread_local char g_error[1024] = {0};

char *error_buffer() { return g_error; }

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
  return code;
}

static constexpr int kMaxDevices = 64;
static hipStream_t g_streams[kMaxDevices] = {nullptr};

int get_stream(hipStream_t *stream) {
  int dev = 0;
  MDX_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices)
    return fail(MDX_ERR_DEVICE, "device index %d out of range", dev);
  if (g_streams[dev] == nullptr) {
    MDX_HIP(hipStreamCreateWithFlags(&g_streams[dev], hipStreamNonBlocking));
  }
  *stream = g_streams[dev];
  return MDX_OK;
}

// ---- device memory cache
namespace {
struct DevicePool {
  std::multimap<size_t, void *> free_blocks;  // capacity -> block
  size_t cached_bytes = 0;
};
std::mutex g_pool_mutex;
DevicePool g_pools[kMaxDevices];
constexpr size_t kPoolCachedLimit = size_t(8) << 30;  // per device
constexpr size_t kPoolBlockLimit = size_t(2) << 30;   // larger blocks are not cached

void trim_device_locked(int dev) {
  DevicePool &pool = g_pools[dev];
  if (pool.free_blocks.empty()) return;
  int current = 0;
  const bool switched = hipGetDevice(&current) == hipSuccess && current != dev &&
                        hipSetDevice(dev) == hipSuccess;
  for (auto &entry : pool.free_blocks) (void)hipFree(entry.second);
  pool.free_blocks.clear();
  pool.cached_bytes = 0;
  if (switched) (void)hipSetDevice(current);
}
}  // namespace

// Calls the library has made to the driver's allocators (device, page-locked host) and
// every request to its block cache, for tests that must show a call path allocates nothing.
static std::atomic<uint64_t> g_driver_allocations{0};
static std::atomic<uint64_t> g_pool_requests{0};
void note_driver_allocation() { g_driver_allocations.fetch_add(1, std::memory_order_relaxed); }

int pool_acquire(void **ptr, size_t bytes, size_t *capacity, int *device) {
  g_pool_requests.fetch_add(1, std::memory_order_relaxed);
  int dev = 0;
  MDX_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices)
    return fail(MDX_ERR_DEVICE, "device index %d out of range", dev);
  // round up so that nearby sizes share blocks: 4 KiB granules up to 1 MiB, 1/8 of
  // the leading power of two above
  size_t want = (bytes + 4095) & ~size_t(4095);
  if (want > (size_t(1) << 20)) {
    size_t granule = size_t(1) << 17;
    while (granule * 16 < want) granule <<= 1;
    want = (want + granule - 1) / granule * granule;
  }
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    DevicePool &pool = g_pools[dev];
    auto it = pool.free_blocks.lower_bound(want);
    if (it != pool.free_blocks.end() && it->first <= want + want / 4) {
      *ptr = it->second;
      *capacity = it->first;
      *device = dev;
      pool.cached_bytes -= it->first;
      pool.free_blocks.erase(it);
      return MDX_OK;
    }
  }
  note_driver_allocation();
  hipError_t e = hipMalloc(ptr, want);
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    {
      std::lock_guard<std::mutex> lock(g_pool_mutex);
      trim_device_locked(dev);
    }
    e = hipMalloc(ptr, want);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *ptr = nullptr;
    return fail(e == hipErrorOutOfMemory ? MDX_ERR_OUT_OF_MEMORY : MDX_ERR_DEVICE,
                "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
  }
  *capacity = want;
  *device = dev;
  return MDX_OK;
}

void pool_release(void *ptr, size_t capacity, int device) {
  if (ptr == nullptr) return;
  if (device >= 0 && device < kMaxDevices && capacity <= kPoolBlockLimit) {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    DevicePool &pool = g_pools[device];
    if (pool.cached_bytes + capacity <= kPoolCachedLimit) {
      pool.free_blocks.emplace(capacity, ptr);
      pool.cached_bytes += capacity;
      return;
    }
  }
  (void)hipFree(ptr);
}

void pool_trim() {
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  for (int dev = 0; dev < kMaxDevices; ++dev) trim_device_locked(dev);
}

static hipStream_t g_copy_streams[kMaxDevices] = {nullptr};

int get_copy_stream(hipStream_t *stream) {
  int dev = 0;
  MDX_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices)
    return fail(MDX_ERR_DEVICE, "device index %d out of range", dev);
  if (g_copy_streams[dev] == nullptr) {
    MDX_HIP(hipStreamCreateWithFlags(&g_copy_streams[dev], hipStreamNonBlocking));
  }
  *stream = g_copy_streams[dev];
  return MDX_OK;
}

// ---- large host-to-device copies
//
// hipMemcpy from pageable memory moves about 30 GB/s on this platform, from page-locked
// memory 57 GB/s.  Large uploads therefore go through two page-locked twins of 64 MiB:
// four host threads fill one twin (58 GB/s) while the previous one is on the wire.  The
// twins are kept per device for the life of the process (released with the cached device
// memory); one upload at a time uses them.
namespace {
constexpr size_t kUploadChunk = size_t(64) << 20;
constexpr size_t kUploadThreshold = size_t(96) << 20;
constexpr int kUploadThreads = 4;
struct UploadTwins {
  void *pinned[2] = {nullptr, nullptr};
  hipEvent_t sent[2] = {nullptr, nullptr};
};
std::mutex g_upload_mutex;
UploadTwins g_upload[kMaxDevices];

void release_upload_twins_locked(int dev) {
  UploadTwins &tw = g_upload[dev];
  for (int b = 0; b < 2; ++b) {
    if (tw.sent[b] != nullptr) (void)hipEventDestroy(tw.sent[b]);
    if (tw.pinned[b] != nullptr) (void)hipHostFree(tw.pinned[b]);
    tw.sent[b] = nullptr;
    tw.pinned[b] = nullptr;
  }
}

bool prepare_upload_twins_locked(int dev) {
  UploadTwins &tw = g_upload[dev];
  if (tw.pinned[1] != nullptr) return true;
  for (int b = 0; b < 2; ++b) {
    note_driver_allocation();
    if (hipHostMalloc(&tw.pinned[b], kUploadChunk, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&tw.sent[b], hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      release_upload_twins_locked(dev);
      return false;
    }
  }
  return true;
}

void parallel_copy(char *dst, const char *src, size_t bytes) {
  const size_t part = (bytes / kUploadThreads + 4095) / 4096 * 4096;
  std::thread workers[kUploadThreads - 1];
  int started = 0;
  for (int t = 1; t < kUploadThreads; ++t) {
    const size_t begin = static_cast<size_t>(t) * part;
    if (begin >= bytes) break;
    const size_t n = std::min(part, bytes - begin);
    workers[started++] = std::thread([=] { memcpy(dst + begin, src + begin, n); });
  }
  memcpy(dst, src, std::min(part, bytes));
  for (int t = 0; t < started; ++t) workers[t].join();
}
}  // namespace

// Copies `bytes` from host to device memory in stream order.  The host range has been
// read when the call returns; the device side completes on `stream`.
int upload(void *dst, const void *src, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return MDX_OK;
  int dev = 0;
  MDX_HIP(hipGetDevice(&dev));
  if (bytes >= kUploadThreshold && dev >= 0 && dev < kMaxDevices) {
    std::lock_guard<std::mutex> lock(g_upload_mutex);
    if (prepare_upload_twins_locked(dev)) {
      UploadTwins &tw = g_upload[dev];
      const char *from = static_cast<const char *>(src);
      char *to = static_cast<char *>(dst);
      int index = 0;
      for (size_t off = 0; off < bytes; off += kUploadChunk, ++index) {
        const int b = index & 1;
        const size_t n = std::min(kUploadChunk, bytes - off);
        if (index >= 2) MDX_HIP(hipEventSynchronize(tw.sent[b]));  // the twin has left
        parallel_copy(static_cast<char *>(tw.pinned[b]), from + off, n);
        MDX_HIP(hipMemcpyAsync(to + off, tw.pinned[b], n, hipMemcpyHostToDevice, stream));
        MDX_HIP(hipEventRecord(tw.sent[b], stream));
      }
      // the twins belong to the next upload once both have left
      MDX_HIP(hipEventSynchronize(tw.sent[0]));
      if (index >= 2) MDX_HIP(hipEventSynchronize(tw.sent[1]));
      return MDX_OK;
    }
  }
  MDX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
  return MDX_OK;
}

// The same for rows that lie `pitch` bytes apart in host memory (a frame range of an
// [atoms][frames][k] array): dst receives them back to back.  Whole rows are gathered into
// the page-locked twins by the copy threads, so the strided source is read once and no
// contiguous host copy is made first.
int upload_rows(void *dst, const void *src, int64_t n_rows, size_t row_bytes, size_t pitch,
                hipStream_t stream) {
  if (n_rows <= 0 || row_bytes == 0) return MDX_OK;
  if (pitch == row_bytes) return upload(dst, src, static_cast<size_t>(n_rows) * row_bytes, stream);
  int dev = 0;
  MDX_HIP(hipGetDevice(&dev));
  const char *from = static_cast<const char *>(src);
  char *to = static_cast<char *>(dst);
  if (row_bytes <= kUploadChunk && dev >= 0 && dev < kMaxDevices) {
    std::lock_guard<std::mutex> lock(g_upload_mutex);
    if (prepare_upload_twins_locked(dev)) {
      UploadTwins &tw = g_upload[dev];
      const int64_t rows_per_chunk = std::max<int64_t>(1, static_cast<int64_t>(kUploadChunk / row_bytes));
      int index = 0;
      for (int64_t r0 = 0; r0 < n_rows; r0 += rows_per_chunk, ++index) {
        const int b = index & 1;
        const int64_t rows = std::min(rows_per_chunk, n_rows - r0);
        if (index >= 2) MDX_HIP(hipEventSynchronize(tw.sent[b]));  // the twin has left
        char *twin = static_cast<char *>(tw.pinned[b]);
        auto part = [=](int64_t a0, int64_t a1) {
          for (int64_t r = a0; r < a1; ++r)
            memcpy(twin + static_cast<size_t>(r) * row_bytes,
                   from + static_cast<size_t>(r0 + r) * pitch, row_bytes);
        };
        const int64_t share = (rows + kUploadThreads - 1) / kUploadThreads;
        std::thread workers[kUploadThreads - 1];
        int started = 0;
        if (static_cast<size_t>(rows) * row_bytes >= (size_t(1) << 20)) {
          for (int t = 1; t < kUploadThreads; ++t) {
            const int64_t a0 = std::min<int64_t>(rows, share * t);
            const int64_t a1 = std::min<int64_t>(rows, share * (t + 1));
            if (a1 > a0) workers[started++] = std::thread(part, a0, a1);
          }
          part(0, std::min<int64_t>(rows, share));
        } else {
          part(0, rows);
        }
        for (int t = 0; t < started; ++t) workers[t].join();
        MDX_HIP(hipMemcpyAsync(to + static_cast<size_t>(r0) * row_bytes, twin,
                               static_cast<size_t>(rows) * row_bytes, hipMemcpyHostToDevice,
                               stream));
        MDX_HIP(hipEventRecord(tw.sent[b], stream));
      }
      MDX_HIP(hipEventSynchronize(tw.sent[0]));
      if (index >= 2) MDX_HIP(hipEventSynchronize(tw.sent[1]));
      return MDX_OK;
    }
  }
  // rows longer than a twin, or no page-locked memory to be had: the driver's pitched copy
  MDX_HIP(hipMemcpy2DAsync(dst, row_bytes, src, pitch, row_bytes, static_cast<size_t>(n_rows),
                           hipMemcpyHostToDevice, stream));
  MDX_HIP(hipStreamSynchronize(stream));
  return MDX_OK;
}

void release_upload_twins() {
  std::lock_guard<std::mutex> lock(g_upload_mutex);
  int current = 0;
  if (hipGetDevice(&current) != hipSuccess) return;
  for (int dev = 0; dev < kMaxDevices; ++dev) {
    if (g_upload[dev].pinned[0] == nullptr && g_upload[dev].pinned[1] == nullptr) continue;
    release_upload_twins_locked(dev);
  }
}

int to_device(const void *src, int location, size_t bytes, DeviceBuffer &staging,
              hipStream_t stream, const void **dptr) {
  if (src == nullptr) return fail(MDX_ERR_INVALID_ARGUMENT, "null data pointer");
  if (location == MDX_DEVICE) {
    *dptr = src;
    return MDX_OK;
  }
  if (location != MDX_HOST)
    return fail(MDX_ERR_INVALID_ARGUMENT, "location must be MDX_HOST or MDX_DEVICE");
  MDX_TRY(staging.alloc(bytes));
  MDX_TRY(upload(staging.ptr, src, bytes, stream));
  *dptr = staging.ptr;
  return MDX_OK;
}

// ---- kernel profiling
static bool g_profiling = false;
struct EventPair {
  hipEvent_t start, stop;
};
static std::vector<EventPair> g_pending[MDX_NUM_SLOTS];
static double g_total_ms[MDX_NUM_SLOTS] = {0};
static int64_t g_launches[MDX_NUM_SLOTS] = {0};

KernelTimer::KernelTimer(int slot, hipStream_t stream) : slot_(slot), stream_(stream) {
  if (!g_profiling || slot < 0 || slot >= MDX_NUM_SLOTS) return;
  if (hipEventCreate(&start_) != hipSuccess) {
    start_ = nullptr;
    return;
  }
  if (hipEventCreate(&stop_) != hipSuccess) {
    (void)hipEventDestroy(start_);
    start_ = stop_ = nullptr;
    return;
  }
  (void)hipEventRecord(start_, stream_);
}

KernelTimer::~KernelTimer() {
  if (start_ == nullptr) return;
  (void)hipEventRecord(stop_, stream_);
  g_pending[slot_].push_back({start_, stop_});
}

static void drain_slot(int slot) {
  for (EventPair &p : g_pending[slot]) {
    float ms = 0.0f;
    if (hipEventSynchronize(p.stop) == hipSuccess &&
        hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
      g_total_ms[slot] += ms;
      g_launches[slot] += 1;
    }
    (void)hipEventDestroy(p.start);
    (void)hipEventDestroy(p.stop);
  }
  g_pending[slot].clear();
}

}  // namespace mdx

using namespace mdx;

extern "C" int mdx_profile_enable(int on) {
  g_profiling = (on != 0);
  return MDX_OK;
}

extern "C" int mdx_profile_reset(void) {
  for (int s = 0; s < MDX_NUM_SLOTS; ++s) {
    drain_slot(s);
    g_total_ms[s] = 0.0;
    g_launches[s] = 0;
  }
  return MDX_OK;
}

extern "C" int mdx_profile_query(int slot, double *total_ms, int64_t *launches) {
  MDX_REQUIRE(slot >= 0 && slot < MDX_NUM_SLOTS, "slot out of range");
  drain_slot(slot);
  if (total_ms) *total_ms = g_total_ms[slot];
  if (launches) *launches = g_launches[slot];
  return MDX_OK;
}

struct mdx_event_s {
  hipEvent_t ev;
};

extern "C" {

int mdx_version(void) { return 100; }

const char *mdx_last_error(void) { return error_buffer(); }

int mdx_device_count(int *count) {
  MDX_REQUIRE(count != nullptr, "count is null");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(MDX_ERR_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  }
  *count = n;
  return MDX_OK;
}

int mdx_set_device(int device) {
  MDX_HIP(hipSetDevice(device));
  return MDX_OK;
}

int mdx_device_info(char *name, size_t name_len, int *compute_units,
                    uint64_t *memory_bytes) {
  int dev = 0;
  MDX_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  MDX_HIP(hipGetDeviceProperties(&prop, dev));
  if (name != nullptr && name_len > 0) {
    snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (memory_bytes) *memory_bytes = static_cast<uint64_t>(prop.totalGlobalMem);
  return MDX_OK;
}

int mdx_synchronize(void) {
  hipStream_t s;
  MDX_TRY(get_stream(&s));
  MDX_HIP(hipStreamSynchronize(s));
  return MDX_OK;
}

int mdx_malloc(void **dptr, size_t bytes) {
  MDX_REQUIRE(dptr != nullptr, "dptr is null");
  *dptr = nullptr;
  note_driver_allocation();
  MDX_HIP(hipMalloc(dptr, bytes == 0 ? 16 : bytes));
  return MDX_OK;
}

int mdx_allocation_counters(uint64_t *driver_allocations, uint64_t *pool_requests,
                            uint64_t *device_free_bytes) {
  MDX_REQUIRE(driver_allocations != nullptr && pool_requests != nullptr &&
                  device_free_bytes != nullptr, "null argument");
  *driver_allocations = g_driver_allocations.load(std::memory_order_relaxed);
  *pool_requests = g_pool_requests.load(std::memory_order_relaxed);
  size_t free_bytes = 0, total_bytes = 0;
  MDX_HIP(hipMemGetInfo(&free_bytes, &total_bytes));
  *device_free_bytes = free_bytes;
  return MDX_OK;
}

int mdx_free(void *dptr) {
  if (dptr == nullptr) return MDX_OK;
  MDX_HIP(hipFree(dptr));
  return MDX_OK;
}

int mdx_release_cached_memory(void) {
  pool_trim();
  release_upload_twins();
  return MDX_OK;
}

int mdx_memcpy_h2d(void *dptr, const void *host, size_t bytes) {
  MDX_REQUIRE(dptr != nullptr && host != nullptr, "null pointer");
  hipStream_t s;
  MDX_TRY(get_stream(&s));
  MDX_TRY(upload(dptr, host, bytes, s));
  MDX_HIP(hipStreamSynchronize(s));
  return MDX_OK;
}

int mdx_memcpy_h2d_rows(void *dptr, const void *host, int64_t n_rows, size_t row_bytes,
                        size_t pitch_bytes) {
  MDX_REQUIRE(dptr != nullptr && host != nullptr, "null pointer");
  MDX_REQUIRE(n_rows >= 0 && pitch_bytes >= row_bytes, "rows must not overlap");
  hipStream_t s;
  MDX_TRY(get_stream(&s));
  MDX_TRY(upload_rows(dptr, host, n_rows, row_bytes, pitch_bytes, s));
  MDX_HIP(hipStreamSynchronize(s));
  return MDX_OK;
}

int mdx_memcpy_d2h(void *host, const void *dptr, size_t bytes) {
  MDX_REQUIRE(dptr != nullptr && host != nullptr, "null pointer");
  hipStream_t s;
  MDX_TRY(get_stream(&s));
  MDX_HIP(hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, s));
  MDX_HIP(hipStreamSynchronize(s));
  return MDX_OK;
}

int mdx_memset(void *dptr, int byte_value, size_t bytes) {
  MDX_REQUIRE(dptr != nullptr, "null pointer");
  hipStream_t s;
  MDX_TRY(get_stream(&s));
  MDX_HIP(hipMemsetAsync(dptr, byte_value, bytes, s));
  return MDX_OK;
}

int mdx_event_create(mdx_event_t *ev) {
  MDX_REQUIRE(ev != nullptr, "ev is null");
  mdx_event_s *e = new (std::nothrow) mdx_event_s;
  if (!e) return fail(MDX_ERR_OUT_OF_MEMORY, "out of host memory");
  hipError_t rc = hipEventCreate(&e->ev);
  if (rc != hipSuccess) {
    delete e;
    return fail(MDX_ERR_DEVICE, "hipEventCreate failed: %s", hipGetErrorString(rc));
  }
  *ev = e;
  return MDX_OK;
}

int mdx_event_record(mdx_event_t ev) {
  MDX_REQUIRE(ev != nullptr, "ev is null");
  hipStream_t s;
  MDX_TRY(get_stream(&s));
  MDX_HIP(hipEventRecord(ev->ev, s));
  return MDX_OK;
}

int mdx_event_elapsed_ms(mdx_event_t start, mdx_event_t stop, float *ms) {
  MDX_REQUIRE(start && stop && ms, "null argument");
  MDX_HIP(hipEventSynchronize(stop->ev));
  MDX_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
  return MDX_OK;
}

int mdx_event_destroy(mdx_event_t ev) {
  if (!ev) return MDX_OK;
  (void)hipEventDestroy(ev->ev);
  delete ev;
  return MDX_OK;
}

}  // extern "C"

// ------------------------------------------------------- synthetic fill (bench)

// 64-bit mix (splitmix64 finaliser) -> 24 random bits -> float in [0,1).
__device__ __forceinline__ float mdx_hash_unit(uint64_t seed, uint64_t i) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return static_cast<float>(z >> 40) * (1.0f / 16777216.0f);
}

__global__ void mdx_fill_uniform_kernel(float *out, uint64_t n, uint64_t seed,
                                        float lo, float hi) {
  uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) out[i] = lo + (hi - lo) * mdx_hash_unit(seed, i);
}

extern "C" int mdx_fill_uniform_f32(float *dptr, uint64_t n, uint64_t seed, float lo,
                                    float hi) {
  MDX_REQUIRE(dptr != nullptr, "null pointer");
  hipStream_t s;
  MDX_TRY(get_stream(&s));
  if (n == 0) return MDX_OK;
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mdx_fill_uniform_kernel, dim3(static_cast<unsigned>(blocks)),
                     dim3(256), 0, s, dptr, n, seed, lo, hi);
  MDX_HIP(hipGetLastError());
  return MDX_OK;
}

// Ornstein-Uhlenbeck (Langevin) velocities, one independent series per (atom, component):
//   v[t+1] = decay * v[t] + kick * xi,  xi standard normal (Box-Muller on the counter
// hash above), v[0] drawn from the stationary distribution kick / sqrt(1 - decay^2).
// Thread = one series, walking the frame axis of the [atoms][frames][k] array.
__global__ __launch_bounds__(256) void mdx_fill_ou_kernel(float *out, int64_t n_series_atoms,
                                                          int64_t n_frames, int k,
                                                          uint64_t seed, float decay,
                                                          float kick) {
  const int64_t series = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (series >= n_series_atoms * k) return;
  const int64_t atom = series / k;
  const int c = static_cast<int>(series - atom * k);
  float *p = out + (atom * n_frames) * k + c;
  const uint64_t base = static_cast<uint64_t>(series) * static_cast<uint64_t>(n_frames + 1) * 2u;
  const float stationary = kick * rsqrtf(fmaxf(1.0f - decay * decay, 1e-12f));
  float v = 0.0f;
  for (int64_t t = -1; t < n_frames - 1; ++t) {
    const uint64_t i = base + static_cast<uint64_t>(t + 1) * 2u;
    const float u1 = fmaxf(mdx_hash_unit(seed, i), 5.9604645e-8f);
    const float u2 = mdx_hash_unit(seed, i + 1);
    const float xi = sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2);
    v = (t < 0) ? stationary * xi : decay * v + kick * xi;
    p[(t + 1) * k] = v;
  }
}

extern "C" int mdx_fill_ou_f32(float *dptr, int64_t n_atoms, int64_t n_frames, int k,
                               uint64_t seed, float decay, float kick) {
  MDX_REQUIRE(dptr != nullptr, "null pointer");
  MDX_REQUIRE(n_atoms >= 0 && n_frames >= 0 && k >= 1, "bad size");
  MDX_REQUIRE(decay >= 0.0f && decay < 1.0f, "decay must lie in [0, 1)");
  hipStream_t s;
  MDX_TRY(get_stream(&s));
  const int64_t n = n_atoms * k;
  if (n == 0 || n_frames == 0) return MDX_OK;
  hipLaunchKernelGGL(mdx_fill_ou_kernel, dim3(static_cast<unsigned>(ceil_div(n, 256))),
                     dim3(256), 0, s, dptr, n_atoms, n_frames, k, seed, decay, kick);
  MDX_HIP(hipGetLastError());
  return MDX_OK;
}
