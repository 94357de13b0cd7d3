// Multi-GPU exchange step of libmdx: one RCCL communicator per process (one process per
// GPU), created from a unique id that the ranks pass around themselves (file, pipe,
// socket: 128 opaque bytes), and an all-reduce on it.  The payloads are histograms and
// correlation sums of 10^4..10^5 bytes, once per calculator run, so the call is
// latency-bound; there is nothing to pipeline.
//
// librccl.so is opened on first use (dlopen), not linked: a single-GPU process never
// loads it, and a machine without it still runs everything else.
#include <dlfcn.h>

#include "mdx_common.h"

using namespace mdx;

namespace {

// The few RCCL declarations used, restated so that the header is not needed at build
// time (values from rccl.h of ROCm 7: ncclSum 0, ncclMax 2, ncclInt64 4, ncclUint64 5,
// ncclFloat64 8, id of 128 bytes).
struct RcclUniqueId {
  char internal[MDX_COMM_ID_BYTES];
};
typedef void *RcclComm;
typedef int (*GetUniqueIdFn)(RcclUniqueId *);
typedef int (*CommInitRankFn)(RcclComm *, int, RcclUniqueId, int);
typedef int (*CommDestroyFn)(RcclComm);
typedef int (*AllReduceFn)(const void *, void *, size_t, int, int, RcclComm, hipStream_t);
typedef const char *(*GetErrorStringFn)(int);

struct RcclApi {
  void *handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllReduceFn all_reduce = nullptr;
  GetErrorStringFn get_error_string = nullptr;
  bool tried = false;
  char why[512] = {0};
};

std::mutex g_api_mutex;
RcclApi g_api;

int load_api() {
  std::lock_guard<std::mutex> lock(g_api_mutex);
  if (g_api.handle != nullptr) return MDX_OK;
  if (g_api.tried) return fail(MDX_ERR_STATE, "%s", g_api.why);
  g_api.tried = true;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h != nullptr) break;
  }
  if (h == nullptr) {
    snprintf(g_api.why, sizeof(g_api.why), "librccl.so could not be opened: %s", dlerror());
    return fail(MDX_ERR_STATE, "%s", g_api.why);
  }
  g_api.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(h, "ncclGetUniqueId"));
  g_api.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(h, "ncclCommInitRank"));
  g_api.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(h, "ncclCommDestroy"));
  g_api.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(h, "ncclAllReduce"));
  g_api.get_error_string = reinterpret_cast<GetErrorStringFn>(dlsym(h, "ncclGetErrorString"));
  if (!g_api.get_unique_id || !g_api.comm_init_rank || !g_api.comm_destroy ||
      !g_api.all_reduce || !g_api.get_error_string) {
    snprintf(g_api.why, sizeof(g_api.why), "librccl.so lacks one of the five entry points used");
    dlclose(h);
    return fail(MDX_ERR_STATE, "%s", g_api.why);
  }
  g_api.handle = h;
  return MDX_OK;
}

#define MDX_RCCL(call)                                                              \
  do {                                                                              \
    int mdx_nccl_rc_ = (call);                                                      \
    if (mdx_nccl_rc_ != 0)                                                          \
      return fail(MDX_ERR_DEVICE, "%s failed: %s", #call,                           \
                  g_api.get_error_string(mdx_nccl_rc_));                            \
  } while (0)

int rccl_dtype(int dtype, int *out, size_t *elem) {
  switch (dtype) {
    case MDX_COMM_U64: *out = 5; *elem = 8; return MDX_OK;
    case MDX_COMM_I64: *out = 4; *elem = 8; return MDX_OK;
    case MDX_COMM_F64: *out = 8; *elem = 8; return MDX_OK;
  }
  return fail(MDX_ERR_INVALID_ARGUMENT, "dtype must be MDX_COMM_U64, _I64 or _F64");
}

int rccl_op(int op, int *out) {
  if (op == MDX_COMM_SUM) { *out = 0; return MDX_OK; }
  if (op == MDX_COMM_MAX) { *out = 2; return MDX_OK; }
  return fail(MDX_ERR_INVALID_ARGUMENT, "op must be MDX_COMM_SUM or MDX_COMM_MAX");
}

}  // namespace

struct mdx_comm_s {
  RcclComm comm = nullptr;
  int rank = 0;
  int size = 1;
  int device = 0;
};

namespace mdx {

int comm_allreduce_device(mdx_comm_s *c, void *dptr, int64_t count, int dtype, int op,
                          hipStream_t stream) {
  MDX_REQUIRE(c != nullptr && c->comm != nullptr, "communicator is null");
  int dev = 0;
  MDX_HIP(hipGetDevice(&dev));
  MDX_REQUIRE(dev == c->device, "communicator belongs to device %d, current device is %d",
              c->device, dev);
  int t = 0, o = 0;
  size_t elem = 0;
  MDX_TRY(rccl_dtype(dtype, &t, &elem));
  MDX_TRY(rccl_op(op, &o));
  if (count == 0) return MDX_OK;
  MDX_RCCL(g_api.all_reduce(dptr, dptr, static_cast<size_t>(count), t, o, c->comm, stream));
  return MDX_OK;
}

}  // namespace mdx

extern "C" {

int mdx_comm_available(void) { return load_api() == MDX_OK ? 1 : 0; }

int mdx_comm_unique_id(void *id, size_t bytes) {
  MDX_REQUIRE(id != nullptr && bytes == MDX_COMM_ID_BYTES, "id must point at %d bytes",
              MDX_COMM_ID_BYTES);
  MDX_TRY(load_api());
  RcclUniqueId uid;
  MDX_RCCL(g_api.get_unique_id(&uid));
  memcpy(id, uid.internal, MDX_COMM_ID_BYTES);
  return MDX_OK;
}

int mdx_comm_create(mdx_comm_t *comm, int rank, int size, const void *id, size_t bytes) {
  MDX_REQUIRE(comm != nullptr, "comm is null");
  *comm = nullptr;
  MDX_REQUIRE(size >= 1 && rank >= 0 && rank < size, "bad rank %d of %d", rank, size);
  MDX_REQUIRE(id != nullptr && bytes == MDX_COMM_ID_BYTES, "id must point at %d bytes",
              MDX_COMM_ID_BYTES);
  MDX_TRY(load_api());
  mdx_comm_s *c = new (std::nothrow) mdx_comm_s;
  if (!c) return fail(MDX_ERR_OUT_OF_MEMORY, "out of host memory");
  c->rank = rank;
  c->size = size;
  hipError_t e = hipGetDevice(&c->device);
  if (e != hipSuccess) {
    delete c;
    return fail(MDX_ERR_DEVICE, "hipGetDevice failed: %s", hipGetErrorString(e));
  }
  RcclUniqueId uid;
  memcpy(uid.internal, id, MDX_COMM_ID_BYTES);
  const int rc = g_api.comm_init_rank(&c->comm, size, uid, rank);
  if (rc != 0) {
    delete c;
    return fail(MDX_ERR_DEVICE, "ncclCommInitRank(rank %d of %d) failed: %s", rank, size,
                g_api.get_error_string(rc));
  }
  *comm = c;
  return MDX_OK;
}

int mdx_comm_destroy(mdx_comm_t comm) {
  if (comm == nullptr) return MDX_OK;
  hipStream_t stream;
  if (get_stream(&stream) == MDX_OK) (void)hipStreamSynchronize(stream);
  if (comm->comm != nullptr && g_api.comm_destroy != nullptr) (void)g_api.comm_destroy(comm->comm);
  delete comm;
  return MDX_OK;
}

int mdx_comm_rank(mdx_comm_t comm, int *rank, int *size) {
  MDX_REQUIRE(comm != nullptr, "comm is null");
  if (rank) *rank = comm->rank;
  if (size) *size = comm->size;
  return MDX_OK;
}

int mdx_comm_allreduce(mdx_comm_t comm, void *data, int location, int64_t count, int dtype,
                       int op) {
  MDX_REQUIRE(comm != nullptr && data != nullptr, "null argument");
  MDX_REQUIRE(count >= 0, "negative count");
  MDX_REQUIRE(location == MDX_HOST || location == MDX_DEVICE, "bad location");
  int t = 0;
  size_t elem = 0;
  MDX_TRY(rccl_dtype(dtype, &t, &elem));
  hipStream_t stream;
  MDX_TRY(get_stream(&stream));
  if (location == MDX_DEVICE) {
    MDX_TRY(comm_allreduce_device(comm, data, count, dtype, op, stream));
    MDX_HIP(hipStreamSynchronize(stream));
    return MDX_OK;
  }
  // host data: through a device block, the collective runs on device memory
  DeviceBuffer staged;
  const size_t bytes = elem * static_cast<size_t>(count);
  MDX_TRY(staged.alloc(bytes));
  MDX_HIP(hipMemcpyAsync(staged.ptr, data, bytes, hipMemcpyHostToDevice, stream));
  MDX_TRY(comm_allreduce_device(comm, staged.ptr, count, dtype, op, stream));
  MDX_HIP(hipMemcpyAsync(data, staged.ptr, bytes, hipMemcpyDeviceToHost, stream));
  MDX_HIP(hipStreamSynchronize(stream));
  return MDX_OK;
}

}  // extern "C"
