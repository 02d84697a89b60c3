// runtime.hip -- error state, device discovery, plain memory helpers and the shared exclusive scan.
#include <dlfcn.h>

#include <cstdlib>
#include <initializer_list>

#include "mhip_internal.hpp"

namespace mhip {

std::string& last_error_storage() {
  thread_local std::string s;
  return s;
}

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error_storage() = buf;
  return code;
}

// ------------------------------------------------------------------------------------------------------------------
// exclusive scan (int32): tiles of 2048 = 256 threads x 8 items.
//   pass 1: per-tile sums            pass 2: one workgroup scans the tile sums      pass 3: per-tile scan + offset
// ------------------------------------------------------------------------------------------------------------------
constexpr int kScanItems = 8;
constexpr int kScanTile = kBlock * kScanItems;

__device__ inline int wave_inclusive_scan_i32(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}
// inclusive scan of one int per thread over the workgroup; returns inclusive value, *total = block sum
__device__ inline int block_inclusive_scan_i32(int v, int* total, int* scratch /*kBlock/64 + 1*/) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int inc = wave_inclusive_scan_i32(v);
  if (lane == 63) scratch[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < nw; ++i) {
    const int s = scratch[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return inc + base;
}

__global__ void __launch_bounds__(kBlock) scan_tile_sums(const int32_t* __restrict__ in, int32_t* __restrict__ sums,
                                                        size_t n) {
  __shared__ int scratch[kBlock / 64 + 1];
  const size_t base = static_cast<size_t>(blockIdx.x) * kScanTile;
  int acc = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const size_t i = base + static_cast<size_t>(k) * kBlock + threadIdx.x;
    if (i < n) acc += in[i];
  }
  int total;
  (void)block_inclusive_scan_i32(acc, &total, scratch);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kBlock) scan_sums(int32_t* __restrict__ sums, size_t ntiles) {
  __shared__ int scratch[kBlock / 64 + 1];
  int carry = 0;
  for (size_t base = 0; base < ntiles; base += kBlock) {
    const size_t i = base + threadIdx.x;
    const int v = (i < ntiles) ? sums[i] : 0;
    int total;
    const int inc = block_inclusive_scan_i32(v, &total, scratch);
    if (i < ntiles) sums[i] = carry + inc - v;
    carry += total;
  }
}

__global__ void __launch_bounds__(kBlock) scan_apply(const int32_t* __restrict__ in, const int32_t* __restrict__ sums,
                                                    int32_t* __restrict__ out, size_t n) {
  __shared__ int scratch[kBlock / 64 + 1];
  const size_t base = static_cast<size_t>(blockIdx.x) * kScanTile + static_cast<size_t>(threadIdx.x) * kScanItems;
  int v[kScanItems];
  int acc = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (base + k < n) ? in[base + k] : 0;
    acc += v[k];
  }
  int total;
  const int inc = block_inclusive_scan_i32(acc, &total, scratch);
  int run = sums[blockIdx.x] + inc - acc;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
  // out[n] = grand total, written by the thread that owns element n-1
  if (n > 0 && base <= n - 1 && n - 1 < base + kScanItems) out[n] = run;
}

__global__ void scan_empty(int32_t* out) { out[0] = 0; }

size_t scan_workspace_bytes(size_t n) { return ((n + kScanTile - 1) / kScanTile + 1) * sizeof(int32_t); }

// NOTE: pass 1 sums a tile with a strided element->thread map, pass 3 scans it with a blocked map; both see the
// same tile, so the tile totals agree.
int exclusive_scan_i32(const int32_t* in, int32_t* out, size_t n, void* workspace, hipStream_t stream) {
  if (n == 0) {
    scan_empty<<<1, 1, 0, stream>>>(out);
    MHIP_LAUNCH_CHECK();
    return MHIP_SUCCESS;
  }
  const size_t ntiles = (n + kScanTile - 1) / kScanTile;
  MHIP_REQUIRE(ntiles < (1u << 31), MHIP_ERR_RUNTIME, "scan too large");
  int32_t* sums = static_cast<int32_t*>(workspace);
  scan_tile_sums<<<static_cast<unsigned>(ntiles), kBlock, 0, stream>>>(in, sums, n);
  MHIP_LAUNCH_CHECK();
  scan_sums<<<1, kBlock, 0, stream>>>(sums, ntiles);
  MHIP_LAUNCH_CHECK();
  scan_apply<<<static_cast<unsigned>(ntiles), kBlock, 0, stream>>>(in, sums, out, n);
  MHIP_LAUNCH_CHECK();
  return MHIP_SUCCESS;
}

struct TraceState {
  int enabled = -1;  // -1: not decided yet (MHIP_TRACE)
  void* lib = nullptr;
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  bool tried = false;
};
TraceState& trace_state() {
  static TraceState t;
  return t;
}
TraceRange::TraceRange(const char* name) : pushed(false) {
  TraceState& t = trace_state();
  if (t.enabled < 0) {
    const char* e = getenv("MHIP_TRACE");
    t.enabled = (e && atoi(e)) ? 1 : 0;
  }
  if (!t.enabled) return;
  if (!t.tried) {
    t.tried = true;
    // rocprofv3 listens to the rocprofiler-sdk flavour of roctx; the roctracer one is the fallback
    for (const char* name : {"librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                             "libroctx64.so", "/opt/rocm/lib/libroctx64.so"}) {
      t.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (t.lib) break;
    }
    if (t.lib) {
      t.push = reinterpret_cast<int (*)(const char*)>(dlsym(t.lib, "roctxRangePushA"));
      t.pop = reinterpret_cast<int (*)()>(dlsym(t.lib, "roctxRangePop"));
    }
  }
  if (t.push && t.pop) {
    t.push(name);
    pushed = true;
  }
}
TraceRange::~TraceRange() {
  if (pushed) trace_state().pop();
}

}  // namespace mhip

using namespace mhip;

extern "C" {

const char* mhip_last_error(void) { return last_error_storage().c_str(); }
int mhip_set_tracing(int enable) {
  trace_state().enabled = enable ? 1 : 0;
  return MHIP_SUCCESS;
}
int mhip_version(void) { return 100; }

int mhip_device_info(int* device_count, char* arch_name, size_t arch_name_len) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    if (device_count) *device_count = 0;
    return fail(MHIP_ERR_NO_DEVICE, "no HIP device visible (%s); libmundy_hip has no CPU fallback",
                e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  }
  if (device_count) *device_count = n;
  if (arch_name && arch_name_len > 0) {
    int dev = 0;
    MHIP_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    MHIP_HIP(hipGetDeviceProperties(&prop, dev));
    strncpy(arch_name, prop.gcnArchName, arch_name_len - 1);
    arch_name[arch_name_len - 1] = 0;
  }
  return MHIP_SUCCESS;
}

int mhip_malloc(void** ptr, size_t bytes) {
  MHIP_REQUIRE(ptr != nullptr, MHIP_ERR_INVALID_ARGUMENT, "mhip_malloc: ptr is null");
  MHIP_HIP(hipMalloc(ptr, bytes ? bytes : 8));
  return MHIP_SUCCESS;
}
int mhip_free(void* ptr) {
  if (ptr) MHIP_HIP(hipFree(ptr));
  return MHIP_SUCCESS;
}
int mhip_memcpy_h2d(void* dst, const void* src_host, size_t bytes, mhip_stream_t stream) {
  if (bytes == 0) return MHIP_SUCCESS;
  MHIP_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
  MHIP_HIP(hipStreamSynchronize(as_stream(stream)));  // pageable source: safe to reuse on return
  return MHIP_SUCCESS;
}
int mhip_memcpy_d2h(void* dst_host, const void* src, size_t bytes, mhip_stream_t stream) {
  if (bytes == 0) return MHIP_SUCCESS;
  MHIP_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
  MHIP_HIP(hipStreamSynchronize(as_stream(stream)));
  return MHIP_SUCCESS;
}
int mhip_stream_synchronize(mhip_stream_t stream) {
  MHIP_HIP(hipStreamSynchronize(as_stream(stream)));
  return MHIP_SUCCESS;
}

}  // extern "C"
