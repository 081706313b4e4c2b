// C ABI plumbing: error reporting, layout helpers.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "sd_common.h"

namespace sda {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return -2;
  }
  return 0;
}

static int current_device() {
  int dev = 0;
  return hipGetDevice(&dev) == hipSuccess ? dev : 0;
}

bool first_use_on_device(unsigned long long& done) {
  const unsigned long long bit = 1ull << (current_device() & 63);
  if (done & bit) return false;
  done |= bit;
  return true;
}

static thread_local int g_cu_limit = 0;

int launch_cus() {
  static int cus[64] = {0};
  const int dev = current_device() & 63;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  return (g_cu_limit > 0 && g_cu_limit < cus[dev]) ? g_cu_limit : cus[dev];
}

}  // namespace sda

extern "C" int sda_set_cu_limit(int cus) {
  const int old = sda::g_cu_limit;
  sda::g_cu_limit = cus > 0 ? cus : 0;
  return old;
}
extern "C" int sda_stream_create_cumask(const uint32_t* mask, int nwords, void** stream) {
  if (!mask || nwords < 1 || !stream) { sda::set_error("stream_create_cumask: bad arguments"); return -1; }
  hipStream_t st = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, mask);
  if (e != hipSuccess) { sda::set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e)); return -2; }
  *stream = (void*)st;
  return 0;
}
extern "C" int sda_stream_create_priority(int priority, void** stream) {
  if (!stream) { sda::set_error("stream_create_priority: bad arguments"); return -1; }
  int least = 0, greatest = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (e != hipSuccess) { sda::set_error("hipDeviceGetStreamPriorityRange: %s", hipGetErrorString(e)); return -2; }
  if (priority > least || priority < greatest) {
    sda::set_error("stream_create_priority: %d outside the device's range [%d (high) .. %d (low)]", priority, greatest, least);
    return -1;
  }
  hipStream_t st = nullptr;
  e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, priority);
  if (e != hipSuccess) { sda::set_error("hipStreamCreateWithPriority: %s", hipGetErrorString(e)); return -2; }
  *stream = (void*)st;
  return 0;
}
extern "C" int sda_stream_destroy(void* stream) {
  if (!stream) return 0;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) { sda::set_error("hipStreamDestroy: %s", hipGetErrorString(e)); return -2; }
  return 0;
}

extern "C" int sda_abi_version(void) { return SDA_ABI_VERSION; }
extern "C" const char* sda_last_error(void) { return sda::g_err; }
extern "C" long sda_rows_alloc(int B, int T) { return sda::rows_alloc(B, T); }
extern "C" int sda_pad_channels(int C) { return (C + SDA_CH_ALIGN - 1) / SDA_CH_ALIGN * SDA_CH_ALIGN; }
extern "C" int sda_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// Small host -> device tables travel INSIDE the kernel-argument block of a tiny kernel (up to 3.75 KB per
// launch) instead of through hipMemcpyAsync: a copy from pageable memory makes the host wait until the stream
// reaches it (a per-step synchronisation), a kernarg payload is captured at launch time and is fully asynchronous.
namespace sda {
constexpr int UPLOAD_WORDS = 960;
struct UploadPayload { uint32_t w[UPLOAD_WORDS]; };
__global__ void upload_words_kernel(uint32_t* __restrict__ dst, const UploadPayload p, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = p.w[i];
}
}  // namespace sda

extern "C" int sda_upload_words(void* dst, const void* src_host, long nwords, void* stream) {
  if (!dst || !src_host || nwords < 1) { sda::set_error("upload_words: bad arguments"); return -1; }
  const uint32_t* src = reinterpret_cast<const uint32_t*>(src_host);
  uint32_t* d = reinterpret_cast<uint32_t*>(dst);
  for (long off = 0; off < nwords; off += sda::UPLOAD_WORDS) {
    sda::UploadPayload p;
    const int n = (int)((nwords - off) < sda::UPLOAD_WORDS ? (nwords - off) : sda::UPLOAD_WORDS);
    memcpy(p.w, src + off, (size_t)n * 4);
    hipLaunchKernelGGL(sda::upload_words_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d + off, p, n);
  }
  return sda::check_launch("upload_words");
}
