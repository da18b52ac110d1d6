// Debug facility, never on the product path: a guard-band device allocator + checker for out-of-bounds WRITES of any
// kernel in the process (ours or a library's).
//
// mlgnn/_lib.py installs mlgnn_canary_malloc / _free as torch's device allocator when MLGNN_CANARY=1 (a
// torch.cuda.memory.CUDAPluggableAllocator), so EVERY tensor -- every output and workspace handed to a C-ABI call, and
// everything ATen allocates around them -- sits between two 4 KiB bands of a byte pattern; the rear band starts at the
// first byte past the requested size (no rounding slack).  After every C-ABI call the binding calls
// mlgnn_canary_check(): device synchronise, one kernel compares all bands of all live and recently freed blocks, and
// the first damaged band is reported with its block, side and offset.  Freed blocks are recycled only after a check
// has seen their bands intact.  The C-ABI entry points themselves never allocate (include/mlgnn.h): this file is the
// one exception and exists for the test-suite.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "mlgnn.h"

namespace mlgnn {

constexpr int64_t kGuard = 4096;
constexpr unsigned char kPattern = 0xA5;

struct Block { char* base; int64_t size; int64_t cap; };      // user pointer = base + kGuard; cap = usable bytes of the block

struct Entry { const unsigned char* user; int64_t size; };

__global__ __launch_bounds__(256) void canary_scan_kernel(const Entry* __restrict__ tab, int n, unsigned long long* __restrict__ first_bad) {
  for (int b = blockIdx.x; b < n; b += gridDim.x) {
    const Entry e = tab[b];
    for (int side = 0; side < 2; ++side) {
      const unsigned char* g = side == 0 ? e.user - kGuard : e.user + e.size;
      for (int i = threadIdx.x; i < kGuard; i += 256)
        if (g[i] != kPattern) {
          // key: block index (high), side, offset -> the smallest key is the first damaged byte of the first block
          const unsigned long long key = ((unsigned long long)b << 20) | ((unsigned long long)side << 16) | (unsigned)i;
          atomicMin(first_bad, key);
        }
    }
  }
}

struct State {
  std::mutex mu;
  std::unordered_map<void*, Block> live;             // user pointer -> block
  std::vector<Block> pending;                        // freed since the last check
  std::multimap<int64_t, Block> cache;               // checked free blocks by capacity
  Entry* dev_tab = nullptr;
  int64_t dev_tab_cap = 0;
  unsigned long long* dev_flag = nullptr;
  int64_t n_alloc = 0, n_reuse = 0, n_checks = 0;
};

static State& st() { static State* s = new State(); return *s; }       // (leaked on purpose: outlives torch's teardown)

static void release_cache_locked(State& s) {
  for (auto& kv : s.cache) (void)hipFree(kv.second.base);
  s.cache.clear();
}

static int64_t check_locked(State& s, char* msg, int64_t msg_len) {
  if (msg && msg_len > 0) msg[0] = 0;
  if (hipDeviceSynchronize() != hipSuccess) {
    if (msg) snprintf(msg, (size_t)msg_len, "hipDeviceSynchronize failed: %s", hipGetErrorString(hipGetLastError()));
    return -1;
  }
  std::vector<Entry> tab;
  std::vector<Block> blocks;
  tab.reserve(s.live.size() + s.pending.size());
  for (auto& kv : s.live) { tab.push_back({(const unsigned char*)kv.first, kv.second.size}); blocks.push_back(kv.second); }
  for (auto& b : s.pending) { tab.push_back({(const unsigned char*)b.base + kGuard, b.size}); blocks.push_back(b); }
  ++s.n_checks;
  int64_t bad = 0;
  if (!tab.empty()) {
    if ((int64_t)tab.size() > s.dev_tab_cap) {
      if (s.dev_tab) (void)hipFree(s.dev_tab);
      s.dev_tab_cap = (int64_t)tab.size() * 2 + 1024;
      if (hipMalloc((void**)&s.dev_tab, (size_t)s.dev_tab_cap * sizeof(Entry)) != hipSuccess) return -1;
    }
    if (!s.dev_flag && hipMalloc((void**)&s.dev_flag, 8) != hipSuccess) return -1;
    unsigned long long none = ~0ull, got = ~0ull;
    (void)hipMemcpy(s.dev_tab, tab.data(), tab.size() * sizeof(Entry), hipMemcpyHostToDevice);
    (void)hipMemcpy(s.dev_flag, &none, 8, hipMemcpyHostToDevice);
    int grid = (int)tab.size() < 1024 ? (int)tab.size() : 1024;
    hipLaunchKernelGGL(canary_scan_kernel, dim3(grid), dim3(256), 0, 0, s.dev_tab, (int)tab.size(), s.dev_flag);
    (void)hipMemcpy(&got, s.dev_flag, 8, hipMemcpyDeviceToHost);
    if (got != none) {
      bad = 1;
      const int64_t b = (int64_t)(got >> 20);
      const int side = (int)((got >> 16) & 1);
      const int off = (int)(got & 0xffff);
      if (msg)
        snprintf(msg, (size_t)msg_len,
                 "guard band damaged: a write %d bytes %s the %lld-byte allocation at %p (%s)",
                 side ? off : (int)(kGuard - off), side ? "past the end of" : "before the start of",
                 (long long)tab[b].size, (const void*)tab[b].user, b < (int64_t)s.live.size() ? "live" : "already freed");
      // repair the pattern so that one overrun is reported once
      const Block& blk = blocks[b];
      (void)hipMemset(blk.base, kPattern, kGuard);
      (void)hipMemset(blk.base + kGuard + blk.size, kPattern, kGuard);
    }
  }
  for (auto& b : s.pending) s.cache.emplace(b.cap, b);
  s.pending.clear();
  return bad;
}

}  // namespace mlgnn

using namespace mlgnn;

extern "C" void* mlgnn_canary_malloc(int64_t size, int device, void* stream) {
  (void)stream;
  State& s = st();
  std::lock_guard<std::mutex> lock(s.mu);
  (void)hipSetDevice(device);
  if (size < 0) return nullptr;
  const int64_t cap = (size + 511) & ~int64_t(511);
  Block b{nullptr, size, cap};
  auto it = s.cache.lower_bound(cap);
  if (it != s.cache.end() && it->first <= cap + cap / 4 + 4096) {       // a checked free block of about this size
    b = it->second;
    b.size = size;
    s.cache.erase(it);
    ++s.n_reuse;
  } else {
    hipError_t rc = hipMalloc((void**)&b.base, (size_t)(cap + 2 * kGuard));
    if (rc != hipSuccess) {                                             // recycle everything and try once more
      (void)hipGetLastError();
      check_locked(s, nullptr, 0);
      release_cache_locked(s);
      rc = hipMalloc((void**)&b.base, (size_t)(cap + 2 * kGuard));
      if (rc != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    ++s.n_alloc;
    (void)hipMemset(b.base, kPattern, kGuard);
  }
  // the rear band starts at the first byte past the REQUESTED size
  (void)hipMemset(b.base + kGuard + size, kPattern, kGuard);
  void* user = b.base + kGuard;
  s.live[user] = b;
  return user;
}

extern "C" void mlgnn_canary_free(void* ptr, int64_t size, int device, void* stream) {
  (void)size; (void)device; (void)stream;
  if (!ptr) return;
  State& s = st();
  std::lock_guard<std::mutex> lock(s.mu);
  auto it = s.live.find(ptr);
  if (it == s.live.end()) return;
  s.pending.push_back(it->second);                   // kernels that use it may still be in flight: recycled after a check
  s.live.erase(it);
}

extern "C" int64_t mlgnn_canary_check(char* message, int64_t message_bytes) {
  State& s = st();
  std::lock_guard<std::mutex> lock(s.mu);
  return check_locked(s, message, message_bytes);
}

extern "C" int64_t mlgnn_canary_stats(int64_t* out4) {
  State& s = st();
  std::lock_guard<std::mutex> lock(s.mu);
  if (out4) { out4[0] = (int64_t)s.live.size(); out4[1] = s.n_alloc; out4[2] = s.n_reuse; out4[3] = s.n_checks; }
  return (int64_t)s.live.size();
}
