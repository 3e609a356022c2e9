// abi_pools.inl — part of icp_abi.hip (one translation unit; included there, in order).
// pools and caches: streams, pinned blocks, device buffers of destroyed objects; live-context and eigen-stream registries
namespace {

// developer aid (ICP_HOST_TIMING=1): where the host side of icp_chain_step spends its time, printed at context destruction
struct HostTiming {
  bool on = std::getenv("ICP_HOST_TIMING") != nullptr;
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long n = 0, n_first = 0;
  std::chrono::steady_clock::time_point last, exit_t;
  bool have_exit = false;
  void start() { if (on) { last = std::chrono::steady_clock::now(); if (have_exit) acc[7] += us(exit_t, last); } }
  void mark(int k) { if (on) { auto t = std::chrono::steady_clock::now(); acc[k] += us(last, t); last = t; } }
  void mark_wait(bool first_use) { if (on) { auto t = std::chrono::steady_clock::now(); acc[3] += us(last, t); if (first_use) { acc[5] += us(last, t); ++n_first; } last = t; } }
  void end() { if (on) { exit_t = last; have_exit = true; ++n; } }
  static double us(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::micro>(b - a).count();
  }
  void report() {
    if (!on || !n) return;
    std::fprintf(stderr, "[icp host timing] steps %ld | us/step: prepare %.1f  launch K1-K5 %.1f  speculation %.1f  wait %.1f  bookkeeping %.1f  caller %.1f | steps drawing from a new basis %ld: wait %.1f, others: wait %.1f\n",
                 n, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[7] / n, n_first, n_first ? acc[5] / n_first : 0.0,
                 n > n_first ? (acc[3] - acc[5]) / (n - n_first) : 0.0);
  }
};
HostTiming g_host_timing;

struct BatchTiming {  // ICP_HOST_TIMING: where a batched step's host time goes (reported with the above)
  bool on = std::getenv("ICP_HOST_TIMING") != nullptr;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  long calls = 0, chains = 0, stepped_alone = 0;
  std::chrono::steady_clock::time_point last;
  void start() { if (on) last = std::chrono::steady_clock::now(); }
  void mark(int k) { if (on) { auto t = std::chrono::steady_clock::now(); acc[k] += HostTiming::us(last, t); last = t; } }
  void report() {
    if (!on || !calls) return;
    std::fprintf(stderr, "[icp batch timing] calls %ld, %.1f chains each (%ld chain steps taken one by one) | us/call: decompositions %.1f  events %.1f  prepare %.1f  launch %.1f  wait for first chain %.1f  record %.1f\n",
                 calls, (double)chains / calls, stepped_alone, acc[4] / calls, acc[5] / calls, acc[0] / calls, acc[1] / calls, acc[2] / calls, acc[3] / calls);
    calls = 0;
  }
};
BatchTiming g_batch_timing;

// Streams and pinned blocks of destroyed contexts, proposals and evaluators are kept for the next ones.  A batch registration makes
// its contexts and chains anew for every job (and chains for every target): hipStreamCreate* takes 3.3 ms, hipStreamDestroy 2.3 ms,
// hipHostFree 0.2 ms — 4 + 8 streams and a dozen pinned blocks per context, a third of the wall time of a 10 targets x 10 chains x 50
// steps job (rocprofv3 --hip-runtime-trace, tools/r4_setup_trace.sh).  Streams are kept per device and priority class (a stream keeps
// the hardware queue it was created on), pinned blocks by size (handed out zeroed); icp_release_cached_models() empties both,
// ICP_NO_POOL=1 switches the pools off.
struct ResourcePool {
  std::mutex mu;
  static constexpr int kDevices = 16, kStreamsPerClass = 96;
  std::vector<hipStream_t> streams[kDevices][2];      // [device][0 = default priority, 1 = greatest]
  std::map<hipStream_t, int> stream_class;            // every pooled or handed-out stream: device * 2 + class
  std::multimap<std::pair<int, size_t>, void*> pinned;        // free blocks by (device they were pinned under, size)
  std::map<void*, std::pair<int, size_t>> pinned_size;        // every block of the pool, handed out or free
  size_t pinned_free_bytes = 0;
  static constexpr size_t kPinnedCap = (size_t)64 << 20;
  bool on = std::getenv("ICP_NO_POOL") == nullptr;
};
ResourcePool g_pool;

hipStream_t take_stream(int device, bool greatest, int priority) {
  if (device >= 0 && device < ResourcePool::kDevices) {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    auto& v = g_pool.streams[device][greatest ? 1 : 0];
    if (g_pool.on && !v.empty()) {
      hipStream_t s = v.back();
      v.pop_back();
      return s;
    }
  }
  hipStream_t s = nullptr;
  HIP_OK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority));
  if (device >= 0 && device < ResourcePool::kDevices) {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    g_pool.stream_class[s] = device * 2 + (greatest ? 1 : 0);
  }
  return s;
}
// (the caller has synchronised with the stream's work or does not care: the stream is synchronised here)
void give_stream(hipStream_t s) {
  if (!s) return;
  (void)hipStreamSynchronize(s);
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    auto it = g_pool.stream_class.find(s);
    if (g_pool.on && it != g_pool.stream_class.end()) {
      auto& v = g_pool.streams[it->second / 2][it->second & 1];
      if ((int)v.size() < ResourcePool::kStreamsPerClass) { v.push_back(s); return; }
    }
    if (it != g_pool.stream_class.end()) g_pool.stream_class.erase(it);
  }
  (void)hipStreamDestroy(s);
}
void pinned_alloc(void** out, size_t bytes) {
  const size_t size = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
  int dev = -1;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    auto it = g_pool.pinned.find({dev, size});
    if (g_pool.on && it != g_pool.pinned.end()) {
      *out = it->second;
      g_pool.pinned.erase(it);
      g_pool.pinned_free_bytes -= size;
      std::memset(*out, 0, size);
      return;
    }
  }
  HIP_OK(hipHostMalloc(out, size, hipHostMallocDefault));
  std::lock_guard<std::mutex> lk(g_pool.mu);
  g_pool.pinned_size[*out] = {dev, size};
}
struct DevicePool {
  std::mutex mu;
  std::map<std::pair<int, size_t>, std::vector<void*>> free;  // (device, bytes) -> blocks
  std::map<void*, int> owner;                                 // every block handed out or kept: the device it was allocated on
  size_t free_bytes = 0;
  // what the pool may keep: 6 GiB, at most an eighth of the device's memory (ICP_POOL_CAP_MB overrides); blocks above 64 MiB are never kept
  static constexpr size_t kMaxBlock = (size_t)64 << 20;
  size_t cap = 0;
  size_t capacity() {
    if (cap) return cap;
    cap = (size_t)6 << 30;
    if (const char* e = std::getenv("ICP_POOL_CAP_MB")) cap = std::max<size_t>(1, (size_t)std::atoll(e)) << 20;
    else {
      size_t fr = 0, total = 0;
      if (hipMemGetInfo(&fr, &total) == hipSuccess && total / 8 < cap) cap = std::max<size_t>(total / 8, (size_t)64 << 20);
    }
    return cap;
  }
};
DevicePool g_dpool;
thread_local int tl_quiesce_depth = 0;

void pinned_free(void* p) {
  if (!p) return;
  {
    std::unique_lock<std::mutex> lk(g_pool.mu);
    auto it = g_pool.pinned_size.find(p);
    if (g_pool.on && it != g_pool.pinned_size.end() && g_pool.pinned_free_bytes + it->second.second <= ResourcePool::kPinnedCap) {
      // (hipHostFree waits for the device's work; a block that goes back to the pool waits the same way: nothing still writes to it.
      // Inside a DeviceQuiesce scope that wait has happened; otherwise it happens here, on the device the block was pinned under and
      // WITHOUT the pool's lock — another host thread's take_stream / pinned_alloc must not wait for this thread's device)
      const std::pair<int, size_t> key = it->second;
      if (tl_quiesce_depth == 0) {
        lk.unlock();
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (key.first >= 0 && key.first != cur) (void)hipSetDevice(key.first);
        (void)hipDeviceSynchronize();
        if (key.first >= 0 && key.first != cur && cur >= 0) (void)hipSetDevice(cur);
        lk.lock();
      }
      if (g_pool.pinned_free_bytes + key.second <= ResourcePool::kPinnedCap) {
        g_pool.pinned.emplace(key, p);
        g_pool.pinned_free_bytes += key.second;
        return;
      }
      it = g_pool.pinned_size.find(p);
    }
    if (it != g_pool.pinned_size.end()) g_pool.pinned_size.erase(it);
  }
  (void)hipHostFree(p);
}

DeviceQuiesce::DeviceQuiesce() {
  if (tl_quiesce_depth++ == 0 && g_pool.on) (void)hipDeviceSynchronize();
}
DeviceQuiesce::~DeviceQuiesce() { --tl_quiesce_depth; }
// frees every block the device pool keeps (all devices); -> bytes released
size_t drain_device_pool() {
  std::vector<void*> blocks;
  size_t bytes = 0;
  {
    std::lock_guard<std::mutex> lk(g_dpool.mu);
    for (auto& kv : g_dpool.free)
      for (void* b : kv.second) { blocks.push_back(b); g_dpool.owner.erase(b); }
    g_dpool.free.clear();
    bytes = g_dpool.free_bytes;
    g_dpool.free_bytes = 0;
  }
  for (void* b : blocks) (void)hipFree(b);
  return bytes;
}
void* device_alloc(size_t bytes) {
  int dev = 0;
  const bool have_dev = hipGetDevice(&dev) == hipSuccess;
  if (g_pool.on && bytes <= DevicePool::kMaxBlock && have_dev) {
    std::lock_guard<std::mutex> lk(g_dpool.mu);
    auto it = g_dpool.free.find({dev, bytes});
    if (it != g_dpool.free.end() && !it->second.empty()) {
      void* p = it->second.back();
      it->second.pop_back();
      g_dpool.free_bytes -= bytes;
      return p;
    }
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  // test hook (tests/test_gpu_edges.py): the n-th allocation of the process "fails" as if the device were full while the pool holds blocks
  static const long fail_at = dev_env("ICP_TEST_FAIL_MALLOC_AT") ? std::atol(dev_env("ICP_TEST_FAIL_MALLOC_AT")) : 0;
  static std::atomic<long> n_malloc{0};
  const bool forced = fail_at > 0 && e == hipSuccess && ++n_malloc == fail_at;
  if (forced) { (void)hipFree(p); p = nullptr; e = hipErrorOutOfMemory; }
  if (e != hipSuccess) {
    // Blocks are kept by exact size: after a change of model, K, rank or scratch size the kept ones fit nothing and only take the
    // room this allocation needs — they are given back to the runtime, and the allocation is tried once more
    (void)hipGetLastError();
    const size_t drained = drain_device_pool();
    if (forced) std::fprintf(stderr, "[icp test hook] hipMalloc #%ld failed on purpose; the pool gave back %zu bytes\n", fail_at, drained);
    if (drained > 0) e = hipMalloc(&p, bytes);
    if (e != hipSuccess) fail(ICP_ERR_DEVICE, std::string("hipMalloc(") + std::to_string(bytes) + " bytes): " + hipGetErrorString(e));
  }
  if (g_pool.on && have_dev) {
    std::lock_guard<std::mutex> lk(g_dpool.mu);
    g_dpool.owner[p] = dev;  // (the device that owns the block: where it is filed when it comes back, whatever device is current then)
  }
  return p;
}
void device_free(void* p, size_t bytes) {
  if (!p) return;
  if (g_pool.on) {
    int dev = -1;
    {
      std::lock_guard<std::mutex> lk(g_dpool.mu);
      auto it = g_dpool.owner.find(p);
      if (it != g_dpool.owner.end()) dev = it->second;
    }
    if (dev >= 0 && bytes <= DevicePool::kMaxBlock) {
      // (outside a DeviceQuiesce scope — a buffer that grows in the middle of a run — the owning device is waited for here, as hipFree would)
      if (tl_quiesce_depth == 0) {
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != dev) (void)hipSetDevice(dev);
        (void)hipDeviceSynchronize();
        if (cur != dev && cur >= 0) (void)hipSetDevice(cur);
      }
      std::lock_guard<std::mutex> lk(g_dpool.mu);
      if (g_dpool.free_bytes + bytes <= g_dpool.capacity()) {
        g_dpool.free[{dev, bytes}].push_back(p);
        g_dpool.free_bytes += bytes;
        return;
      }
    }
    std::lock_guard<std::mutex> lk(g_dpool.mu);
    g_dpool.owner.erase(p);
  }
  (void)hipFree(p);
}

// A host that is about to make MANY contexts announces it (icp_ctx_expect): their streams are made AHEAD, by a helper thread, while the
// first context's one-off host work (the model's Gram matrix and factorisations: ≈ 0.5 s at the face model's size) keeps the calling
// thread busy: hipStreamCreateWithPriority takes 2.4 ms, three quarters of what a further context costs.  Streams of the default
// priority class (what every context after the first takes), straight into the pool.  Only on that explicit hint (round 6: until then
// the first keyed context of every model did it — also for the one-chain-per-GPU layout, 24 streams nobody would use).
struct StreamPrewarm {
  std::mutex mu;
  std::thread worker;
  std::atomic<bool> stop{false};
  void join() {
    std::lock_guard<std::mutex> lk(mu);
    if (worker.joinable()) worker.join();
  }
  // (process exit: the worker is told to stop between two streams and waited for, so that it is not inside the runtime afterwards)
  ~StreamPrewarm() { stop.store(true); if (worker.joinable()) worker.join(); }
};
StreamPrewarm g_prewarm;
void prewarm_streams(int device, int n) {
  if (!g_pool.on || device < 0 || device >= ResourcePool::kDevices || n <= 0) return;
  static const bool off = std::getenv("ICP_NO_STREAM_PREWARM") != nullptr;  // (operational switch)
  if (off) return;
  std::lock_guard<std::mutex> lk(g_prewarm.mu);
  if (g_prewarm.worker.joinable()) g_prewarm.worker.join();
  g_prewarm.worker = std::thread([device, n] {
    if (hipSetDevice(device) != hipSuccess) return;
    for (int i = 0; i < n && !g_prewarm.stop.load(std::memory_order_relaxed); ++i) {
      {  // (never more than the hint asks for, counting what the pool already holds)
        std::lock_guard<std::mutex> plk(g_pool.mu);
        if ((int)g_pool.streams[device][0].size() >= std::min(n, (int)ResourcePool::kStreamsPerClass)) return;
      }
      hipStream_t s = nullptr;
      if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0) != hipSuccess) return;
      std::lock_guard<std::mutex> plk(g_pool.mu);
      auto& v = g_pool.streams[device][0];
      if ((int)v.size() >= ResourcePool::kStreamsPerClass) { (void)hipStreamDestroy(s); return; }
      g_pool.stream_class[s] = device * 2;
      v.push_back(s);
    }
  });
}

void drain_pools() {
  g_prewarm.join();
  (void)drain_device_pool();
  std::vector<hipStream_t> ss;
  std::vector<void*> blocks;
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    for (auto& dev : g_pool.streams)
      for (auto& v : dev) {
        for (hipStream_t s : v) { ss.push_back(s); g_pool.stream_class.erase(s); }
        v.clear();
      }
    for (auto& kv : g_pool.pinned) { blocks.push_back(kv.second); g_pool.pinned_size.erase(kv.second); }
    g_pool.pinned.clear();
    g_pool.pinned_free_bytes = 0;
  }
  for (hipStream_t s : ss) (void)hipStreamDestroy(s);
  for (void* b : blocks) (void)hipHostFree(b);
}

// Contexts alive in this process.  The speculative decompositions of icp_chain_step keep a few workgroups waiting on the
// device and put three streams per context to work; the runtime multiplexes streams onto four hardware queues, and beyond
// two contexts (measured: tools/multichain.py) the waiting kernels cost the other chains more than they gain.
std::atomic<int> g_live_contexts{0};

// Eigen streams of the live contexts.  The decompositions of one proposal share its work buffer and its warm-start chain, so
// they must run one after the other: they do, in launch order, as long as they are launched on ONE stream.  Ordinarily
// that is the context's own eigen stream; the chains of icp_chain_step_batched have theirs launched together on the eigen
// stream of the batch's first context.  A context whose decompositions move from one stream to another first waits, on the
// host, for those on the old one (a transition between single and batched stepping: rare) — if that stream still exists.
std::mutex g_eig_streams_mu;
std::set<hipStream_t> g_eig_streams;
}  // namespace

hipStream_t LazyStream::get() {
  if (s || device < 0) return s;
  int cur = -1;
  (void)hipGetDevice(&cur);
  if (cur != device) HIP_OK(hipSetDevice(device));
  s = take_stream(device, greatest, priority);
  if (eigen) { std::lock_guard<std::mutex> lk(g_eig_streams_mu); g_eig_streams.insert(s); }
  if (cur != device && cur >= 0) (void)hipSetDevice(cur);
  return s;
}
void LazyStream::arm(int dev, bool great, int prio, bool is_eigen, bool now) {
  device = dev; greatest = great; priority = prio; eigen = is_eigen;
  if (now) (void)get();
}
void LazyStream::sync() { if (s) HIP_OK(hipStreamSynchronize(s)); }
