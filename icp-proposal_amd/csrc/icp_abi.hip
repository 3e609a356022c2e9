// icp_abi.hip — host side of libicp_proposal_amd.so: the C ABI of include/icp_proposal.h.
//
// Owns device memory, the per-context HIP stream, the state / posterior / likelihood caches that stand in
// for the reference's Memoize wrappers (NonRigidIcpProposal.scala:49, evaluators/EvaluationCaching.scala:32),
// and the order in which kernels are enqueued.  Every entry point enqueues its whole kernel sequence on the
// context stream and synchronises ONCE, when results are copied back.  No CPU fallback exists: without a HIP
// device icp_ctx_create fails.
#include "../../include/icp_proposal.h"
#include "../../include/icp_sincos.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <set>
#include <thread>
#include <vector>

#include "icp_kernels.hpp"

using namespace icp;

namespace {

thread_local std::string g_err;

struct IcpError {
  int code;
  std::string msg;
};

[[noreturn]] void fail(int code, const std::string& msg) { throw IcpError{code, msg}; }

#define HIP_OK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) fail(ICP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// Device buffers of destroyed objects are kept for the next ones, by exact size and device (device_alloc / device_free below): a batch
// registration makes its chains anew for every target — 20 memoised posteriors of 13 buffers per proposal, 25,000 hipMalloc + hipFree
// in a job of 10 targets x 10 chains (0.38 s of its 1.35 s with chains of 50 steps; hipFree waits for the device every time).  A
// buffer goes back after the device has finished — once per destroyed object (DeviceQuiesce) instead of once per buffer — and comes out
// with whatever it held: as from hipMalloc, nothing may be assumed about a new buffer's contents (every completion word, counter and
// status of this file is set when its buffer is made).  ICP_NO_POOL=1: plain hipMalloc / hipFree.
void* device_alloc(size_t bytes);
void device_free(void* p, size_t bytes);
struct DeviceQuiesce {  // scope of an object's destruction: ONE wait for the device in front of the buffers' return
  DeviceQuiesce();
  ~DeviceQuiesce();
};

template <class T>
struct DBuf {
  T* p = nullptr;
  size_t n = 0;
  bool owned = true;  // false: a view of a buffer another object owns (immutable model / target data shared between contexts)
  DBuf() = default;
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  ~DBuf() { release(); }
  void release() {
    if (p && owned) device_free(p, sizeof(T) * (n ? n : 1));
    p = nullptr;
    n = 0;
    owned = true;
  }
  void alias(const DBuf& o) {
    release();
    p = o.p;
    n = o.n;
    owned = false;
  }
  void alloc(size_t count) {
    release();
    n = count;
    p = (T*)device_alloc(sizeof(T) * (count ? count : 1));
  }
  void upload(const T* src, size_t count) {
    alloc(count);
    if (count) {
      HIP_OK(hipMemcpy(p, src, sizeof(T) * count, hipMemcpyHostToDevice));
      HIP_OK(hipStreamSynchronize(nullptr));  // (as fill_bytes: the copy has reached the device before any launch can read it)
    }
  }
  // (hipMemset may return before the device has filled device memory, and what it enqueues on the null stream is not ordered against
  // this library's non-blocking streams: a launch issued right behind it could see — or, worse, count into — the buffer before the
  // fill lands.  Seen once the device was busy with another thread's batches: a batch gate's arrival counter zeroed AFTER the first
  // arrivals, every later gate of that slot two seconds late.  The null stream is waited for here.)
  void fill_bytes(int v) {
    HIP_OK(hipMemset(p, v, sizeof(T) * (n ? n : 1)));
    HIP_OK(hipStreamSynchronize(nullptr));
  }
};

constexpr double kSigma2 = 1e-5;  // regularisation of Scalismo's DiscreteLowRankGaussianProcess.coefficients
constexpr int kStateSlots = 8;
constexpr int kPosteriorMemo = 20;  // NonRigidIcpProposal.scala:49
constexpr int kEvalMemo = 3;        // evaluators/EvaluationCaching.scala:32
constexpr int kMaxRank = 500;
constexpr int kCholMaxRankAbi = 256;  // (= kCholMaxRank of kernels_posterior.hip: ranks whose factorisation hands the factor out)

// ---- host-side mesh preprocessing (one-off, at context creation)

void boundary_flags(int V, int T, const int32_t* tris, std::vector<uint8_t>& flags) {
  std::vector<int64_t> keys(3 * (size_t)T);
  for (int t = 0; t < T; ++t)
    for (int e = 0; e < 3; ++e) {
      int64_t a = tris[3 * t + e], b = tris[3 * t + (e + 1) % 3];
      if (a > b) std::swap(a, b);
      keys[3 * (size_t)t + e] = a * (int64_t)V + b;
    }
  std::sort(keys.begin(), keys.end());
  flags.assign(V, 0);
  for (size_t i = 0; i < keys.size();) {
    size_t j = i + 1;
    while (j < keys.size() && keys[j] == keys[i]) ++j;
    if (j - i == 1) {  // edge owned by exactly one triangle (Scalismo pointIsOnBoundary)
      flags[keys[i] / V] = 1;
      flags[keys[i] % V] = 1;
    }
    i = j;
  }
}

void vertex_adjacency(int V, int T, const int32_t* tris, std::vector<int>& off, std::vector<int>& adj) {
  off.assign(V + 1, 0);
  for (int i = 0; i < 3 * T; ++i) off[tris[i] + 1]++;
  for (int v = 0; v < V; ++v) off[v + 1] += off[v];
  adj.assign(std::max(3 * T, 1), 0);
  std::vector<int> fill(V, 0);
  for (int t = 0; t < T; ++t)
    for (int e = 0; e < 3; ++e) {
      int v = tris[3 * t + e];
      adj[off[v] + fill[v]++] = t;  // ascending triangle id per vertex
    }
}

bool host_cholesky(int n, std::vector<double>& a) {
  for (int j = 0; j < n; ++j) {
    double s = a[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) s -= a[(size_t)j * n + k] * a[(size_t)j * n + k];
    if (!(s > 0.0)) return false;
    double l = std::sqrt(s);
    a[(size_t)j * n + j] = l;
    for (int i = j + 1; i < n; ++i) {
      double v = a[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) v -= a[(size_t)i * n + k] * a[(size_t)j * n + k];
      a[(size_t)i * n + j] = v / l;
    }
  }
  return true;
}

// inverse of an SPD matrix from its Cholesky factor (one-off host work at context creation)
bool host_spd_inverse(int n, const std::vector<double>& a, std::vector<double>& inv) {
  std::vector<double> l = a;
  if (!host_cholesky(n, l)) return false;
  inv.assign((size_t)n * n, 0.0);
  std::vector<double> e(n);
  for (int c = 0; c < n; ++c) {
    std::fill(e.begin(), e.end(), 0.0);
    e[c] = 1.0;
    for (int i = 0; i < n; ++i) {
      double v = e[i];
      for (int k = 0; k < i; ++k) v -= l[(size_t)i * n + k] * e[k];
      e[i] = v / l[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double v = e[i];
      for (int k = i + 1; k < n; ++k) v -= l[(size_t)k * n + i] * e[k];
      e[i] = v / l[(size_t)i * n + i];
    }
    for (int i = 0; i < n; ++i) inv[(size_t)i * n + c] = e[i];
  }
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      double v = 0.5 * (inv[(size_t)i * n + j] + inv[(size_t)j * n + i]);
      inv[(size_t)i * n + j] = inv[(size_t)j * n + i] = v;
    }
  return true;
}

void check_triangles(int V, int T, const int32_t* tris, const char* what) {
  for (int i = 0; i < 3 * T; ++i)
    if (tris[i] < 0 || tris[i] >= V) fail(ICP_ERR_INVALID_ARG, std::string(what) + ": triangle vertex id out of range");
}

// Rotation(phi,theta,psi,centre) = Rz(phi)·Ry(theta)·Rx(psi), with the sines and cosines of
// include/icp_sincos.h: plain arithmetic, the same bits here, on the device (the pose walks of the on-device chain loop) and in the oracle
Pose pose_from_theta(const double* th) {
  Pose p;
  icp_rotation_matrix(th[4], th[5], th[6], p.R);
  for (int d = 0; d < 3; ++d) { p.t[d] = th[1 + d]; p.ctr[d] = th[7 + d]; }
  p.s = th[0];
  return p;
}

struct DeviceMesh {
  int V = 0, T = 0, n_boundary = 0;
  DBuf<double> verts;
  DBuf<int> tris;
  DBuf<int> tri_order;       // position in the sphere list -> triangle (coherent_triangle_order)
  DBuf<float4> spheres;      // sphere_floats4(T): spheres in that order, then the triangle ids
  DBuf<uint8_t> boundary;
};

// Immutable device data of one statistical model / one target mesh, shared by every context of a device that was created from
// the same arrays (64 chains on one GPU have 64 contexts — per-chain scratch, caches, streams — but ONE copy of the basis and of
// the target; the BFM-sized model is 2 x 137 MB).  Contexts hold them through shared_ptr and address them through aliasing DBufs.
struct SharedModel {
  DBuf<double> ref, mean, Q, Qp, sqrt_lambda, inv_sqrt_lambda, G, Ginv, P;
  DBuf<int> tris, adj_off, adj, tri_order;
  DBuf<uint8_t> boundary;
  int n_boundary = 0;
  int device = 0;
};
struct SharedTarget {
  DeviceMesh mesh;
};
struct SharedKey {
  int device, a, b, c;
  uint64_t hash;
  bool operator<(const SharedKey& o) const {
    if (device != o.device) return device < o.device;
    if (a != o.a) return a < o.a;
    if (b != o.b) return b < o.b;
    if (c != o.c) return c < o.c;
    return hash < o.hash;
  }
};
std::mutex g_shared_mu;
std::map<SharedKey, std::weak_ptr<SharedModel>> g_shared_models;
std::map<SharedKey, std::weak_ptr<SharedTarget>> g_shared_targets;
// The two most recently used models stay alive between contexts (icp_release_cached_models drops them): a batch registration
// builds one context per target, one after the other, over the SAME model — whose derived data (Q in two layouts, the Gram
// matrix QᵀQ on the host, two r × r inverses: 0.35 s at N = 28,561, rank 200) was rebuilt for every target once the previous
// target's context, its last user, had been destroyed.
// (on the heap and never destroyed: at process exit the runtime may be gone before this library's static objects are)
std::shared_ptr<SharedModel>* const g_model_keep = new std::shared_ptr<SharedModel>[2];
int g_model_keep_next = 0;

uint64_t hash_words_serial(uint64_t h, const void* data, size_t bytes);
// … and large arrays in pieces on several threads, the pieces' hashes hashed in order (the same value whatever the thread count:
// the pieces are fixed 8 MiB): 137 MB of basis in ≈ 1.5 ms instead of 9 — per context created (a batch registration makes dozens)
uint64_t hash_words(uint64_t h, const void* data, size_t bytes) {
  constexpr size_t kPiece = (size_t)8 << 20;
  if (bytes < 2 * kPiece) return hash_words_serial(h, data, bytes);
  const size_t n = (bytes + kPiece - 1) / kPiece;
  std::vector<uint64_t> part(n);
  const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::thread> th;
  std::atomic<size_t> next{0};
  auto work = [&] {
    for (size_t i; (i = next.fetch_add(1)) < n;)
      part[i] = hash_words_serial(0x9E3779B97F4A7C15ull + i, (const unsigned char*)data + i * kPiece, std::min(kPiece, bytes - i * kPiece));
  };
  for (unsigned t = 1; t < hw; ++t) th.emplace_back(work);
  work();
  for (auto& t : th) t.join();
  return hash_words_serial(h, part.data(), sizeof(uint64_t) * n);
}
// word-wise multiply-xor (identity of the arrays, not security); four independent lanes: one lane's dependent multiply chain
// made 35 ms of every context creation at the face model's 137 MB of basis
uint64_t hash_words_serial(uint64_t h, const void* data, size_t bytes) {
  const unsigned char* p = (const unsigned char*)data;
  constexpr uint64_t kMul = 0x9E3779B97F4A7C15ull;
  uint64_t a = h, b = h ^ 0x243F6A8885A308D3ull, c = h ^ 0x13198A2E03707344ull, d = h ^ 0xA4093822299F31D0ull;
  size_t i = 0;
  for (; i + 32 <= bytes; i += 32) {
    uint64_t w[4];
    std::memcpy(w, p + i, 32);
    a = (a ^ w[0]) * kMul; a ^= a >> 29;
    b = (b ^ w[1]) * kMul; b ^= b >> 29;
    c = (c ^ w[2]) * kMul; c ^= c >> 29;
    d = (d ^ w[3]) * kMul; d ^= d >> 29;
  }
  h = a;
  h = (h ^ b) * kMul; h ^= h >> 29;
  h = (h ^ c) * kMul; h ^= h >> 29;
  h = (h ^ d) * kMul; h ^= h >> 29;
  for (; i + 8 <= bytes; i += 8) {
    uint64_t w;
    std::memcpy(&w, p + i, 8);
    h = (h ^ w) * kMul;
    h ^= h >> 29;
  }
  for (; i < bytes; ++i) h = (h ^ p[i]) * 0x100000001B3ull;
  return h;
}

struct QueryScratch {
  DBuf<double> thr2;
  DBuf<float4> qrec;
  DBuf<float> thrA;
  DBuf<int> cnt, cand;
  size_t cap = 0, cand_cap = 0;
  QueryBuffers get() const { return QueryBuffers{thr2.p, qrec.p, thrA.p, cnt.p, cand.p, cand_cap}; }
};

constexpr size_t kMaxCandidates = (size_t)64 << 20;  // ints (256 MiB of candidate lists, 512 per query): more queries than that are batched

struct StateSlot {
  std::vector<double> theta;
  bool valid = false;
  bool reserved = false;  // handed out to a step whose launches are in flight: not to be recycled
  uint64_t stamp = 0;
  Pose pose;
  DBuf<double> coeffs, x;
  DBuf<double> defo;        // per point mean + Q·c (icp_ctx::state: a pose move re-poses these instead of reading the basis again)
  bool defo_valid = false;
  DBuf<float4> spheres;
  bool spheres_valid = false;
  int n_surf = 0;  // model ids [0, n_surf) already projected onto the target surface
  DBuf<double> surf_cp, surf_d2;
  DBuf<int> surf_tri;
  int n_nnv = 0;   // ... and their surface points already matched to the nearest target vertex
  DBuf<int> surf_nnv;
  // … plus one detached range each, [lo, hi): searched ahead of ids that were left to another stream (ensure_*_prefix(…, reserve));
  // joined to the prefix as soon as that reaches lo
  int lo_surf = 0, hi_surf = 0, lo_nnv = 0, hi_nnv = 0;
};

}  // namespace

// icp_runtime_stats (include/icp_proposal.h): per context and for the process
struct RuntimeStats {
  std::atomic<int64_t> wait_timeouts{0}, speculation_giveups{0}, pipeline_fallbacks{0}, step_redos{0}, gate_timeouts{0};
};
RuntimeStats g_runtime_stats;
// which path the chain steps took (icp_ctx_step_paths): [0] the five merged launches, [1] the wide step, [2] per-stage kernels,
// [3] steps inside icp_chains_run_on_device
struct StepPaths { std::atomic<int64_t> n[4] = {{0}, {0}, {0}, {0}}; };
StepPaths g_step_paths;

struct icp_ctx {
  int device = 0;
  RuntimeStats stats;
  StepPaths paths;
  hipStream_t stream = nullptr;
  // icp_chain_step alternates between two streams: the five launches of a step go to one of them in order, the next step's
  // to the other.  Launches 1-3 of a step do not depend on the finish launch of the step before it and run beside it; what
  // they must not overtake is that step's searches (same scratch, same hints), so launch 1 waits on the device for the word
  // the finish launch of that step raises when it starts (StepBeginArgs::wait_flag).  No event crosses the two streams.
  hipStream_t front_stream = nullptr;            // the second of the two (`stream` is the first, and everybody else's)
  // every eigen-decomposition of the context runs on this stream, beside the chain (launch order = execution order, so the
  // decompositions of one proposal never overlap each other; the two directions of a step share ONE launch)
  hipStream_t eig_stream = nullptr;
  hipStream_t eig_stream2 = nullptr;  // ranks above 64: decompositions started ahead alternate between the two (each with a work buffer of its own)
  hipStream_t eig_last2 = nullptr; // (the wide step's second eigen stream, see batch_eig2)
  hipStream_t eig_last = nullptr;  // where this context's latest decompositions were launched: eig_stream, or the eigen stream
                                   // of the first context of a batch (see eigen_stream_for)
  hipEvent_t ev_ready = nullptr;                 // stream -> eig_stream: "M is complete"
  hipEvent_t ev_side = nullptr;                  // front_stream -> stream: factorisations / tails that went to the side stream are done
  hipEvent_t ev_sum = nullptr;                   // front_stream -> eig_stream: the partials of the latest posterior are summed
  hipEvent_t ev_asm = nullptr;                   // eig_stream -> stream: … and read (the next regression may overwrite them)
  hipEvent_t ev_join = nullptr;                  // stream -> front_stream, when another entry point has used `stream`
  hipEvent_t ev_inst = nullptr;                  // stream -> side: "the state's points are complete" (a posterior whose searches run on the side stream)
  const void* ev_inst_slot = nullptr;            // … the state slot it was recorded for by the caller of posterior(…, side), if any
  hipEvent_t ev_front = nullptr;                 // side -> stream: "… and so are its searches' results" (the evaluator's reductions read them)
  bool front_on_side = false;                    // ev_front is on record and nobody has waited for it yet
  int front_side_K = 0;                          // … the model ids 0..K whose surface search is part of that front (0: none)
  bool front_stream_used = false;                // a step is (or may still be) on front_stream: other entry points drain it first
  // ICP_NO_PIPELINE=1, or a first launch once timed out on its word (a tool that lets one kernel run at a time, in an order
  // of its own): every step on `stream`, nothing launched ahead, no device-side waits
  bool pipeline_off = std::getenv("ICP_NO_PIPELINE") != nullptr;
  bool stream_used_elsewhere = false;            // an entry point other than the chain step has enqueued on `stream`
  int last_back_seq = 0;                         // sequence number of the last finish launch
  int* h_wait_error = nullptr;                   // pinned: a front gave up waiting (never expected)
  std::recursive_mutex mu;
  int N = 0, T = 0, r = 0;
  DBuf<double> ref, mean, Q, Qp, sqrt_lambda, inv_sqrt_lambda, G, Ginv, P;  // P = (G + σ²I)⁻¹
  DBuf<int> tris, adj_off, adj;
  DBuf<int> tri_order;  // sphere-list order of the model's triangles (from the reference shape; patches stay patches under the model's deformations)
  DBuf<uint8_t> boundary;
  int n_boundary = 0;
  DeviceMesh target;
  std::shared_ptr<SharedModel> shared_model;    // owners of what the members above alias (ref … boundary; target.*)
  std::shared_ptr<SharedTarget> shared_target;
  DBuf<int> hint_surf;  // [N] last target triangle of model id i
  DBuf<int> hint_nnv;   // [N] last nearest target vertex of that surface point
  StateSlot slots[kStateSlots];
  uint64_t clock = 0;
  QueryScratch scratch;
  QueryScratch scratch_v;  // second scratch: the merged step launches run a surface and a vertex search side by side
  QueryScratch scratch_t;  // third: … and the evaluator's target -> model surface search
  QueryScratch scratch_n;  // the wide step's second search stage: nearest target vertices of the model-side surface points …
  QueryScratch scratch_tn; // … and nearest model vertices of the evaluator's target-side surface points (their own candidate counters)
  QueryScratch scratch_p;  // the proposal's own model ids where the evaluator's searches run as a sequence of their own (the wide step)
  QueryScratch scratch_en; // … and the nearest vertices of the evaluator's own ids in that case
  // staging for small host<->device transfers of one API call
  double* h_stage = nullptr;  // pinned
  DBuf<double> d_stage;
  size_t stage_cap = 0, stage_used = 0;
  // results of one API call: [64 status ints | res_cap doubles] in ONE device block and one pinned block of the same layout, so
  // that a call's statuses and results come back in a single copy (finish)
  static constexpr size_t kStatusDoubles = 48;  // 96 status ints: [0,16) the tails' own, [16,64) their posteriors' (relayed), [64] the direct tail's
  double* h_out = nullptr;    // pinned
  DBuf<double> d_out;
  double* h_res = nullptr;    // = h_out + kStatusDoubles
  DBuf<double> d_res;         // view
  int* h_status = nullptr;    // = (int*)h_out
  DBuf<int> d_status;         // view
  DBuf<int> d_done;            // [0] completion counter of the step's last launch; [1] counter and [2] "partials ready" word
                               // of its regression launch
  int* h_flag = nullptr;       // pinned: sequence number of the last finished step
  int step_seq = 0;
  std::vector<struct icp_evaluator*> evaluators;  // live evaluators (a proposal being destroyed drops their pending half steps)
  std::vector<struct icp_proposal*> proposals;    // live proposals (icp_ctx_set_rotation forgets what they memoised under a triple)
  icp_idle_fn idle_fn = nullptr;  // icp_ctx_set_idle_hook
  void* idle_arg = nullptr;
  bool counted = false;          // included in g_live_contexts
  bool speculation_off = false;  // a speculative decomposition timed out once (see resolve_speculation): not tried again
  // member of a batch between icp_chain_step_batched_issue and _collect / _abandon (set and cleared under `mu`, which is NOT
  // held in between): every other entry point on this context fails with ICP_ERR_BUSY meanwhile
  bool batch_busy = false;

  Profiler prof;
  bool profiling = false;
  DBuf<long long> d_wait_ticks;  // profiling: time the steps' first launches spent waiting on the device (StepBeginArgs::wait_ticks)
  DBuf<unsigned long long> d_search_counters;  // profiling: executed tests of the searches (SurfaceTask::stats)
  bool count_searches = false;                 // icp_ctx_profile_search_counters
  // argument arrays of the icp_chain_step_batched launches led by this context: pinned copy, device copy
  // (kBatchRing of each, used in turn: a caller may keep that many batches in flight on this context's stream)
  static constexpr int kBatchRing = ICP_MAX_BATCHES_IN_FLIGHT;
  // tickets issued on this launch context and not yet collected / abandoned: one more than the ring holds would rewrite the pinned
  // argument slot, the eigen records and the gate word of a batch still on the device — refused with ICP_ERR_BUSY (icp_chain_step_batched_issue)
  std::atomic<int> tickets_in_flight{0};
  void* batch_pinned[kBatchRing] = {};
  DBuf<unsigned char> batch_device[kBatchRing];
  size_t batch_bytes[kBatchRing] = {};
  int batch_turn = 0;
  // the wide step (kernels_wide.hip) led by this context: per-chain records (pinned + device copy), events stream -> side streams
  void* wide_pinned[kBatchRing] = {};
  DBuf<unsigned char> wide_device[kBatchRing];
  size_t wide_bytes[kBatchRing] = {};
  hipEvent_t ev_wide_sum[kBatchRing] = {};   // stream -> eigen / finish streams: the partials are summed
  hipEvent_t ev_wide_fac[kBatchRing] = {};   // finish stream -> eigen stream: M is complete (ranks <= 64)
  hipEvent_t ev_wide_head[kBatchRing] = {};  // stream -> second stream: the new instances are complete
  hipEvent_t ev_wide_eval[kBatchRing] = {};  // stream -> second stream: the evaluator's own sequence is through
  int wide_turn = 0;
  double* h_wide_z = nullptr;  // pinned: the coefficients a wide step is GIVEN (random-walk / pose proposals), read by its first launch
  // … and of their decompositions: the records of launch_posterior_eigen_many (pinned, read in place by the kernel), the counter
  // its workgroups announce themselves in and what it will hold once every workgroup launched so far has started (the gate of
  // launch_step_batch) and the gate's pinned error word
  void* batch_eig_rec[kBatchRing] = {};
  size_t batch_eig_rec_bytes[kBatchRing] = {};
  int batch_eig_turn = 0;
  DBuf<int> batch_gate;                                   // one counter word per ring slot (a later batch's workgroups must not open an earlier batch's gate)
  int batch_gate_expected[kBatchRing] = {};
  int* h_gate_error = nullptr;
  // eigen streams of the batches this context carries, one per batch in flight (keyed by the batch's first chain): created
  // together, so that the runtime spreads them over different hardware queues — the member contexts' own eigen streams
  // collide on one queue for some batch sizes (24 chains in three groups: 49k instead of 70k it/s)
  hipStream_t batch_eig[kBatchRing] = {};
  hipStream_t batch_eig2[kBatchRing] = {};  // … and a second one each (the wide step alternates: two decompositions of a chain in flight)
  const void* batch_eig_owner[kBatchRing] = {};
  int batch_eig_evict = 0;

  void bind() { HIP_OK(hipSetDevice(device)); }

  // scratch for K queries against a set of n_elems elements (every query may list every element as a candidate)
  QueryBuffers query_scratch(size_t K, size_t n_elems, int which = 0) {
    QueryScratch& scratch = which == 1 ? scratch_v : which == 2 ? scratch_t : which == 3 ? scratch_n : which == 4 ? scratch_tn : which == 5 ? scratch_p : which == 6 ? scratch_en : this->scratch;
    if (K > scratch.cap) {
      HIP_OK(hipStreamSynchronize(stream));
      if (front_stream) HIP_OK(hipStreamSynchronize(front_stream));
      size_t cap = std::max<size_t>(K, 4096);
      scratch.thr2.alloc(cap + 8);
      scratch.qrec.alloc(cap + 8);
      scratch.thrA.alloc(cap + 8);
      scratch.cnt.alloc(cap + 8);
      scratch.cap = cap;
    }
    const size_t want = std::min(kMaxCandidates, std::max<size_t>((K + 4) * std::min<size_t>(std::max<size_t>(n_elems, 1), (size_t)kCandStrideMax), 1));
    if (want > scratch.cand_cap) {
      HIP_OK(hipStreamSynchronize(stream));
      if (front_stream) HIP_OK(hipStreamSynchronize(front_stream));
      scratch.cand.alloc(want);
      scratch.cand_cap = want;
    }
    return scratch.get();
  }

  // copies `count` doubles to the device through the pinned staging area (valid until the call's final sync)
  const double* stage(const double* src, size_t count) {
    if (stage_used + count > stage_cap) fail(ICP_ERR_INVALID_ARG, "internal: staging area exhausted");
    double* h = h_stage + stage_used;
    double* d = d_stage.p + stage_used;
    std::memcpy(h, src, sizeof(double) * count);
    HIP_OK(hipMemcpyAsync(d, h, sizeof(double) * count, hipMemcpyHostToDevice, stream));
    stage_used += count;
    return d;
  }

  // same, into a device buffer of the caller's (one copy instead of staging + device-to-device)
  void stage_to(double* dst, const double* src, size_t count) {
    if (stage_used + count > stage_cap) fail(ICP_ERR_INVALID_ARG, "internal: staging area exhausted");
    double* h = h_stage + stage_used;
    std::memcpy(h, src, sizeof(double) * count);
    HIP_OK(hipMemcpyAsync(dst, h, sizeof(double) * count, hipMemcpyHostToDevice, stream));
    stage_used += count;
  }

  void finish(size_t n_res, size_t n_status) {
    if (n_status) HIP_OK(hipMemcpyAsync(h_out, d_out.p, sizeof(double) * (kStatusDoubles + n_res), hipMemcpyDeviceToHost, stream));
    else if (n_res) HIP_OK(hipMemcpyAsync(h_res, d_res.p, sizeof(double) * n_res, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    stage_used = 0;
  }

  // Rotation matrices supplied by the caller for given Euler triples (icp_ctx_set_rotation): the reference delegates
  // Rotation(phi, theta, psi, centre) to Scalismo (ModelFittingParameters.scala:79-86), whose convention cannot be verified in
  // this image — a host that passes Scalismo's own matrix keeps that convention its own; without an entry for a triple the
  // library's Rz·Ry·Rx is used.  Small LRU table, exact comparison of the three angles.
  struct RotationEntry { double angles[3]; double R[9]; uint64_t stamp; bool valid = false; };
  static constexpr int kRotationEntries = 32;
  RotationEntry rotations[kRotationEntries];
  uint64_t rotation_clock = 0;
  // Convention check (icp_ctx_set_rotation): every supplied matrix is compared with the library's own Rz(phi)·Ry(theta)·Rx(psi)
  // (include/icp_sincos.h).  A host whose matrices all agree to rounding (kRotationTol per entry) has the library's convention —
  // Scalismo's Rotation(phi, theta, psi, centre) of ModelFittingParameters.scala:79-86, if the host is the Scala adapter — and the pose
  // walks of the on-device loop, which make the proposed pose's matrix on the device, are open to it; one disagreement closes them
  // for this context for good (icp_chains_run_on_device; icp_ctx_rotation_convention reports both counts).
  static constexpr double kRotationTol = 2e-15;
  int64_t rotations_verified = 0, rotations_mismatched = 0;
  Pose pose_of(const double* theta);

  StateSlot& state(const double* theta);
  StateSlot* find_state(const double* theta);
  StateSlot& fresh_state();
  void alloc_slot(StateSlot& s);
  void ensure_model_spheres(StateSlot& s);
  // (st / which: the stream and the scratch set of the search — the context stream and set 0 unless a posterior runs its searches aside)
  // reserve > prefix: the ids between them are left to somebody else (a posterior's searches on the side stream, issued next); what
  // lies behind them is searched now, as a detached range
  void ensure_surface_prefix(StateSlot& s, int K, hipStream_t st = nullptr, int which = 0, int reserve = 0);
  void ensure_nnv_prefix(StateSlot& s, int K, hipStream_t st = nullptr, int which = 0, int reserve = 0);
};

namespace {
struct Bound {  // selects the context's device and (if enabled) its profiler for the calling thread
  // chain_path: the caller is the merged chain step, which orders its two streams itself.  Every other entry point works
  // on `stream` alone and shares scratch with the fronts: it first lets `stream` wait for the last front in flight.
  explicit Bound(icp_ctx* c, bool chain_path = false, bool batch_owner = false) {
    if (c->batch_busy && !batch_owner) throw IcpError{ICP_ERR_BUSY, "the context belongs to a batch in flight (icp_chain_step_batched_issue): collect or abandon it first"};
    c->bind();
    g_prof = c->profiling ? &c->prof : nullptr;
    if (!chain_path) {
      if (c->front_stream_used) { (void)hipStreamSynchronize(c->front_stream); c->front_stream_used = false; }
      c->stream_used_elsewhere = true;
    }
  }
  ~Bound() { g_prof = nullptr; }
};
}  // namespace

Pose icp_ctx::pose_of(const double* theta) {
  Pose p = pose_from_theta(theta);
  for (auto& e : rotations)
    if (e.valid && e.angles[0] == theta[4] && e.angles[1] == theta[5] && e.angles[2] == theta[6]) {
      for (int k = 0; k < 9; ++k) p.R[k] = e.R[k];
      e.stamp = ++rotation_clock;
      break;
    }
  return p;
}

StateSlot* icp_ctx::find_state(const double* theta) {
  const size_t P = 10 + (size_t)r;
  for (auto& s : slots)
    if (s.valid && std::memcmp(s.theta.data(), theta, sizeof(double) * P) == 0) return &s;
  return nullptr;
}

// least recently used slot, emptied (buffers allocated on first use); the caller fills it and sets `valid`
// device buffers of a state slot (all slots at context creation: an allocation is a synchronising runtime call of 50-100 µs,
// which a chain's first steps would otherwise pay one slot at a time)
void icp_ctx::alloc_slot(StateSlot& s) {
  if (s.x.p) return;
  s.coeffs.alloc(r);
  s.x.alloc(3 * (size_t)N);
  s.defo.alloc(3 * (size_t)N);
  s.spheres.alloc(sphere_floats4(T));
  s.surf_cp.alloc(3 * (size_t)N);
  s.surf_d2.alloc(N);
  s.surf_tri.alloc(N);
  s.surf_nnv.alloc(N);
}

StateSlot& icp_ctx::fresh_state() {
  StateSlot* lru = nullptr;
  for (auto& s : slots) {
    if (s.reserved) continue;
    if (!lru) { lru = &s; continue; }
    if (!s.valid) { if (lru->valid) lru = &s; }
    else if (lru->valid && s.stamp < lru->stamp) lru = &s;
  }
  if (!lru) fail(ICP_ERR_DEVICE, "internal: every state slot is reserved");
  StateSlot& s = *lru;
  alloc_slot(s);
  s.valid = false;
  s.defo_valid = false;
  s.spheres_valid = false;
  s.n_surf = 0;
  s.n_nnv = 0;
  s.lo_surf = s.hi_surf = s.lo_nnv = s.hi_nnv = 0;
  return s;
}

StateSlot& icp_ctx::state(const double* theta) {
  const size_t P = 10 + (size_t)r;
  if (StateSlot* hit = find_state(theta)) {
    hit->stamp = ++clock;
    return *hit;
  }
  // a pose move (PoseProposals.scala: 0.4 of the configs[3]/[4] mixture) leaves the coefficients alone: the points are the kept
  // deformations of a state with the same coefficients under the new pose — 0.7 MB instead of the basis' 137 MB at N = 28,561, rank 200
  StateSlot* same = nullptr;
  for (auto& o : slots)
    if (o.valid && o.defo_valid && std::memcmp(o.theta.data() + 10, theta + 10, sizeof(double) * r) == 0) { same = &o; break; }
  if (same) same->stamp = ++clock;  // (not the one recycled below, unless every other slot is reserved: in place works, too)
  StateSlot& s = fresh_state();
  s.theta.assign(theta, theta + P);
  s.valid = true;
  s.stamp = ++clock;
  s.pose = pose_of(theta);
  stage_to(s.coeffs.p, theta + 10, r);
  if (same) launch_instance_pose(stream, N, ref.p, s.pose, same->defo.p, s.x.p, s.defo.p);
  else launch_instance_keep(stream, N, r, Qp.p, ref.p, mean.p, s.pose, s.coeffs.p, s.x.p, s.defo.p);  // ModelFittingParameters.scala:108-110
  s.defo_valid = true;
  return s;
}

void icp_ctx::ensure_model_spheres(StateSlot& s) {
  if (s.spheres_valid) return;
  launch_tri_spheres(stream, T, s.x.p, tris.p, tri_order.p, s.spheres.p);
  s.spheres_valid = true;
}

// target.operations.closestPointOnSurface(currentMesh.point(id)) for id in [0, K) (NonRigidIcpProposal.scala:96-97,
// IndependentPointDistanceEvaluator.scala:41-43): shared by every proposal / evaluator of this context.
void icp_ctx::ensure_surface_prefix(StateSlot& s, int K, hipStream_t st, int which, int reserve) {
  if (K > N) fail(ICP_ERR_INVALID_ARG, "model id count exceeds the number of model points");
  auto join = [&] { if (s.hi_surf > 0 && s.n_surf >= s.lo_surf) { s.n_surf = std::max(s.n_surf, s.hi_surf); s.lo_surf = s.hi_surf = 0; } };
  auto search = [&](int k0, int k1) {
    const int n = k1 - k0;
    QueryBuffers qb = query_scratch(n, target.T, which);
    launch_surface_query(st ? st : stream, target.T, target.verts.p, target.tris.p, target.spheres.p, n, s.x.p + 3 * (size_t)k0,
                         hint_surf.p + k0, qb, s.surf_cp.p + 3 * (size_t)k0, s.surf_d2.p + k0, s.surf_tri.p + k0);
  };
  join();
  if (K <= s.n_surf) return;
  if (reserve > s.n_surf) {  // ids [prefix, reserve) are somebody else's: [reserve, K) detached (once)
    if (s.hi_surf == 0 && reserve < K) { search(reserve, K); s.lo_surf = reserve; s.hi_surf = K; }
    return;
  }
  while (s.n_surf < K) {
    const int k1 = s.hi_surf > 0 ? std::min(K, s.lo_surf) : K;  // (up to a detached range, which then joins)
    if (k1 > s.n_surf) search(s.n_surf, k1);
    s.n_surf = std::max(s.n_surf, k1);
    join();
  }
}

// target.pointSet.findClosestPoint(targetPoint).id (NonRigidIcpProposal.scala:98)
void icp_ctx::ensure_nnv_prefix(StateSlot& s, int K, hipStream_t st, int which, int reserve) {
  ensure_surface_prefix(s, K, st, which, reserve);
  auto join = [&] { if (s.hi_nnv > 0 && s.n_nnv >= s.lo_nnv) { s.n_nnv = std::max(s.n_nnv, s.hi_nnv); s.lo_nnv = s.hi_nnv = 0; } };
  auto search = [&](int k0, int k1) {
    const int n = k1 - k0;
    QueryBuffers qb = query_scratch(n, target.V, which);
    launch_vertex_query(st ? st : stream, target.V, target.verts.p, n, s.surf_cp.p + 3 * (size_t)k0, hint_nnv.p + k0, qb, nullptr,
                        s.surf_nnv.p + k0);
  };
  join();
  if (K <= s.n_nnv) return;
  if (reserve > s.n_nnv) {
    if (s.hi_nnv == 0 && reserve < K) { search(reserve, K); s.lo_nnv = reserve; s.hi_nnv = K; }
    return;
  }
  while (s.n_nnv < K) {
    const int k1 = s.hi_nnv > 0 ? std::min(K, s.lo_nnv) : K;
    if (k1 > s.n_nnv) search(s.n_nnv, k1);
    s.n_nnv = std::max(s.n_nnv, k1);
    join();
  }
}

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

void drain_pools() {
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

struct PosteriorEntry {
  std::vector<double> theta;
  bool valid = false, eig_valid = false, eig_checked = false;  // eig_checked: its status has reached the host copy
  bool reserved = false;  // handed out to a step whose launches are in flight: not to be recycled
  hipEvent_t eig_done = nullptr;  // recorded on the eigen stream behind the launch that holds the entry's decomposition
  hipEvent_t eig_done_shared = nullptr;  // … or, not owned, the event of the entry it shared that launch with
  // a shared event out of the device's ring (next_batch_event, icp_chain_step_batched): the slot's generation counter (static
  // storage) and its value when the event was recorded for this entry — a slot recorded again since stands for LATER work on
  // possibly another stream, which orders nothing of this entry's: such a waiter synchronises with the eigen streams on the host
  // instead (eigen_event() returns nullptr then; await_eigen / mpart_for_write fall back to sync_eigen)
  const uint64_t* eig_shared_gen = nullptr;
  uint64_t eig_shared_gen_value = 0;
  bool eigen_event_stale() const { return eig_event_valid && eig_done_shared && eig_shared_gen && *eig_shared_gen != eig_shared_gen_value; }
  bool eig_event_valid = false;   // an event stands for the latest decomposition of this entry (the chain step's own launches
                                  // of ranks <= 64 record none: their consumers wait for the completion word on the device, and
                                  // an event record is 2-3 µs of host time on the accepted path)
  int done_value = 0;             // … and what the entry's word in icp_proposal::eig_words holds once it is complete (0: none)
  uint64_t stamp = 0;
  DBuf<int> id, aux;
  DBuf<double> pt, nhat, e;
  DBuf<uint8_t> keep;
  DBuf<double> coeffs, M, alpha, V, Vt, S;
  int status_off = 0;  // this entry's 3 ints inside the proposal's status buffer
  ~PosteriorEntry() { if (eig_done) (void)hipEventDestroy(eig_done); }
  hipEvent_t eigen_event() const { return (!eig_event_valid || eigen_event_stale()) ? nullptr : (eig_done_shared ? eig_done_shared : eig_done); }
  CorrBuffers corr() const { return CorrBuffers{id.p, aux.p, pt.p, keep.p, nhat.p, e.p}; }
};

}  // namespace

struct icp_proposal {
  icp_ctx* ctx = nullptr;
  icp_proposal_params prm{};
  int K = 0;
  DBuf<double> target_pts;
  DBuf<int> hint_nn;      // TargetSampling: last nearest model vertex of each target point
  DBuf<int> nn_id;
  DBuf<double> work;      // r*r scratch of the eigen / direct-tail kernels
  DBuf<double> Mpart;     // split-K partial normal matrices of the regression kernel, two halves: the merged step alternates
                          // between them so that a speculative decomposition can still read the previous step's
  size_t mpart_half_doubles = 0;
  int mpart_half = 0;
  static constexpr int kMpartRing = 4;  // (a speculative decomposition reads the partials of the step that started it: with four
                                        // buffers used in turn the writer of a buffer practically never finds its reader still at work)
  PosteriorEntry* mpart_reader[kMpartRing] = {nullptr, nullptr, nullptr, nullptr};  // the entry whose decomposition reads the buffer
  double* mpart_for_write(int half, hipStream_t st);  // `st` (where the writer runs) waits for that reader first, if it is still at work
  DBuf<double> fscratch;  // (r+1)·r + 8 factorisation scratch (ranks too large for LDS)
  const double* warm_ptr = nullptr;  // eigenvectors of the most recent posterior (inside its memo entry): warm start of the next
  bool warm_valid = false;
  // The eigen-decompositions of this proposal run on the context's eigen stream (icp_ctx::eig_stream) in launch order — they
  // share `work` and the warm start, so they must not overlap each other — beside the chain's own streams: the decomposition
  // of a state that is not needed yet — the other ICP direction of a freshly accepted state — overlaps the chain's next
  // steps instead of delaying a later one.
  // Speculative decomposition (icp_chain_step, ICP_SPECULATION=1): the KL basis of the PROPOSED state's posterior is started as soon as its
  // normal matrix exists, before the caller has decided whether to accept.  The next call tells: its current state is
  // the proposed one (the basis is already on its way) or not (the decomposition is cancelled through `h_cancel`).
  int sampler = ICP_SAMPLER_EIGEN; // icp_proposal_set_sampler: what the "decomposition" of a posterior writes into V / S
  int* h_eig = nullptr;            // pinned: eigen status of every memo entry, written by the decomposition itself
  DBuf<int> eig_words;             // per memo entry: sequence number of its last finished decomposition (EigenRequest::done_word)
  int eig_seq = 0;
  int* h_cancel = nullptr;         // pinned, 16 slots: the decomposition with sequence number q gives up once slot q%16 holds q
  int spec_seq = 0;
  PosteriorEntry* spec_entry = nullptr;
  // fills the request of a speculative decomposition of `e` (the caller launches it, possibly together with another
  // proposal's, on the context's eigen stream and records e.eig_done behind it)
  void speculate_eigen(PosteriorEntry& e, const PosteriorEntry& cur, int splits, int half /* of Mpart: the step's partials */,
                       const int* ready, int ready_seq, EigenSpec* spec_out, EigenRequest* rq_out);
  void resolve_speculation(const double* theta_cur);
  DBuf<int> status;       // 3 ints per memo entry: {chol(M), chol(G+σ²M), eigen}
  // (pinned: the copy of `status` into it is a true asynchronous copy — into a pageable vector it was a synchronising one, 20-30 µs
  // per step of the per-stage paths)
  struct PinnedInts {
    int* p = nullptr; size_t n = 0;
    void assign(size_t count, int v) {
      if (p) pinned_free(p);
      pinned_alloc((void**)&p, sizeof(int) * count);
      n = count;
      for (size_t i = 0; i < count; ++i) p[i] = v;
    }
    int& operator[](size_t i) { return p[i]; }
    int* data() { return p; }
    ~PinnedInts() { if (p) pinned_free(p); }
  } h_status;
  std::unique_ptr<PosteriorEntry[]> memo;
  uint64_t clock = 0;

  // side: (optional) the stream the factorisation goes to, behind the regression on the context stream (see icp_chain_eval_step)
  PosteriorEntry& posterior(const double* theta, bool want_aux, hipStream_t side = nullptr);
  bool side_factor_pending = false;  // a factorisation on the side stream may still read Mpart / write fscratch
  bool side_asm_pending = false;     // … and a decomposition's first launch on the eigen stream the summed partials
  double* side_parts = nullptr;      // the summed partials of the latest posterior(…, side), until the next regression …
  const PosteriorEntry* side_parts_entry = nullptr;  // … and the entry they belong to
  void issue_factor(PosteriorEntry& e, PosteriorFactorIO io, double* parts, int splits, hipStream_t side, bool root_here);
  PosteriorEntry* find_entry(const double* theta);
  PosteriorEntry& fresh_entry();
  void alloc_entry(PosteriorEntry& e);
  void prepare_eigen(PosteriorEntry& e, EigenRequest* rq);
  void ensure_eigen(PosteriorEntry& e);  // enqueue on the context's eigen stream (no-op if done or in flight)
  // … or on `es` (eig_stream / eig_stream2) with that stream's work buffer; the caller has made `es` wait for the entry's M
  // part: as launch_posterior_eigen's — 1 issues the chain's head, 2 what follows it and the event behind everything
  void ensure_eigen_on(PosteriorEntry& e, hipStream_t es, int part = 0);
  EigenRequest pending_rq{};
  PosteriorEntry* pending_entry = nullptr;
  DBuf<double> work2;  // eig_stream2's (ranks above 64)
  unsigned eig_flip = 0;
  void await_eigen(PosteriorEntry& e);   // make the context stream wait for it
  void check_status(PosteriorEntry& e);
};

namespace {
// Launches 1-3 of a merged step (proposal -> instance -> searches -> correspondences), enqueued.  icp_chain_step issues
// them itself, or finds them already issued by icp_chain_step_prelaunch for exactly its arguments.
struct StepFront {
  bool valid = false;
  int n_props = 0, generator = -1;
  icp_proposal* props[2] = {nullptr, nullptr};
  std::vector<double> theta_cur, key;  // key: z (generator >= 0) or the proposed state (generator < 0)
  PosteriorEntry* ec[2] = {nullptr, nullptr};
  PosteriorEntry* ep[2] = {nullptr, nullptr};
  StateSlot* s = nullptr;
  bool eigen_first_use = false;
  int parity = 0;  // which half of the pinned coefficient area its first launch writes, and which of the two streams the step uses
  hipStream_t stream = nullptr;
  int splits[2] = {1, 1};                  // launch 4: split count and partial-sum buffers of every posterior
  double* mpart[2] = {nullptr, nullptr};
  int mpart_half[2] = {0, 0};
  int Ksurf = 0;
};
constexpr int kCoeffArea = 512;  // doubles per half of that area (>= kMaxRank)
constexpr int kReduceArea = 16 + 2 * kCoeffArea;  // pinned result of launch 4's likelihood reduction: 8 doubles per parity
}  // namespace

struct icp_evaluator {
  icp_ctx* ctx = nullptr;
  icp_evaluator_params prm{};
  StepFront front;  // pre-launched first half of the next step, if any
  int front_parity = 0;
  // Acceptance estimate of the chain stepped through this evaluator (icp_chain_step): the decomposition of the PROPOSED
  // state is started speculatively unless next to nothing is being accepted.
  std::vector<double> last_prop;  // the state the previous merged step proposed
  double acc_ema = 0.5;
  DBuf<double> target_pts;
  // target-side queries against the CURRENT model surface
  int Kt = 0;              // number of target-side query points (decimated target, or all target vertices for Hausdorff)
  const double* d_tpts = nullptr;
  DBuf<int> hint_tri, hint_nnv, t2m_tri, t2m_nnv;
  DBuf<double> t2m_cp, t2m_d2;
  struct Memo {
    std::vector<double> theta;
    bool valid = false;
    uint64_t stamp = 0;
    double value = 0.0, aux[4] = {0, 0, 0, 0};
    int status = 0;
  } memo[kEvalMemo];
  uint64_t clock = 0;
};

namespace {

template <class F>
int guard(F&& f) {
  try {
    f();
    return ICP_OK;
  } catch (const IcpError& e) {
    g_err = e.msg;
    return e.code;
  } catch (const std::bad_alloc&) {
    g_err = "host out of memory";
    return ICP_ERR_DEVICE;
  } catch (const std::exception& e) {
    g_err = e.what();
    return ICP_ERR_DEVICE;
  }
}

void require(bool ok, const char* msg) {
  if (!ok) fail(ICP_ERR_INVALID_ARG, msg);
}

void check_theta_finite(const icp_ctx* ctx, const double* theta) {
  require(theta != nullptr, "theta is null");
  for (int i = 0; i < 10 + ctx->r; ++i)
    if (!std::isfinite(theta[i])) fail(ICP_ERR_NOT_FINITE, "theta contains a non-finite value");
}

}  // namespace

// ===================================================================== posterior (NonRigidIcpProposal.scala:88-153)

PosteriorEntry* icp_proposal::find_entry(const double* theta) {
  const size_t P = 10 + (size_t)ctx->r;
  // (two states of a chain share their first ten numbers — the pose — more often than not: the last coefficient tells most entries
  // apart before the comparison of the whole vector; compared as bits, like memcmp does)
  uint64_t last;
  std::memcpy(&last, theta + P - 1, sizeof last);
  for (int i = 0; i < kPosteriorMemo; ++i) {
    if (!memo[i].valid) continue;
    uint64_t mine;
    std::memcpy(&mine, memo[i].theta.data() + P - 1, sizeof mine);
    if (mine == last && std::memcmp(memo[i].theta.data(), theta, sizeof(double) * P) == 0) return &memo[i];
  }
  return nullptr;
}

// least recently used memo entry, emptied; the caller fills it and sets `valid`
// device buffers of a memo entry (all entries at proposal creation, see icp_ctx::alloc_slot)
void icp_proposal::alloc_entry(PosteriorEntry& e) {
  if (e.M.p) return;
  const int r = ctx->r, Ka = std::max(K, 1);
  e.id.alloc(Ka); e.aux.alloc(Ka); e.pt.alloc(3 * (size_t)Ka); e.nhat.alloc(3 * (size_t)Ka); e.e.alloc(3 * (size_t)Ka);
  e.keep.alloc(Ka);
  e.coeffs.alloc(r); e.M.alloc((size_t)r * r);
  e.alpha.alloc(r); e.V.alloc((size_t)r * r); e.Vt.alloc((size_t)r * r); e.S.alloc(r);
  e.status_off = 3 * (int)(&e - &memo[0]);
  HIP_OK(hipEventCreateWithFlags(&e.eig_done, hipEventDisableTiming));
}

PosteriorEntry& icp_proposal::fresh_entry() {
  PosteriorEntry* lru = nullptr;
  for (int i = 0; i < kPosteriorMemo; ++i) {
    PosteriorEntry& e = memo[i];
    if (e.reserved) continue;
    if (!lru) { lru = &e; continue; }
    if (!e.valid) { if (lru->valid) lru = &e; }
    else if (lru->valid && e.stamp < lru->stamp) lru = &e;
  }
  if (!lru) fail(ICP_ERR_DEVICE, "internal: every posterior entry is reserved");
  PosteriorEntry& e = *lru;
  alloc_entry(e);
  e.valid = false;
  e.eig_valid = false;
  e.eig_checked = false;
  return e;
}

PosteriorEntry& icp_proposal::posterior(const double* theta, bool want_aux, hipStream_t side) {
  icp_ctx& c = *ctx;
  const int r = c.r;
  const size_t P = 10 + (size_t)r;
  if (side_factor_pending) {
    // a factorisation that went to the side stream ahead of its use (icp_chain_eval_step, pose moves) may still be writing the M and
    // alpha of an entry this call hands out — and it reads the partials and the factor scratch a new posterior would overwrite
    HIP_OK(hipStreamWaitEvent(c.stream, c.ev_side, 0));
    side_factor_pending = false;
  }
  if (PosteriorEntry* hit = find_entry(theta)) {
    PosteriorEntry& e = *hit;
    e.stamp = ++clock;
    if (want_aux && prm.direction == ICP_MODEL_SAMPLING) {
      // diagnostic request for corr_aux on a cached entry: recompute the nearest-vertex ids into it
      StateSlot& s = c.state(theta);
      c.ensure_nnv_prefix(s, K);
      HIP_OK(hipMemcpyAsync(e.aux.p, s.surf_nnv.p, sizeof(int) * K, hipMemcpyDeviceToDevice, c.stream));
    }
    return e;
  }
  // With a side stream the WHOLE posterior goes there — searches, correspondences, regression and (as before) the factorisation —
  // behind the state's instance: the caller's evaluator searches the same state on the context stream at the same time (its own
  // query range, its own scratch set) instead of 65 µs later.  fs / sw: the stream and the scratch set of this posterior's front.
  const hipStream_t fs = side ? side : c.stream;
  const int sw = side ? 1 : 0;
  PosteriorEntry& e = fresh_entry();
  if (e.eig_event_valid && e.eigen_event()) {  // a decomposition that may still read this entry's M (started ahead, its state not kept)
    HIP_OK(hipStreamWaitEvent(fs, e.eigen_event(), 0));
    e.eig_event_valid = false;
  }
  e.theta.assign(theta, theta + P);
  e.valid = true;
  e.stamp = ++clock;
  StateSlot& s = c.state(theta);  // :141 currentMesh
  if (side) {  // (the points, the slot's coefficients: launched or copied on the context stream, possibly just now)
    if (c.ev_inst_slot != (const void*)&s) HIP_OK(hipEventRecord(c.ev_inst, c.stream));  // (else: on record already, ahead of launches that need not be waited for)
    HIP_OK(hipStreamWaitEvent(side, c.ev_inst, 0));
  }
  const EntryInit init{s.coeffs.p, e.coeffs.p, r, status.p + e.status_off};  // (status: {chol, eigen sweeps (diagnostic), eigen})
  if (prm.direction == ICP_TARGET_SAMPLING) {
    // :117-118 nearest vertex of the current mesh for every decimated-target point
    QueryBuffers qb = c.query_scratch(K, c.N, sw);
    launch_vertex_query(fs, c.N, s.x.p, K, target_pts.p, hint_nn.p, qb, nullptr, nn_id.p);
    launch_correspond_target(fs, K, s.x.p, target_pts.p, nn_id.p, c.boundary.p, prm.boundary_aware, s.pose, c.ref.p,
                             c.mean.p, c.tris.p, c.adj_off.p, c.adj.p, e.corr(), init);
  } else {
    // :94-99 closest surface point of the target for model ids 0 until K; nearest target vertex only when the
    // boundary test can change anything (the target has boundary vertices) or the caller asked for it
    c.ensure_surface_prefix(s, K, fs, sw);
    const bool need_nnv = want_aux || (prm.boundary_aware && c.target.n_boundary > 0);
    if (need_nnv) c.ensure_nnv_prefix(s, K, fs, sw);
    launch_correspond_model(fs, K, s.x.p, s.surf_cp.p, need_nnv ? s.surf_nnv.p : nullptr, c.target.boundary.p,
                            prm.boundary_aware, s.pose, c.ref.p, c.mean.p, c.tris.p, c.adj_off.p, c.adj.p, e.corr(), init);
  }
  // :152 interpolatedModel.posterior(uncertainDisplacements)
  const double wt = 1.0 / (prm.tangential_noise * prm.tangential_noise);
  const double kappa = 1.0 / (prm.noise_along_normal * prm.noise_along_normal) - wt;
  int splits = 1;
  if (side_factor_pending) {  // the partials and the factor scratch are still being read / written over there
    if (!side) HIP_OK(hipStreamWaitEvent(c.stream, c.ev_side, 0));  // (on the side stream itself: stream order)
    side_factor_pending = false;
  }
  if (side_asm_pending) {
    HIP_OK(hipStreamWaitEvent(fs, c.ev_asm, 0));
    side_asm_pending = false;
  }
  side_parts = nullptr;
  side_parts_entry = nullptr;
  double* parts = mpart_for_write(0, fs);
  launch_regression(fs, K, r, c.Q.p, e.corr(), wt, kappa, parts, &splits);
  if (side) {  // what the state's slot now holds of this front (surface points, distances, nearest vertices of ids 0..K) is complete
    HIP_OK(hipEventRecord(c.ev_front, side));
    c.front_on_side = true;
    c.front_side_K = prm.direction == ICP_TARGET_SAMPLING ? 0 : K;
  }
  PosteriorFactorIO io{parts, splits, e.M.p, e.alpha.p, status.p + e.status_off, fscratch.p};
  // the Cholesky-root sampler at ranks above 64 (below, k_posterior_root runs where the decomposition would): the factorisation
  // itself hands the factor out — V := L, S := 1/diag(L) — and nothing is decomposed at all
  const bool root_here = sampler == ICP_SAMPLER_CHOLESKY_ROOT && !eigen_speculation_supported(r);
  if (root_here) { io.Lout = e.V.p; io.Sout = e.S.p; }
  issue_factor(e, io, parts, splits, side, root_here);  // (behind the regression in stream order, on either stream)
  return e;
}

// the one-workgroup part of a posterior — sum of the split-K partials, factorisation — on `side` (the caller has made it wait
// for the regression) or on the context stream
void icp_proposal::issue_factor(PosteriorEntry& e, PosteriorFactorIO io, double* parts, int splits, hipStream_t side, bool root_here) {
  icp_ctx& c = *ctx;
  const int r = c.r;
  if (side) {
    launch_sum_partials(side, r, parts, splits);
    io.splits = 1;
    HIP_OK(hipEventRecord(c.ev_sum, side));
    side_parts = parts;
    side_parts_entry = &e;
    launch_posterior_factor(side, r, 1, &io);
    HIP_OK(hipEventRecord(c.ev_side, side));
    side_factor_pending = true;
  } else {
    launch_posterior_factor(c.stream, r, 1, &io);
  }
  if (root_here) {  // "decomposed" as soon as the factorisation is through: an event behind it stands for the basis
    if (!e.eig_done) HIP_OK(hipEventCreateWithFlags(&e.eig_done, hipEventDisableTiming));
    HIP_OK(hipEventRecord(e.eig_done, side ? side : c.stream));
    e.eig_done_shared = nullptr; e.eig_shared_gen = nullptr;
    e.eig_event_valid = true;
    e.done_value = 0;
    e.eig_valid = true;
    e.eig_checked = false;
    h_eig[e.status_off / 3] = 0;
  }
}

void sync_eigen(icp_ctx& c);

double* icp_proposal::mpart_for_write(int half, hipStream_t st) {
  if (PosteriorEntry* rd = mpart_reader[half]) {
    // a cancelled reader (eig_valid withdrawn) may read anything; a finished one has left its status in pinned memory
    // (-1 while in flight): the wait — an API call per step otherwise — is only enqueued for a kept one still at work
    const bool at_work = rd->eig_valid && *(volatile int*)(h_eig + rd->status_off / 3) == -1;
    if (at_work) {
      if (rd->eigen_event()) HIP_OK(hipStreamWaitEvent(st, rd->eigen_event(), 0));
      else sync_eigen(*ctx);  // (no event on record: wait on the host — four steps behind, never seen in practice)
    }
    mpart_reader[half] = nullptr;
  }
  return Mpart.p + (size_t)half * mpart_half_doubles;
}

// the request of the (ordinary) decomposition of `e`; the caller launches it on the context's eigen stream behind an
// ev_ready wait and records e.eig_done
void icp_proposal::prepare_eigen(PosteriorEntry& e, EigenRequest* rq) {
  if (!e.eig_done) HIP_OK(hipEventCreateWithFlags(&e.eig_done, hipEventDisableTiming));
  // the kernel reads all of Vwarm before it writes V, so the two may be the same buffer (a reused memo entry)
  h_eig[e.status_off / 3] = -1;  // in flight; the decomposition stores its status here when it ends
  e.done_value = ++eig_seq;
  // Every decomposition inherits the (tiny) deviation from orthogonality of the basis it starts from and adds that of its own
  // first-order correction (<= 1e-11): every 128th starts cold, from the identity, which puts an end to the accumulation.
  if (((eig_seq + 1) & 127) == 0) warm_valid = false;
  *rq = EigenRequest{e.M.p, warm_valid ? warm_ptr : nullptr, e.V.p, e.Vt.p, e.S.p, work.p, status.p + e.status_off + 2, nullptr,
                     h_eig + e.status_off / 3, eig_words.p + e.status_off / 3, e.done_value, ctx->sqrt_lambda.p};
  rq->root = sampler == ICP_SAMPLER_CHOLESKY_ROOT;
  if (rq->root) rq->Vwarm = nullptr;
  warm_ptr = e.V.p;
  warm_valid = true;
  e.eig_valid = true;
}

// the eigen stream of the batch whose first chain lives on `owner`, out of the launch context's pool
hipStream_t batch_eigen_stream(icp_ctx& lead, const void* owner, int second = 0) {
  // A slot's streams are made when the slot is first handed out — the second one only at ranks above 64, where a wide step's
  // decompositions alternate between two —, not all eight with the first batch: most launch contexts carry one batch at a time, a
  // stream takes 3.3 ms to make, and every stream more makes it likelier that two of them share one of the runtime's hardware queues.
  auto slot = [&](int k) {
    if (!lead.batch_eig[k]) {
      std::lock_guard<std::mutex> lk(g_eig_streams_mu);
      lead.batch_eig[k] = take_stream(lead.device, false, 0);
      g_eig_streams.insert(lead.batch_eig[k]);
      if (lead.r > 64) {  // (the two made one after the other: neighbours among the hardware queues)
        lead.batch_eig2[k] = take_stream(lead.device, false, 0);
        g_eig_streams.insert(lead.batch_eig2[k]);
      }
    }
    return (second && lead.batch_eig2[k]) ? lead.batch_eig2[k] : lead.batch_eig[k];
  };
  for (int k = 0; k < icp_ctx::kBatchRing; ++k)
    if (lead.batch_eig_owner[k] == owner) return slot(k);
  for (int k = 0; k < icp_ctx::kBatchRing; ++k)
    if (!lead.batch_eig_owner[k]) { lead.batch_eig_owner[k] = owner; return slot(k); }
  const int k = (lead.batch_eig_evict = (lead.batch_eig_evict + 1) % icp_ctx::kBatchRing);  // (more batches than the ring holds — not through icp_chain_step_batched_issue, which refuses them: shared)
  lead.batch_eig_owner[k] = owner;
  return slot(k);
}

// waits for every decomposition of this context that may still be running
void sync_eigen(icp_ctx& c) {
  if (c.eig_last && c.eig_last != c.eig_stream) {
    std::lock_guard<std::mutex> lk(g_eig_streams_mu);
    if (g_eig_streams.count(c.eig_last)) HIP_OK(hipStreamSynchronize(c.eig_last));
  }
  if (c.eig_last2 && c.eig_last2 != c.eig_stream2) {
    std::lock_guard<std::mutex> lk(g_eig_streams_mu);
    if (g_eig_streams.count(c.eig_last2)) HIP_OK(hipStreamSynchronize(c.eig_last2));
  }
  HIP_OK(hipStreamSynchronize(c.eig_stream));
  if (c.eig_stream2) HIP_OK(hipStreamSynchronize(c.eig_stream2));
}
// the stream the next decompositions of this context go to (see g_eig_streams)
hipStream_t eigen_stream_for(icp_ctx& c, hipStream_t want) {
  if (c.eig_last && c.eig_last != want) sync_eigen(c);
  c.eig_last = want;
  return want;
}
// the wide step's pair of eigen streams (the second one only carries decompositions that use the proposals' second work buffer)
void eigen_streams_for(icp_ctx& c, hipStream_t e0, hipStream_t e1) {
  if ((c.eig_last && c.eig_last != e0) || (c.eig_last2 && c.eig_last2 != e1)) sync_eigen(c);
  c.eig_last = e0;
  c.eig_last2 = e1;
}

void icp_proposal::ensure_eigen(PosteriorEntry& e) {
  if (e.eig_valid) return;
  icp_ctx& c = *ctx;
  EigenRequest rq;
  prepare_eigen(e, &rq);
  HIP_OK(hipEventRecord(c.ev_ready, c.stream));  // M of this entry may still be in flight on the context stream
  const hipStream_t es = eigen_stream_for(c, c.eig_stream);
  HIP_OK(hipStreamWaitEvent(es, c.ev_ready, 0));
  if (!launch_posterior_eigen_pair(es, c.r, c.sqrt_lambda.p, 1, &rq)) {  // ranks > 64: no completion word
    e.done_value = 0;
    launch_posterior_eigen(es, c.r, rq.M, c.sqrt_lambda.p, rq.Vwarm, rq.V, rq.Vt, rq.S, rq.work, rq.status, nullptr, rq.host_status);
  }
  HIP_OK(hipEventRecord(e.eig_done, es));
  e.eig_done_shared = nullptr;
  e.eig_shared_gen = nullptr;
  e.eig_event_valid = true;
}

void icp_proposal::ensure_eigen_on(PosteriorEntry& e, hipStream_t es, int part) {
  icp_ctx& c = *ctx;
  if (part != 2) {
    if (e.eig_valid) return;
    prepare_eigen(e, &pending_rq);
    if (es == c.eig_stream2) pending_rq.work = work2.p;  // (allocated and zeroed with the proposal: a memset issued here could land in the kernels)
    e.done_value = 0;
    pending_entry = &e;
  } else if (pending_entry != &e) {
    return;  // (part 1 found the basis on record: nothing was started)
  }
  const EigenRequest& rq = pending_rq;
  launch_posterior_eigen(es, c.r, rq.M, c.sqrt_lambda.p, rq.Vwarm, rq.V, rq.Vt, rq.S, rq.work, rq.status, nullptr, rq.host_status, part);
  if (part == 1) return;
  pending_entry = nullptr;
  HIP_OK(hipEventRecord(e.eig_done, es));
  e.eig_done_shared = nullptr;
  e.eig_shared_gen = nullptr;
  e.eig_event_valid = true;
}

void icp_proposal::await_eigen(PosteriorEntry& e) {
  if (e.eigen_event()) HIP_OK(hipStreamWaitEvent(ctx->stream, e.eigen_event(), 0));
  else if (e.eig_valid) sync_eigen(*ctx);  // started by a chain step without an event: wait on the host
}

// ready / ready_seq: the word the regression launch that fills the current half of Mpart raises when it is done — the
// decomposition waits for it on the device (an event between that launch and the next one on the context stream would
// hold the latter back by several µs)
void icp_proposal::speculate_eigen(PosteriorEntry& e, const PosteriorEntry& cur, int splits, int half, const int* ready, int ready_seq,
                                   EigenSpec* spec_out, EigenRequest* rq_out) {
  e.eig_event_valid = false;
  ++spec_seq;
  *spec_out = EigenSpec{splits, h_cancel + (spec_seq & 15), spec_seq, ready, ready_seq, ctx->profiling ? ctx->d_wait_ticks.p + 1 : nullptr};
  // warm start: the basis of the current state's posterior (complete, or ahead of this launch on the same stream)
  const double* warm = (eig_seq & 127) == 127 ? nullptr : (cur.eig_valid ? cur.V.p : (warm_valid ? warm_ptr : nullptr));  // (see prepare_eigen)
  h_eig[e.status_off / 3] = -1;  // in flight
  e.done_value = ++eig_seq;
  *rq_out = EigenRequest{Mpart.p + (size_t)half * mpart_half_doubles, warm, e.V.p, e.Vt.p, e.S.p, work.p, status.p + e.status_off + 2,
                         spec_out, h_eig + e.status_off / 3, eig_words.p + e.status_off / 3, e.done_value};
  rq_out->root = sampler == ICP_SAMPLER_CHOLESKY_ROOT;
  mpart_reader[half] = &e;
  e.eig_valid = true;
  e.eig_checked = false;
  spec_entry = &e;
}

// the caller's next current state decides the fate of the decomposition started for the last proposed state
void icp_proposal::resolve_speculation(const double* theta_cur) {
  if (!spec_entry) return;
  PosteriorEntry& e = *spec_entry;
  spec_entry = nullptr;
  // A decomposition that gave up waiting for its input (k_posterior_eigen_rr) has said so in its pinned status.  That
  // happens when the runtime puts its stream on a hardware queue ahead of the launch it waits for — many streams in one
  // process, or a tool that serialises kernels — and each occurrence stalls the step for the time-out: once is enough.
  if (h_eig[e.status_off / 3] == kEigenGaveUp && !ctx->speculation_off) {
    ctx->speculation_off = true;
    ++ctx->stats.speculation_giveups; ++g_runtime_stats.speculation_giveups;
  }
  const size_t P = 10 + (size_t)ctx->r;
  if (e.valid && e.eig_valid && std::memcmp(e.theta.data(), theta_cur, sizeof(double) * P) == 0) {
    warm_ptr = e.V.p;  // accepted: this is the basis the next decompositions start from
    warm_valid = true;
    return;
  }
  __atomic_store_n(h_cancel + (spec_seq & 15), spec_seq, __ATOMIC_RELEASE);  // rejected (or the entry was recycled meanwhile)
  e.eig_valid = false;
  e.eig_checked = false;
}

// must be called after a synchronising copy of `status` into h_status
void icp_proposal::check_status(PosteriorEntry& e) {
  const int* st = h_status.data() + e.status_off;
  if (st[0]) {
    e.valid = false;
    fail(ICP_ERR_NOT_SPD, "posterior normal equations are not positive definite (non-finite correspondences?)");
  }
  if (st[2]) {
    e.eig_valid = false;
    warm_valid = false;
    fail(ICP_ERR_NOT_FINITE, "posterior eigen-decomposition did not converge");
  }
}

namespace {

void sync_proposal_status(icp_proposal* p) {
  icp_ctx& c = *p->ctx;
  HIP_OK(hipMemcpyAsync(p->h_status.data(), p->status.p, sizeof(int) * 3 * kPosteriorMemo, hipMemcpyDeviceToHost, c.stream));
}

void sync_proposal_status_if(icp_proposal* p, bool needed) {
  if (needed) sync_proposal_status(p);
}

// ===================================================================== evaluators

// enqueue everything logValue(theta) needs; partial results land in d_res[base .. base+8).  In two parts, so that a caller can put a
// posterior of the same state on the side stream between them (icp_chain_eval_step): the searches — with the model ids below
// `reserve` (`reserve_nnv` for their nearest target vertices) left to that posterior's own searches — and the whole target-to-model
// half first; the model-to-target reductions, which read what both streams' searches have written, behind the side stream's event.
void enqueue_eval_searches(icp_evaluator* ev, StateSlot& s, int base, int reserve = 0, int reserve_nnv = 0) {
  icp_ctx& c = *ev->ctx;
  const icp_evaluator_params& p = ev->prm;
  double* out = c.d_res.p + base;
  HIP_OK(hipMemsetAsync(out, 0, sizeof(double) * 8, c.stream));
  const bool m2t = p.kind == ICP_EVAL_HAUSDORFF || p.mode != ICP_TARGET_TO_MODEL;
  const bool t2m = p.kind == ICP_EVAL_HAUSDORFF || p.mode != ICP_MODEL_TO_TARGET;
  const int Km = p.kind == ICP_EVAL_HAUSDORFF ? c.N : p.n_model_ids;
  if (m2t) {
    c.ensure_surface_prefix(s, Km, nullptr, 0, reserve);
    if (p.kind == ICP_EVAL_COLLECTIVE_AVG_HAUSDORFF_BOUNDARY_AWARE && c.target.n_boundary > 0) {
      // (a front already on the side stream that leaves the nearest vertices of its ids to this search: wait for its surface points)
      if (c.front_on_side && s.n_nnv < c.front_side_K && reserve_nnv <= s.n_nnv) {
        HIP_OK(hipStreamWaitEvent(c.stream, c.ev_front, 0));
        c.front_on_side = false;
      }
      c.ensure_nnv_prefix(s, Km, nullptr, 0, reserve_nnv);
    }
  }
  if (t2m) {
    const int Kt = ev->Kt;
    c.ensure_model_spheres(s);
    QueryBuffers qb = c.query_scratch(Kt, c.T);
    launch_surface_query(c.stream, c.T, s.x.p, c.tris.p, s.spheres.p, Kt, ev->d_tpts, ev->hint_tri.p, qb, ev->t2m_cp.p,
                         ev->t2m_d2.p, ev->t2m_tri.p);
    if (p.kind == ICP_EVAL_INDEPENDENT_POINT_DISTANCE) {
      launch_sum_gauss_logpdf(c.stream, Kt, ev->t2m_d2.p, p.gauss_mean, p.gauss_sigma, out + 4);  // :49-54
    } else if (p.kind == ICP_EVAL_HAUSDORFF) {
      launch_dist_max(c.stream, Kt, ev->t2m_d2.p, out + 5);
    } else {
      // Collective…Evaluator.scala:56-60: nearest MODEL-sample vertex of the surface point, tested against the
      // TARGET's boundary flags (sic); ids beyond the target's vertex count count as interior.
      const bool flags = c.target.n_boundary > 0;
      if (flags) {
        QueryBuffers qb2 = c.query_scratch(Kt, c.N);
        launch_vertex_query(c.stream, c.N, s.x.p, Kt, ev->t2m_cp.p, ev->hint_nnv.p, qb2, nullptr, ev->t2m_nnv.p);
      }
      launch_dist_stats(c.stream, Kt, ev->t2m_d2.p, flags ? c.target.boundary.p : nullptr, flags ? ev->t2m_nnv.p : nullptr,
                        c.target.V, out + 4);
    }
  }
}
void enqueue_eval_reductions(icp_evaluator* ev, StateSlot& s, int base) {
  icp_ctx& c = *ev->ctx;
  const icp_evaluator_params& p = ev->prm;
  double* out = c.d_res.p + base;
  // a posterior of this state whose searches ran on the side stream has filled the slot's ids 0..K: the reductions wait for it
  if (c.front_on_side) { HIP_OK(hipStreamWaitEvent(c.stream, c.ev_front, 0)); c.front_on_side = false; }
  const bool m2t = p.kind == ICP_EVAL_HAUSDORFF || p.mode != ICP_TARGET_TO_MODEL;
  const int Km = p.kind == ICP_EVAL_HAUSDORFF ? c.N : p.n_model_ids;
  if (!m2t) return;
  if (s.n_surf < Km) fail(ICP_ERR_DEVICE, "internal: the evaluator's model ids were not all searched");
  if (p.kind == ICP_EVAL_INDEPENDENT_POINT_DISTANCE) {
    launch_sum_gauss_logpdf(c.stream, Km, s.surf_d2.p, p.gauss_mean, p.gauss_sigma, out + 0);  // IndependentPointDistanceEvaluator.scala:40-46
  } else if (p.kind == ICP_EVAL_HAUSDORFF) {
    launch_dist_max(c.stream, Km, s.surf_d2.p, out + 1);  // (finish_eval reads the maxima only: res[1], res[5])
  } else {
    const bool flags = c.target.n_boundary > 0;  // Collective…Evaluator.scala:44-48
    if (flags && s.n_nnv < Km) fail(ICP_ERR_DEVICE, "internal: the evaluator's nearest vertices were not all searched");
    launch_dist_stats(c.stream, Km, s.surf_d2.p, flags ? c.target.boundary.p : nullptr, flags ? s.surf_nnv.p : nullptr,
                      c.target.V, out + 0);
  }
}
void enqueue_eval(icp_evaluator* ev, StateSlot& s, int base) {
  enqueue_eval_searches(ev, s, base);
  enqueue_eval_reductions(ev, s, base);
}

double gauss_logpdf(double x, double mu, double sigma) {  // Breeze Gaussian.logPdf
  double d = (x - mu) / sigma;
  return -d * d / 2.0 - (std::log(std::sqrt(2.0 * M_PI)) + std::log(sigma));
}
double expo_logpdf(double x, double rate) { return -rate * x + std::log(rate); }  // Breeze Exponential.logPdf

// combine the partial reductions exactly as the reference's computeLogValue does
int finish_eval(const icp_evaluator* ev, const double* res, double* value, double* aux) {
  const icp_evaluator_params& p = ev->prm;
  int status = ICP_OK;
  aux[0] = aux[1] = aux[2] = aux[3] = 0.0;
  if (p.kind == ICP_EVAL_INDEPENDENT_POINT_DISTANCE) {
    double m2t = res[0], t2m = res[4];
    aux[0] = m2t; aux[1] = t2m;
    *value = p.mode == ICP_MODEL_TO_TARGET ? m2t : p.mode == ICP_TARGET_TO_MODEL ? t2m : 0.5 * m2t + 0.5 * t2m;  // :60-64
  } else if (p.kind == ICP_EVAL_HAUSDORFF) {
    double hd = std::max(res[1], res[5]);
    aux[0] = hd; aux[1] = res[1]; aux[2] = res[5];
    *value = expo_logpdf(hd, p.exp_rate);  // HausdorffDistanceEvaluator.scala:33-34
  } else {
    double a, h;
    const double a0 = res[0] / res[2], h0 = res[1], a1 = res[4] / res[6], h1 = res[5];
    if (p.mode == ICP_MODEL_TO_TARGET) { a = a0; h = h0; if (res[2] == 0.0) status = ICP_ERR_EMPTY; }
    else if (p.mode == ICP_TARGET_TO_MODEL) { a = a1; h = h1; if (res[6] == 0.0) status = ICP_ERR_EMPTY; }
    else {
      a = 0.5 * a0 + 0.5 * a1; h = std::max(h0, h1);  // :71-75
      if (res[2] == 0.0 || res[6] == 0.0) status = ICP_ERR_EMPTY;
    }
    aux[0] = a; aux[1] = h; aux[2] = res[2]; aux[3] = res[6];
    *value = gauss_logpdf(a, p.gauss_mean, p.gauss_sigma) + expo_logpdf(h, p.exp_rate);  // :77
  }
  if (status == ICP_OK && std::isnan(*value)) status = ICP_ERR_NOT_FINITE;
  return status;
}

icp_evaluator::Memo* eval_lookup(icp_evaluator* ev, const double* theta) {
  const size_t P = 10 + (size_t)ev->ctx->r;
  for (auto& m : ev->memo)
    if (m.valid && std::memcmp(m.theta.data(), theta, sizeof(double) * P) == 0) {
      m.stamp = ++ev->clock;
      return &m;
    }
  return nullptr;
}

icp_evaluator::Memo* eval_store(icp_evaluator* ev, const double* theta) {
  const size_t P = 10 + (size_t)ev->ctx->r;
  icp_evaluator::Memo* lru = &ev->memo[0];
  for (auto& m : ev->memo) {
    if (!m.valid) { lru = &m; break; }
    if (m.stamp < lru->stamp) lru = &m;
  }
  lru->theta.assign(theta, theta + P);
  lru->valid = true;
  lru->stamp = ++ev->clock;
  return lru;
}

bool pose_equal(const double* a, const double* b) {  // NonRigidIcpProposal.scala:72: everything but the shape must match
  for (int i = 0; i < 10; ++i)
    if (a[i] != b[i]) return false;
  return true;
}

}  // namespace

namespace {
// the target's immutable device data (vertices, triangles, boundary flags, bounding spheres in patch order): shared by every context of
// the device made from the same arrays (g_shared_targets).  The caller holds g_shared_mu and has bound the device.
void attach_target(icp_ctx* ctx, const icp_mesh_desc* target, int device) {
  uint64_t th = hash_words(0x5678, target->points, sizeof(double) * 3 * (size_t)target->n_points);
  th = hash_words(th, target->triangles, sizeof(int32_t) * 3 * (size_t)target->n_triangles);
  const SharedKey tkey{device, target->n_points, target->n_triangles, 0, th};
  // (a batch registration attaches one target after the other: the entries of targets nobody holds any more are dropped)
  for (auto it = g_shared_targets.begin(); it != g_shared_targets.end();)
    it = it->second.expired() ? g_shared_targets.erase(it) : std::next(it);
  std::shared_ptr<SharedTarget> stg = g_shared_targets[tkey].lock();
  if (!stg) {
    stg = std::make_shared<SharedTarget>();
    DeviceMesh& tg = stg->mesh;
    tg.V = target->n_points; tg.T = target->n_triangles;
    std::vector<uint8_t> tb;
    boundary_flags(tg.V, tg.T, target->triangles, tb);
    tg.n_boundary = (int)std::count(tb.begin(), tb.end(), (uint8_t)1);
    tg.verts.upload(target->points, (size_t)3 * tg.V);
    tg.tris.upload(target->triangles, (size_t)3 * tg.T);
    tg.boundary.upload(tb.data(), tb.size());
    tg.spheres.alloc(sphere_floats4(tg.T));
    {
      const std::vector<int> order = coherent_triangle_order(tg.V, tg.T, target->points, target->triangles);
      tg.tri_order.upload(order.data(), order.size());
    }
    launch_tri_spheres(ctx->stream, tg.T, tg.verts.p, tg.tris.p, tg.tri_order.p, tg.spheres.p);
    HIP_OK(hipStreamSynchronize(ctx->stream));  // (complete before another context may find it)
    g_shared_targets[tkey] = stg;
  }
  ctx->shared_target = stg;
  DeviceMesh& tg = ctx->target;
  const DeviceMesh& o = stg->mesh;
  tg.V = o.V; tg.T = o.T; tg.n_boundary = o.n_boundary;
  tg.verts.alias(o.verts); tg.tris.alias(o.tris); tg.tri_order.alias(o.tri_order); tg.spheres.alias(o.spheres); tg.boundary.alias(o.boundary);
}
}  // namespace

// ===================================================================== C ABI

extern "C" {

const char* icp_status_string(int status) {
  switch (status) {
    case ICP_OK: return "ok";
    case ICP_ERR_INVALID_ARG: return "invalid argument";
    case ICP_ERR_DEVICE: return "HIP device error";
    case ICP_ERR_NOT_FINITE: return "non-finite result";
    case ICP_ERR_NOT_SPD: return "matrix not positive definite";
    case ICP_ERR_EMPTY: return "no points left after the boundary filter";
    case ICP_ERR_BUSY: return "context busy: part of a batch in flight";
    default: return "unknown status";
  }
}

const char* icp_last_error(void) { return g_err.c_str(); }

int icp_ctx_rank(const icp_ctx* ctx) { return ctx ? ctx->r : ICP_ERR_INVALID_ARG; }
int icp_ctx_device(const icp_ctx* ctx) { return ctx ? ctx->device : ICP_ERR_INVALID_ARG; }

int icp_ctx_create(const icp_model_desc* model, const icp_mesh_desc* target, int device, icp_ctx** out) {
  if (out) *out = nullptr;
  icp_ctx* ctx = nullptr;
  int rc = guard([&] {
    require(model && target && out, "null argument");
    require(model->n_points > 0 && model->n_triangles >= 0 && model->rank > 0 && model->rank <= kMaxRank,
            "model sizes out of range (rank must be in [1,500])");
    require(model->ref_points && model->basis && model->variance && (model->triangles || model->n_triangles == 0),
            "model arrays missing");
    require(target->n_points > 0 && target->n_triangles >= 0 && target->points &&
                (target->triangles || target->n_triangles == 0),
            "target arrays missing");
    const int N = model->n_points, T = model->n_triangles, r = model->rank;
    check_triangles(N, T, model->triangles, "model");
    check_triangles(target->n_points, target->n_triangles, target->triangles, "target");
    for (int j = 0; j < r; ++j) require(model->variance[j] > 0.0 && std::isfinite(model->variance[j]), "variance must be positive");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
      fail(ICP_ERR_DEVICE, std::string("no usable HIP device (this library has no CPU fallback): ") + hipGetErrorString(e));
    if (device < 0) {
      const char* lr = std::getenv("LOCAL_RANK");
      device = lr ? std::atoi(lr) % ndev : 0;
    }
    require(device < ndev, "device ordinal out of range");

    ctx = new icp_ctx();
    ctx->device = device;
    ctx->N = N; ctx->T = T; ctx->r = r;
    ctx->bind();
    // The runtime multiplexes streams onto a small pool of hardware queues PER PRIORITY (four by default), and two
    // streams on one hardware queue run one kernel at a time: with other streams alive in the process (torch's,
    // RCCL's: default priority) the two step streams ended up sharing a queue and a step cost 15 % more (measured under
    // torch.distributed.run).  The streams of the first context of a process — the one-chain-per-GPU layout — are
    // therefore created at the greatest priority: a pool of their own.  Further contexts (several chains on one GPU
    // from one process) take the default priority: with every stream in the greatest-priority pool their aggregate
    // rate fell from 22k to 15k it/s (tools/multichain.py, 4-16 contexts).
    int prio_least = 0, prio_greatest = 0;
    HIP_OK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    if (g_live_contexts.load(std::memory_order_relaxed) > 0) prio_greatest = 0;
    if (const char* sp = dev_env("ICP_STREAM_PRIORITY")) {  // A/B switch: 0 = default priority everywhere
      if (std::atoi(sp) == 0) prio_greatest = 0;
    }
    // (out of the pool of streams of destroyed contexts where it has any of that priority class: take_stream)
    const bool greatest = prio_greatest != 0;
    ctx->stream = take_stream(device, greatest, prio_greatest);
    ctx->front_stream = take_stream(device, greatest, prio_greatest);
    ctx->eig_stream = take_stream(device, greatest, prio_greatest);
    // (ranks above 64 only: a stream costs a few MB of the runtime's own memory; created HERE, next to its sibling, and not on first
    // use: the runtime maps streams to its hardware queues in creation order, and a latecomer shared one with the context stream)
    if (ctx->r > 64) ctx->eig_stream2 = take_stream(device, greatest, prio_greatest);
    { std::lock_guard<std::mutex> lk(g_eig_streams_mu); g_eig_streams.insert(ctx->eig_stream); }
    HIP_OK(hipEventCreateWithFlags(&ctx->ev_ready, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ctx->ev_side, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ctx->ev_sum, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ctx->ev_asm, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ctx->ev_inst, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ctx->ev_front, hipEventDisableTiming));
    pinned_alloc((void**)&ctx->h_wait_error, sizeof(int) * 16);
    ctx->h_wait_error[0] = 0;

    // ---- model and target: the immutable device data is shared between the contexts of a device made from the same arrays
    std::lock_guard<std::mutex> shared_lk(g_shared_mu);
    uint64_t mh = hash_words(0x1234, model->ref_points, sizeof(double) * 3 * N);
    mh = hash_words(mh, model->basis, sizeof(double) * 3 * N * r);
    mh = hash_words(mh, model->variance, sizeof(double) * r);
    if (model->mean_deformation) mh = hash_words(mh, model->mean_deformation, sizeof(double) * 3 * N);
    mh = hash_words(mh, model->triangles, sizeof(int32_t) * 3 * T);
    const SharedKey mkey{device, N, T, r, mh};
    std::shared_ptr<SharedModel> sm = g_shared_models[mkey].lock();
    if (!sm) {
      sm = std::make_shared<SharedModel>();
      sm->device = device;
      // Q = Φ·diag(√λ) in two layouts, Gram matrix G = QᵀQ and chol(G + σ²I) (one-off host work)
      std::vector<double> Q((size_t)3 * N * r), Qp((size_t)3 * N * r), sl(r), isl(r);
      for (int j = 0; j < r; ++j) { sl[j] = std::sqrt(model->variance[j]); isl[j] = 1.0 / sl[j]; }
      for (size_t row = 0; row < (size_t)3 * N; ++row)
        for (int j = 0; j < r; ++j) {
          double q = model->basis[row * r + j] * sl[j];
          Q[row * r + j] = q;
          size_t i = row / 3, d = row % 3;
          Qp[((size_t)j * 3 + d) * N + i] = q;
        }
      std::vector<double> G((size_t)r * r, 0.0);
      for (size_t row = 0; row < (size_t)3 * N; ++row) {
        const double* q = &Q[row * r];
        for (int a = 0; a < r; ++a) {
          double qa = q[a];
          double* g = &G[(size_t)a * r];
          for (int b = 0; b <= a; ++b) g[b] += qa * q[b];
        }
      }
      for (int a = 0; a < r; ++a)
        for (int b = a + 1; b < r; ++b) G[(size_t)a * r + b] = G[(size_t)b * r + a];
      std::vector<double> Gs = G, Ginv, Pinv;
      for (int a = 0; a < r; ++a) Gs[(size_t)a * r + a] += kSigma2;
      if (!host_spd_inverse(r, Gs, Pinv)) fail(ICP_ERR_NOT_SPD, "Q^T Q + sigma^2 I is not positive definite");
      if (!host_spd_inverse(r, G, Ginv)) fail(ICP_ERR_NOT_SPD, "Q^T Q is not positive definite (linearly dependent basis functions)");
      std::vector<double> mean((size_t)3 * N, 0.0);
      if (model->mean_deformation) std::memcpy(mean.data(), model->mean_deformation, sizeof(double) * 3 * N);
      std::vector<uint8_t> mb;
      boundary_flags(N, T, model->triangles, mb);
      std::vector<int> off, adj;
      vertex_adjacency(N, T, model->triangles, off, adj);
      sm->n_boundary = (int)std::count(mb.begin(), mb.end(), (uint8_t)1);
      sm->ref.upload(model->ref_points, (size_t)3 * N);
      sm->mean.upload(mean.data(), mean.size());
      sm->Q.upload(Q.data(), Q.size());
      sm->Qp.upload(Qp.data(), Qp.size());
      sm->sqrt_lambda.upload(sl.data(), r);
      sm->inv_sqrt_lambda.upload(isl.data(), r);
      sm->G.upload(G.data(), G.size());
      sm->Ginv.upload(Ginv.data(), Ginv.size());
      sm->P.upload(Pinv.data(), Pinv.size());
      sm->tris.upload(model->triangles, (size_t)3 * T);
      {
        const std::vector<int> order = coherent_triangle_order(N, T, model->ref_points, model->triangles);
        sm->tri_order.upload(order.data(), order.size());
      }
      sm->adj_off.upload(off.data(), off.size());
      sm->adj.upload(adj.data(), adj.size());
      sm->boundary.upload(mb.data(), mb.size());
      g_shared_models[mkey] = sm;
    }
    ctx->shared_model = sm;
    if (g_model_keep[0] != sm && g_model_keep[1] != sm) {
      std::shared_ptr<SharedModel> evicted = std::move(g_model_keep[g_model_keep_next]);
      g_model_keep[g_model_keep_next] = sm;
      g_model_keep_next ^= 1;
      if (evicted && evicted.use_count() == 1 && evicted->device != device) {  // its last owner: freed under its own device
        (void)hipSetDevice(evicted->device);
        evicted.reset();
        (void)hipSetDevice(device);
      }
    }
    ctx->n_boundary = sm->n_boundary;
    ctx->ref.alias(sm->ref); ctx->mean.alias(sm->mean); ctx->Q.alias(sm->Q); ctx->Qp.alias(sm->Qp);
    ctx->sqrt_lambda.alias(sm->sqrt_lambda); ctx->inv_sqrt_lambda.alias(sm->inv_sqrt_lambda);
    ctx->G.alias(sm->G); ctx->Ginv.alias(sm->Ginv); ctx->P.alias(sm->P);
    ctx->tris.alias(sm->tris); ctx->tri_order.alias(sm->tri_order); ctx->adj_off.alias(sm->adj_off); ctx->adj.alias(sm->adj);
    ctx->boundary.alias(sm->boundary);

    attach_target(ctx, target, device);

    ctx->hint_surf.alloc(N); ctx->hint_surf.fill_bytes(0xFF);
    ctx->hint_nnv.alloc(N); ctx->hint_nnv.fill_bytes(0xFF);
    ctx->stage_cap = 64 * (size_t)(10 + r) + 4096;
    pinned_alloc((void**)&ctx->h_stage, sizeof(double) * ctx->stage_cap);
    ctx->d_stage.alloc(ctx->stage_cap);
    const size_t res_cap = std::max<size_t>(2048, 3 * (size_t)N + 64);
    pinned_alloc((void**)&ctx->h_out, sizeof(double) * (icp_ctx::kStatusDoubles + res_cap));
    ctx->d_out.alloc(icp_ctx::kStatusDoubles + res_cap);
    ctx->h_status = (int*)ctx->h_out;
    ctx->h_res = ctx->h_out + icp_ctx::kStatusDoubles;
    ctx->d_status.p = (int*)ctx->d_out.p; ctx->d_status.n = 2 * icp_ctx::kStatusDoubles; ctx->d_status.owned = false;
    ctx->d_res.p = ctx->d_out.p + icp_ctx::kStatusDoubles; ctx->d_res.n = res_cap; ctx->d_res.owned = false;
    pinned_alloc((void**)&ctx->h_flag, sizeof(int) * 16);
    ctx->h_flag[0] = 0;
    ctx->d_done.alloc(4);
    ctx->d_done.fill_bytes(0);
    ctx->d_wait_ticks.alloc(2);
    ctx->d_wait_ticks.fill_bytes(0);
    for (auto& sl : ctx->slots) ctx->alloc_slot(sl);
    HIP_OK(hipStreamSynchronize(ctx->stream));
    ++g_live_contexts;
    ctx->counted = true;
    *out = ctx;
  });
  if (rc != ICP_OK && ctx) {
    icp_ctx_destroy(ctx);
  }
  return rc;
}

void icp_release_cached_models(void) {
  std::lock_guard<std::mutex> lk(g_shared_mu);
  int caller_device = -1;  // (a model's buffers are freed under ITS device; the caller's current device is put back afterwards)
  const bool have_device = hipGetDevice(&caller_device) == hipSuccess;
  for (int i = 0; i < 2; ++i) {
    if (g_model_keep[i]) (void)hipSetDevice(g_model_keep[i]->device);
    g_model_keep[i].reset();
  }
  if (have_device) (void)hipSetDevice(caller_device);
  drain_pools();
}

void icp_ctx_destroy(icp_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  DeviceQuiesce _q;  // (its device buffers go back to the pool: device_free)
  if (ctx->eig_last && ctx->eig_last != ctx->eig_stream) {  // decompositions of this context on a batch's stream
    std::lock_guard<std::mutex> lk(g_eig_streams_mu);
    if (g_eig_streams.count(ctx->eig_last)) (void)hipStreamSynchronize(ctx->eig_last);
  }
  if (ctx->eig_last2 && ctx->eig_last2 != ctx->eig_stream2) {
    std::lock_guard<std::mutex> lk(g_eig_streams_mu);
    if (g_eig_streams.count(ctx->eig_last2)) (void)hipStreamSynchronize(ctx->eig_last2);
  }
  for (hipStream_t* pool : {ctx->batch_eig, ctx->batch_eig2})
    for (int k = 0; k < icp_ctx::kBatchRing; ++k)
      if (pool[k]) {
        std::lock_guard<std::mutex> lk(g_eig_streams_mu);
        g_eig_streams.erase(pool[k]);
        give_stream(pool[k]);
        pool[k] = nullptr;
      }
  if (ctx->eig_stream2) {
    give_stream(ctx->eig_stream2);
  }
  if (ctx->eig_stream) {
    std::lock_guard<std::mutex> lk(g_eig_streams_mu);
    g_eig_streams.erase(ctx->eig_stream);
    (void)hipStreamSynchronize(ctx->eig_stream);
    library_release_stream(ctx->eig_stream);
    give_stream(ctx->eig_stream);
  }
  if (ctx->ev_ready) (void)hipEventDestroy(ctx->ev_ready);
  if (ctx->ev_side) (void)hipEventDestroy(ctx->ev_side);
  if (ctx->ev_sum) (void)hipEventDestroy(ctx->ev_sum);
  if (ctx->ev_asm) (void)hipEventDestroy(ctx->ev_asm);
  if (ctx->front_stream) {
    (void)hipStreamSynchronize(ctx->front_stream);
    library_release_stream(ctx->front_stream);
    give_stream(ctx->front_stream);
  }
  if (ctx->stream) {
    (void)hipStreamSynchronize(ctx->stream);
    library_release_stream(ctx->stream);
    give_stream(ctx->stream);
  }
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->ev_inst) (void)hipEventDestroy(ctx->ev_inst);
  if (ctx->ev_front) (void)hipEventDestroy(ctx->ev_front);
  if (ctx->h_wait_error) pinned_free(ctx->h_wait_error);
  if (ctx->h_wide_z) pinned_free(ctx->h_wide_z);
  for (void* bp : ctx->wide_pinned)
    if (bp) pinned_free(bp);
  for (hipEvent_t ev : ctx->ev_wide_sum)
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : ctx->ev_wide_fac)
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : ctx->ev_wide_head)
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : ctx->ev_wide_eval)
    if (ev) (void)hipEventDestroy(ev);
  if (ctx->h_gate_error) pinned_free(ctx->h_gate_error);
  for (void* bp : ctx->batch_eig_rec)
    if (bp) pinned_free(bp);
  g_host_timing.report();
  g_batch_timing.report();
  for (auto& r : ctx->prof.pool) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  if (ctx->h_stage) pinned_free(ctx->h_stage);
  if (ctx->h_out) pinned_free(ctx->h_out);
  if (ctx->h_flag) pinned_free(ctx->h_flag);
  for (void* bp : ctx->batch_pinned)
    if (bp) pinned_free(bp);
  if (ctx->counted) --g_live_contexts;
  delete ctx;
}

int icp_ctx_set_target(icp_ctx* ctx, const icp_mesh_desc* target) {
  return guard([&] {
    require(ctx && target, "null argument");
    require(target->n_points > 0 && target->n_triangles >= 0 && target->points && (target->triangles || target->n_triangles == 0),
            "target arrays missing");
    check_triangles(target->n_points, target->n_triangles, target->triangles, "target");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    require(ctx->proposals.empty() && ctx->evaluators.empty(), "the context still has proposals or evaluators made for its present target");
    Bound _b(ctx);
    HIP_OK(hipStreamSynchronize(ctx->stream));
    HIP_OK(hipStreamSynchronize(ctx->front_stream));
    sync_eigen(*ctx);
    {
      std::lock_guard<std::mutex> shared_lk(g_shared_mu);
      attach_target(ctx, target, ctx->device);
    }
    // what was cached against the old target: the states' surface points and nearest vertices, the search hints
    for (auto& s : ctx->slots) { s.valid = false; s.defo_valid = false; s.spheres_valid = false; s.n_surf = s.n_nnv = 0; s.lo_surf = s.hi_surf = s.lo_nnv = s.hi_nnv = 0; }
    HIP_OK(hipMemsetAsync(ctx->hint_surf.p, 0xFF, sizeof(int) * ctx->N, ctx->stream));
    HIP_OK(hipMemsetAsync(ctx->hint_nnv.p, 0xFF, sizeof(int) * ctx->N, ctx->stream));
    HIP_OK(hipStreamSynchronize(ctx->stream));
    ctx->stage_used = 0;
  });
}

namespace { void release_front(StepFront& F); }

int icp_ctx_set_rotation(icp_ctx* ctx, const double* angles, const double* R) {
  return guard([&] {
    require(ctx && angles, "null argument");
    for (int k = 0; k < 3; ++k) require(std::isfinite(angles[k]), "angles must be finite");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    icp_ctx::RotationEntry* slot = nullptr;
    for (auto& e : ctx->rotations)
      if (e.valid && e.angles[0] == angles[0] && e.angles[1] == angles[1] && e.angles[2] == angles[2]) { slot = &e; break; }
    // Everything cached under a theta with these angles was posed with the matrix in force so far: whenever that changes — a first
    // registration (the library's own convention until now), a replacement, a withdrawal, an eviction — the state slots, the
    // posterior memo entries of every proposal and the evaluators' memoised values of such thetas are dropped.  (Pending half steps
    // and decompositions in flight are drained first: they hold such entries.)
    auto forget = [&](const double* a) {
      auto same = [&](const std::vector<double>& th) { return th.size() >= 7 && th[4] == a[0] && th[5] == a[1] && th[6] == a[2]; };
      bool any = false;
      for (auto& sl : ctx->slots) any = any || (sl.valid && same(sl.theta));
      for (icp_proposal* p : ctx->proposals)
        for (int i = 0; i < kPosteriorMemo; ++i) any = any || (p->memo[i].valid && same(p->memo[i].theta));
      for (icp_evaluator* ev : ctx->evaluators) {
        for (auto& m : ev->memo) any = any || (m.valid && same(m.theta));
        any = any || (ev->front.valid && same(ev->front.theta_cur));
      }
      if (!any) return;
      if (ctx->batch_busy) throw IcpError{ICP_ERR_BUSY, "the context belongs to a batch in flight"};
      ctx->bind();
      HIP_OK(hipStreamSynchronize(ctx->stream));
      HIP_OK(hipStreamSynchronize(ctx->front_stream));
      sync_eigen(*ctx);
      for (icp_evaluator* ev : ctx->evaluators) {
        if (ev->front.valid) release_front(ev->front);
        for (auto& m : ev->memo)
          if (m.valid && same(m.theta)) m.valid = false;
        ev->last_prop.clear();
      }
      for (auto& sl : ctx->slots)
        if (sl.valid && same(sl.theta)) sl.valid = false;
      for (icp_proposal* p : ctx->proposals) {
        for (int i = 0; i < kPosteriorMemo; ++i) {
          PosteriorEntry& en = p->memo[i];
          if (en.valid && same(en.theta)) { en.valid = false; en.eig_valid = false; en.eig_checked = false; }
        }
        p->spec_entry = nullptr;
      }
    };
    if (!R) {  // withdraw the entry
      if (slot) { slot->valid = false; forget(angles); }
      return;
    }
    // the matrix must be a rotation (orthonormal to 1e-9, determinant +1): a wrong layout would otherwise pass silently
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        double d = 0.0;
        for (int k = 0; k < 3; ++k) d += R[3 * a + k] * R[3 * b + k];
        require(std::fabs(d - (a == b ? 1.0 : 0.0)) <= 1e-9, "R is not orthonormal (row-major 3x3 rotation expected)");
      }
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
    require(det > 0.0, "R is a reflection, not a rotation");
    {  // the convention check: does the caller's matrix agree with the library's Rz·Ry·Rx for these angles?
      double own[9];
      icp_rotation_matrix(angles[0], angles[1], angles[2], own);
      double dmax = 0.0;
      for (int k = 0; k < 9; ++k) dmax = std::max(dmax, std::fabs(own[k] - R[k]));
      if (dmax <= icp_ctx::kRotationTol) ++ctx->rotations_verified; else ++ctx->rotations_mismatched;
    }
    if (slot) {  // replacement: only if the matrix really differs
      bool differs = false;
      for (int k = 0; k < 9; ++k) differs = differs || slot->R[k] != R[k];
      if (differs) forget(angles);
    } else {
      slot = &ctx->rotations[0];
      for (auto& e : ctx->rotations) {
        if (!e.valid) { slot = &e; break; }
        if (e.stamp < slot->stamp) slot = &e;
      }
      if (slot->valid) forget(slot->angles);  // eviction: that triple falls back to the library's convention
      forget(angles);
    }
    for (int k = 0; k < 3; ++k) slot->angles[k] = angles[k];
    for (int k = 0; k < 9; ++k) slot->R[k] = R[k];
    slot->valid = true;
    slot->stamp = ++ctx->rotation_clock;
  });
}

int icp_ctx_rotation_convention(icp_ctx* ctx, int64_t* verified, int64_t* mismatched) {
  if (!ctx) return ICP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  if (verified) *verified = ctx->rotations_verified;
  if (mismatched) *mismatched = ctx->rotations_mismatched;
  return ICP_OK;
}

int icp_ctx_runtime_stats(const icp_ctx* ctx, icp_runtime_stats* out) {
  if (!out) return ICP_ERR_INVALID_ARG;
  const RuntimeStats& s = ctx ? ctx->stats : g_runtime_stats;
  std::memset(out, 0, sizeof(*out));
  out->wait_timeouts = s.wait_timeouts.load();
  out->speculation_giveups = s.speculation_giveups.load();
  out->pipeline_fallbacks = s.pipeline_fallbacks.load();
  out->step_redos = s.step_redos.load();
  out->gate_timeouts = s.gate_timeouts.load();
  return ICP_OK;
}

int icp_ctx_step_paths(const icp_ctx* ctx, int64_t* out) {
  if (!out) return ICP_ERR_INVALID_ARG;
  const StepPaths& p = ctx ? ctx->paths : g_step_paths;
  for (int k = 0; k < 4; ++k) out[k] = p.n[k].load(std::memory_order_relaxed);
  return ICP_OK;
}

int icp_ctx_set_idle_hook(icp_ctx* ctx, icp_idle_fn fn, void* arg) {
  if (!ctx) return ICP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  ctx->idle_fn = fn;
  ctx->idle_arg = fn ? arg : nullptr;
  return ICP_OK;
}

int icp_ctx_profile_start(icp_ctx* ctx, int32_t max_launches) {
  return guard([&] {
    require(ctx && max_launches > 0, "bad argument");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    ctx->bind();
    HIP_OK(hipStreamSynchronize(ctx->stream));
    while (ctx->prof.pool.size() < (size_t)max_launches) {
      Profiler::Rec r;
      HIP_OK(hipEventCreate(&r.a));
      HIP_OK(hipEventCreate(&r.b));
      r.id = 0;
      ctx->prof.pool.push_back(r);
    }
    ctx->prof.used = 0;
    ctx->prof.overflow = false;
    ctx->d_wait_ticks.fill_bytes(0);
    if (!ctx->d_search_counters.p) ctx->d_search_counters.alloc(kSearchCounters);
    ctx->d_search_counters.fill_bytes(0);
    ctx->prof.counters = ctx->count_searches ? ctx->d_search_counters.p : nullptr;
    ctx->profiling = true;
  });
}

int icp_ctx_profile_search_counters(icp_ctx* ctx, int32_t on) {
  if (!ctx) return ICP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  ctx->count_searches = on != 0;
  return ICP_OK;
}

int icp_ctx_profile_stop(icp_ctx* ctx, icp_kernel_stat* stats, int32_t capacity, int32_t* n_out) {
  return guard([&] {
    require(ctx && stats && n_out && capacity >= KID_COUNT, "bad argument (capacity must be >= 32)");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    ctx->bind();
    HIP_OK(hipStreamSynchronize(ctx->stream));
    HIP_OK(hipStreamSynchronize(ctx->front_stream));
    sync_eigen(*ctx);
    for (hipStream_t bs : ctx->batch_eig)
      if (bs) HIP_OK(hipStreamSynchronize(bs));  // (decompositions of batches this context carried)
    for (hipStream_t bs : ctx->batch_eig2)
      if (bs) HIP_OK(hipStreamSynchronize(bs));
    ctx->profiling = false;
    std::vector<icp_kernel_stat> acc(KID_COUNT);
    for (int i = 0; i < KID_COUNT; ++i) {
      std::memset(&acc[i], 0, sizeof(icp_kernel_stat));
      std::strncpy(acc[i].name, kKernelNames[i], sizeof(acc[i].name) - 1);
      acc[i].min_ms = 1e300;
    }
    for (size_t i = 0; i < ctx->prof.used; ++i) {
      float ms = 0.f;
      HIP_OK(hipEventElapsedTime(&ms, ctx->prof.pool[i].a, ctx->prof.pool[i].b));
      icp_kernel_stat& a = acc[ctx->prof.pool[i].id];
      a.calls++;
      a.total_ms += ms;
      a.min_ms = std::min(a.min_ms, (double)ms);
      a.max_ms = std::max(a.max_ms, (double)ms);
    }
    int n = 0;
    for (int i = 0; i < KID_COUNT; ++i)
      if (acc[i].calls > 0) stats[n++] = acc[i];
    {  // how much of k_step_begin's time was spent waiting ON THE DEVICE for the previous step / the decomposition it draws from
      long long both[2] = {0, 0};
      HIP_OK(hipMemcpy(both, ctx->d_wait_ticks.p, sizeof(both), hipMemcpyDeviceToHost));
      const long long ticks = both[0];
      if (both[1] > 0 && n < capacity) {  // … and of the speculative decompositions' time waiting for their input (EigenSpec::wait_ticks)
        icp_kernel_stat w;
        std::memset(&w, 0, sizeof(w));
        std::strncpy(w.name, "k_posterior_eigen.device_wait", sizeof(w.name) - 1);
        w.calls = acc[KID_EIGEN].calls;
        w.total_ms = (double)both[1] * 1e-5;
        stats[n++] = w;
      }
      if (ticks > 0 && n < capacity) {
        icp_kernel_stat w;
        std::memset(&w, 0, sizeof(w));
        std::strncpy(w.name, "k_step_begin.device_wait", sizeof(w.name) - 1);
        w.calls = acc[KID_STEP_BEGIN].calls;
        w.total_ms = (double)ticks * 1e-5;  // 100 MHz ticks
        stats[n++] = w;
      }
    }
    {  // executed tests of the searches (counted per wave while profiling): rows "count.*", the number in `calls`
      unsigned long long cnt[kSearchCounters] = {};
      if (ctx->d_search_counters.p && ctx->prof.counters) HIP_OK(hipMemcpy(cnt, ctx->d_search_counters.p, sizeof(cnt), hipMemcpyDeviceToHost));
      static const char* names[5] = {"count.surface_ball_tests", "count.surface_sphere_tests", "count.surface_exact_tests",
                                     "count.vertex_filter_tests", "count.vertex_exact_tests"};
      for (int k = 0; k < 5; ++k)
        if (cnt[k] > 0 && n < capacity) {
          icp_kernel_stat w;
          std::memset(&w, 0, sizeof(w));
          std::strncpy(w.name, names[k], sizeof(w.name) - 1);
          w.calls = (int64_t)cnt[k];
          stats[n++] = w;
        }
    }
    *n_out = n;
    if (ctx->prof.overflow) fail(ICP_ERR_INVALID_ARG, "profiler event pool too small: raise max_launches");
  });
}

int icp_transformed_mesh(icp_ctx* ctx, const double* theta, double* points_out) {
  return guard([&] {
    require(ctx && points_out, "null argument");
    check_theta_finite(ctx, theta);
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    Bound _b(ctx);
    StateSlot& s = ctx->state(theta);
    HIP_OK(hipMemcpyAsync(points_out, s.x.p, sizeof(double) * 3 * ctx->N, hipMemcpyDeviceToHost, ctx->stream));
    ctx->finish(0, 0);
  });
}

int icp_vertex_normals(icp_ctx* ctx, const double* theta, double* normals_out) {
  return guard([&] {
    require(ctx && normals_out, "null argument");
    check_theta_finite(ctx, theta);
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    Bound _b(ctx);
    StateSlot& s = ctx->state(theta);
    DBuf<double> nrm;
    nrm.alloc(3 * (size_t)ctx->N);
    launch_vertex_normals(ctx->stream, ctx->N, s.x.p, ctx->tris.p, ctx->adj_off.p, ctx->adj.p, nrm.p);
    HIP_OK(hipMemcpyAsync(normals_out, nrm.p, sizeof(double) * 3 * ctx->N, hipMemcpyDeviceToHost, ctx->stream));
    ctx->finish(0, 0);
  });
}

namespace {
// shared body of the four stand-alone search entry points
void run_search(icp_ctx* ctx, bool surface, int V, int T, const double* verts, const int* tris, const float4* spheres,
                int32_t n, const double* queries, double* points_out, int32_t* index_out, double* dist2_out) {
  require(n >= 0 && (queries || n == 0), "bad query array");
  if (n == 0) return;
  DBuf<double> q, cp, d2;
  DBuf<int> idx;
  q.upload(queries, 3 * (size_t)n);
  cp.alloc(3 * (size_t)n);
  d2.alloc(n);
  idx.alloc(n);
  QueryBuffers qb = ctx->query_scratch(n, surface ? T : V);
  if (surface) launch_surface_query(ctx->stream, T, verts, tris, spheres, n, q.p, nullptr, qb, cp.p, d2.p, idx.p);
  else launch_vertex_query(ctx->stream, V, verts, n, q.p, nullptr, qb, d2.p, idx.p);
  if (points_out && surface) HIP_OK(hipMemcpyAsync(points_out, cp.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, ctx->stream));
  if (index_out) HIP_OK(hipMemcpyAsync(index_out, idx.p, sizeof(int) * n, hipMemcpyDeviceToHost, ctx->stream));
  if (dist2_out) HIP_OK(hipMemcpyAsync(dist2_out, d2.p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  ctx->finish(0, 0);
}
}  // namespace

int icp_closest_point_on_target(icp_ctx* ctx, int32_t n, const double* queries, double* points_out, int32_t* triangle_out,
                                double* dist2_out) {
  return guard([&] {
    require(ctx, "null context");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    Bound _b(ctx);
    const DeviceMesh& t = ctx->target;
    run_search(ctx, true, t.V, t.T, t.verts.p, t.tris.p, t.spheres.p, n, queries, points_out, triangle_out, dist2_out);
  });
}

int icp_closest_target_vertex(icp_ctx* ctx, int32_t n, const double* queries, int32_t* id_out, double* dist2_out) {
  return guard([&] {
    require(ctx, "null context");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    Bound _b(ctx);
    const DeviceMesh& t = ctx->target;
    run_search(ctx, false, t.V, t.T, t.verts.p, t.tris.p, t.spheres.p, n, queries, nullptr, id_out, dist2_out);
  });
}

int icp_closest_model_vertex(icp_ctx* ctx, const double* theta, int32_t n, const double* queries, int32_t* id_out,
                             double* dist2_out) {
  return guard([&] {
    require(ctx, "null context");
    check_theta_finite(ctx, theta);
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    Bound _b(ctx);
    StateSlot& s = ctx->state(theta);
    run_search(ctx, false, ctx->N, ctx->T, s.x.p, ctx->tris.p, nullptr, n, queries, nullptr, id_out, dist2_out);
  });
}

int icp_closest_point_on_model(icp_ctx* ctx, const double* theta, int32_t n, const double* queries, double* points_out,
                               int32_t* triangle_out, double* dist2_out) {
  return guard([&] {
    require(ctx, "null context");
    check_theta_finite(ctx, theta);
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    Bound _b(ctx);
    StateSlot& s = ctx->state(theta);
    ctx->ensure_model_spheres(s);
    run_search(ctx, true, ctx->N, ctx->T, s.x.p, ctx->tris.p, s.spheres.p, n, queries, points_out, triangle_out, dist2_out);
  });
}

// --------------------------------------------------------------------- proposal

int icp_proposal_create(icp_ctx* ctx, const icp_proposal_params* params, icp_proposal** out) {
  if (out) *out = nullptr;
  icp_proposal* p = nullptr;
  int rc = guard([&] {
    require(ctx && params && out, "null argument");
    require(params->direction == ICP_MODEL_SAMPLING || params->direction == ICP_TARGET_SAMPLING, "unknown direction");
    require(params->step_length != 0.0 && std::isfinite(params->step_length), "step_length must be finite and non-zero");
    require(params->tangential_noise > 0.0 && params->noise_along_normal > 0.0, "noise standard deviations must be positive");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    Bound _b(ctx);
    p = new icp_proposal();
    p->ctx = ctx;
    p->prm = *params;
    if (params->direction == ICP_TARGET_SAMPLING) {
      require(params->n_target_points >= 0 && (params->target_points || params->n_target_points == 0), "bad target points");
      p->K = params->n_target_points;
      p->target_pts.upload(params->target_points, 3 * (size_t)p->K);
      p->hint_nn.alloc(std::max(p->K, 1));
      p->hint_nn.fill_bytes(0xFF);
      p->nn_id.alloc(std::max(p->K, 1));
    } else {
      require(params->n_model_ids >= 0 && params->n_model_ids <= ctx->N, "n_model_ids out of range");
      p->K = params->n_model_ids;
    }
    p->prm.target_points = nullptr;  // caller memory is not retained
    p->work.alloc(eigen_work_doubles(ctx->r));
    p->work.fill_bytes(0);  // holds the completion counter of the eigenvector replay kernel
    if (ctx->eig_stream2) {  // ranks above 64: the second eigen stream's work area
      p->work2.alloc(eigen_work_doubles(ctx->r));
      p->work2.fill_bytes(0);
    }
    p->mpart_half_doubles = (size_t)regression_splits(std::max(p->K, 1)) * (ctx->r + 1) * (ctx->r + 1);
    p->Mpart.alloc(icp_proposal::kMpartRing * p->mpart_half_doubles);
    p->fscratch.alloc((size_t)(ctx->r + 1) * ctx->r + 8);
    pinned_alloc((void**)&p->h_cancel, sizeof(int) * 16);
    for (int i = 0; i < 16; ++i) p->h_cancel[i] = 0;
    pinned_alloc((void**)&p->h_eig, sizeof(int) * kPosteriorMemo);
    for (int i = 0; i < kPosteriorMemo; ++i) p->h_eig[i] = 0;
    p->status.alloc(3 * kPosteriorMemo);
    p->status.fill_bytes(0);
    p->eig_words.alloc(kPosteriorMemo);
    p->eig_words.fill_bytes(0);
    p->h_status.assign(3 * kPosteriorMemo, 0);
    p->memo.reset(new PosteriorEntry[kPosteriorMemo]);
    for (int i = 0; i < kPosteriorMemo; ++i) p->alloc_entry(p->memo[i]);
    ctx->proposals.push_back(p);
    *out = p;
  });
  if (rc != ICP_OK && p) delete p;
  return rc;
}

namespace { void release_front(StepFront& F); }

void icp_proposal_destroy(icp_proposal* p) {
  if (!p) return;
  {
    std::lock_guard<std::recursive_mutex> lk(p->ctx->mu);
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    (void)hipStreamSynchronize(p->ctx->front_stream);
    try { sync_eigen(*p->ctx); } catch (...) {}
    DeviceQuiesce _q;
    for (icp_evaluator* ev : p->ctx->evaluators)  // a half step launched ahead with this proposal holds entries of it
      if (ev->front.valid && (ev->front.props[0] == p || ev->front.props[1] == p)) release_front(ev->front);
    if (g_host_timing.on && eigen_speculation_supported(p->ctx->r)) eigen_debug_dump(p->work.p, p->ctx->r);
    if (p->h_cancel) pinned_free(p->h_cancel);
    if (p->h_eig) pinned_free(p->h_eig);
    auto& live = p->ctx->proposals;
    live.erase(std::remove(live.begin(), live.end(), p), live.end());
    delete p;
  }
}

int icp_proposal_num_candidates(const icp_proposal* p) { return p ? p->K : ICP_ERR_INVALID_ARG; }

int icp_proposal_set_sampler(icp_proposal* p, int32_t sampler) {
  return guard([&] {
    require(p != nullptr, "null argument");
    require(sampler == ICP_SAMPLER_EIGEN || sampler == ICP_SAMPLER_CHOLESKY_ROOT, "unknown sampler");
    icp_ctx& c = *p->ctx;
    require(sampler == ICP_SAMPLER_EIGEN || c.r <= kCholMaxRankAbi, "the Cholesky-root sampler covers ranks up to 256");
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    if (p->sampler == sampler) return;
    Bound _b(&c);
    // whatever was decomposed (or is being decomposed) the other way is dropped: its V / S mean something else
    HIP_OK(hipStreamSynchronize(c.stream));
    sync_eigen(c);
    for (icp_evaluator* ev : c.evaluators)
      if (ev->front.valid && (ev->front.props[0] == p || ev->front.props[1] == p)) release_front(ev->front);
    // Ranks above 64 have no decomposition of the root kind: there the posterior's own factorisation hands the factor out
    // (PosteriorFactorIO::Lout / Sout, written only when the posterior is computed), and the eigen route has no `root` form.  A
    // memoised posterior would be a memo hit that never rewrites V / S the new way — the sampler would draw from one kind of
    // buffer read as the other.  Those entries are forgotten altogether: the next use recomputes the posterior under the new sampler.
    const bool refactor = !eigen_speculation_supported(c.r);
    for (int i = 0; i < kPosteriorMemo; ++i) {
      p->memo[i].eig_valid = false; p->memo[i].eig_checked = false; p->memo[i].eig_event_valid = false;
      if (refactor) p->memo[i].valid = false;
    }
    p->side_parts = nullptr; p->side_parts_entry = nullptr;
    p->spec_entry = nullptr;
    p->warm_valid = false;
    p->sampler = sampler;
  });
}

int icp_proposal_propose(icp_proposal* p, const double* theta, const double* z, double* theta_out, int32_t* corr_id_out) {
  return guard([&] {
    require(p && z && theta_out, "null argument");
    icp_ctx& c = *p->ctx;
    check_theta_finite(&c, theta);
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    Bound _b(&c);
    const int r = c.r;
    const double* dz = c.stage(z, r);                 // :55 the caller's standard normals (on their way before the wait for the basis)
    PosteriorEntry& e = p->posterior(theta, false);  // NonRigidIcpProposal.scala:54
    p->ensure_eigen(e);
    p->await_eigen(e);
    // (the entry's status words travel with the proposal: one result copy, not two)
    launch_propose(c.stream, r, e.alpha.p, e.V.p, e.S.p, c.inv_sqrt_lambda.p, c.P.p, kSigma2, e.coeffs.p, dz,
                   p->prm.step_length, c.d_res.p, p->sampler == ICP_SAMPLER_CHOLESKY_ROOT, p->status.p + e.status_off, c.d_status.p);
    std::vector<int> ids;
    std::vector<uint8_t> keep;
    if (corr_id_out && p->K > 0) {
      ids.resize(p->K);
      keep.resize(p->K);
      HIP_OK(hipMemcpyAsync(ids.data(), e.id.p, sizeof(int) * p->K, hipMemcpyDeviceToHost, c.stream));
      HIP_OK(hipMemcpyAsync(keep.data(), e.keep.p, p->K, hipMemcpyDeviceToHost, c.stream));
    }
    c.finish(r, 3);
    for (int k = 0; k < 3; ++k) p->h_status.data()[e.status_off + k] = c.h_status[k];
    p->check_status(e);
    static const bool dbg = dev_env("ICP_DEBUG_EIGEN") != nullptr;
    if (dbg) std::fprintf(stderr, "eigen sweeps %d\n", p->h_status[e.status_off + 1]);
    std::memcpy(theta_out, theta, sizeof(double) * 10);
    for (int j = 0; j < r; ++j) {
      if (!std::isfinite(c.h_res[j])) fail(ICP_ERR_NOT_FINITE, "proposed coefficients are not finite");
      theta_out[10 + j] = c.h_res[j];
    }
    if (corr_id_out)
      for (int k = 0; k < p->K; ++k) corr_id_out[k] = keep[k] ? ids[k] : -1;
  });
}

int icp_proposal_log_transition(icp_proposal* p, const double* theta_from, const double* theta_to, double* out) {
  return guard([&] {
    require(p && out, "null argument");
    icp_ctx& c = *p->ctx;
    check_theta_finite(&c, theta_from);
    check_theta_finite(&c, theta_to);
    if (!pose_equal(theta_from, theta_to)) {  // NonRigidIcpProposal.scala:72-74
      *out = -INFINITY;
      return;
    }
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    Bound _b(&c);
    PosteriorEntry& e = p->posterior(theta_from, false);  // :76
    const double* dto = c.stage(theta_to + 10, c.r);
    TransitionTailIO io{e.alpha.p, e.M.p, e.coeffs.p, dto, p->prm.step_length, c.d_res.p, c.d_status.p};
    launch_transition_tails(c.stream, c.r, 1, &io, c.Ginv.p, kSigma2);
    sync_proposal_status(p);
    c.finish(1, 1);
    p->check_status(e);
    if (c.h_status[0] != 0) {  // the fixed-point form did not contract for this model/noise: direct factorisation
      const double* dto2 = c.stage(theta_to + 10, c.r);
      io.c_to = dto2;
      sync_eigen(c);  // (the direct form borrows the eigen work buffer)
      launch_transition_tail_direct(c.stream, c.r, io, c.G.p, kSigma2, p->work.p);
      c.finish(1, 1);
      if (c.h_status[0] != 0) fail(ICP_ERR_NOT_SPD, "G + sigma^2 M is not positive definite");
    }
    if (std::isnan(c.h_res[0])) fail(ICP_ERR_NOT_FINITE, "NaN transition probability");
    *out = c.h_res[0];
  });
}

int icp_proposal_posterior(icp_proposal* p, const double* theta, icp_posterior_view* view) {
  return guard([&] {
    require(p && view, "null argument");
    icp_ctx& c = *p->ctx;
    check_theta_finite(&c, theta);
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    Bound _b(&c);
    const int r = c.r, K = p->K;
    PosteriorEntry& e = p->posterior(theta, view->corr_aux != nullptr);
    if (view->V || view->S) { p->ensure_eigen(e); p->await_eigen(e); }
    view->n_candidates = K;
    auto d2h = [&](void* dst, const void* src, size_t bytes) {
      if (dst && bytes) HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c.stream));
    };
    d2h(view->corr_id, e.id.p, sizeof(int) * K);
    d2h(view->corr_aux, e.aux.p, sizeof(int) * K);
    d2h(view->corr_point, e.pt.p, sizeof(double) * 3 * K);
    d2h(view->keep, e.keep.p, K);
    d2h(view->alpha, e.alpha.p, sizeof(double) * r);
    d2h(view->M, e.M.p, sizeof(double) * r * r);
    d2h(view->V, e.V.p, sizeof(double) * r * r);
    d2h(view->S, e.S.p, sizeof(double) * r);
    sync_proposal_status(p);
    c.finish(0, 0);
    p->check_status(e);
  });
}

// --------------------------------------------------------------------- evaluators

int icp_evaluator_create(icp_ctx* ctx, const icp_evaluator_params* params, icp_evaluator** out) {
  if (out) *out = nullptr;
  icp_evaluator* ev = nullptr;
  int rc = guard([&] {
    require(ctx && params && out, "null argument");
    require(params->kind >= 0 && params->kind <= 2, "unknown evaluator kind");
    require(params->kind == ICP_EVAL_HAUSDORFF || (params->mode >= 0 && params->mode <= 2), "unknown evaluation mode");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    Bound _b(ctx);
    ev = new icp_evaluator();
    ev->ctx = ctx;
    ev->prm = *params;
    if (params->kind == ICP_EVAL_HAUSDORFF) {
      require(params->exp_rate > 0.0, "exp_rate must be positive");
      ev->Kt = ctx->target.V;  // MeshMetrics.hausdorffDistance: every target vertex against the model surface
      ev->d_tpts = ctx->target.verts.p;
    } else {
      require(params->gauss_sigma > 0.0, "gauss_sigma must be positive");
      require(params->kind != ICP_EVAL_COLLECTIVE_AVG_HAUSDORFF_BOUNDARY_AWARE || params->exp_rate > 0.0, "exp_rate must be positive");
      require(params->n_model_ids >= 0 && params->n_model_ids <= ctx->N, "n_model_ids out of range");
      require(params->n_target_points >= 0 && (params->target_points || params->n_target_points == 0), "bad target points");
      ev->Kt = params->n_target_points;
      ev->target_pts.upload(params->target_points, 3 * (size_t)ev->Kt);
      ev->d_tpts = ev->target_pts.p;
    }
    ev->prm.target_points = nullptr;
    const size_t Ka = std::max(ev->Kt, 1);
    ev->hint_tri.alloc(Ka); ev->hint_tri.fill_bytes(0xFF);
    ev->hint_nnv.alloc(Ka); ev->hint_nnv.fill_bytes(0xFF);
    ev->t2m_tri.alloc(Ka); ev->t2m_nnv.alloc(Ka);
    ev->t2m_cp.alloc(3 * Ka); ev->t2m_d2.alloc(Ka);
    ctx->evaluators.push_back(ev);
    *out = ev;
  });
  if (rc != ICP_OK && ev) delete ev;
  return rc;
}

void icp_evaluator_destroy(icp_evaluator* e) {
  if (!e) return;
  std::lock_guard<std::recursive_mutex> lk(e->ctx->mu);
  (void)hipSetDevice(e->ctx->device);
  (void)hipStreamSynchronize(e->ctx->stream);
  DeviceQuiesce _q;
  // a pre-launched half step holds a state slot of the context and memo entries of its proposals
  if (e->front.valid) release_front(e->front);
  auto& evs = e->ctx->evaluators;
  evs.erase(std::remove(evs.begin(), evs.end(), e), evs.end());
  delete e;
}

int icp_evaluator_log_value(icp_evaluator* e, const double* theta, double* out, double* aux) {
  int status = ICP_OK;
  int rc = guard([&] {
    require(e && out, "null argument");
    icp_ctx& c = *e->ctx;
    check_theta_finite(&c, theta);
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    Bound _b(&c);
    icp_evaluator::Memo* m = eval_lookup(e, theta);  // evaluators/EvaluationCaching.scala:32-36
    if (!m) {
      StateSlot& s = c.state(theta);
      enqueue_eval(e, s, 0);
      c.finish(8, 0);
      m = eval_store(e, theta);
      m->status = finish_eval(e, c.h_res, &m->value, m->aux);
    }
    *out = m->value;
    if (aux) std::memcpy(aux, m->aux, sizeof(double) * 4);
    status = m->status;
    if (status != ICP_OK) g_err = icp_status_string(status);
  });
  return rc != ICP_OK ? rc : status;
}

int icp_prior_log_value(int32_t rank, const double* theta, double* out) {
  return guard([&] {
    require(rank > 0 && theta && out, "bad argument");
    double nn = 0.0;
    for (int j = 0; j < rank; ++j) nn += theta[10 + j] * theta[10 + j];
    *out = -0.5 * nn - 0.5 * rank * std::log(2.0 * M_PI);  // MultivariateNormalDistribution(0, I).logpdf
  });
}

// --------------------------------------------------------------------- deterministic non-rigid ICP (next row 1)

int icp_fit_deterministic(icp_ctx* ctx, const icp_fit_params* prm, const double* theta_init, int32_t n_iterations, int32_t n_sigma,
                          const double* sigma2_seq, double* theta_out) {
  return guard([&] {
    require(ctx && prm && theta_out && sigma2_seq, "null argument");
    require(n_iterations >= 0 && n_sigma >= 0, "negative iteration count");
    require(prm->direction == ICP_MODEL_SAMPLING || prm->direction == ICP_TARGET_SAMPLING, "unknown direction");
    require(std::isfinite(prm->step_length), "step_length must be finite");
    icp_ctx& c = *ctx;
    check_theta_finite(&c, theta_init);
    const bool model_side = prm->direction == ICP_MODEL_SAMPLING;
    const int K = model_side ? prm->n_model_ids : prm->n_target_points;
    require(K >= 0 && (K == 0 || (model_side ? (const void*)prm->model_ids : (const void*)prm->target_points)), "bad sample list");
    if (model_side)
      for (int k = 0; k < K; ++k) require(prm->model_ids[k] >= 0 && prm->model_ids[k] < c.N, "model id out of range");
    for (int i = 0; i < n_sigma; ++i) require(sigma2_seq[i] > 0.0 && std::isfinite(sigma2_seq[i]), "sigma2 must be positive");
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    Bound _b(&c);
    const int r = c.r, Ka = std::max(K, 1);
    const Pose pose = c.pose_of(theta_init);
    DBuf<double> coeffs, x, P, cp, pts, Mpart, M, alpha, e, nhat, pt;
    DBuf<int> ids, nn, hint, corr_id, aux, status;
    DBuf<uint8_t> keep;
    coeffs.upload(theta_init + 10, r);
    x.alloc(3 * (size_t)c.N);
    P.alloc(3 * (size_t)Ka); cp.alloc(3 * (size_t)Ka);
    hint.alloc(Ka); hint.fill_bytes(0xFF);
    nn.alloc(Ka); corr_id.alloc(Ka); aux.alloc(Ka); keep.alloc(Ka);
    e.alloc(3 * (size_t)Ka); nhat.alloc(3 * (size_t)Ka); pt.alloc(3 * (size_t)Ka);
    if (model_side) ids.upload(prm->model_ids, K);
    else pts.upload(prm->target_points, 3 * (size_t)K);
    Mpart.alloc((size_t)regression_splits(Ka) * (r + 1) * (r + 1));
    M.alloc((size_t)r * r); alpha.alloc(r);
    DBuf<double> fscratch;
    fscratch.alloc((size_t)(r + 1) * r + 8);
    status.alloc(4); status.fill_bytes(0);
    const CorrBuffers cb{corr_id.p, aux.p, pt.p, keep.p, nhat.p, e.p};
    for (int si = 0; si < n_sigma; ++si) {
      const double wt = 1.0 / sigma2_seq[si];                                     // isotropic noise N(0, sigma2·I) (:81)
      for (int it = 0; it <= n_iterations; ++it) {                                // nbIterations = numIterations .. 0 (:55-104)
        launch_instance(c.stream, c.N, r, c.Qp.p, c.ref.p, c.mean.p, pose, coeffs.p, x.p);      // :61
        if (K > 0) {
          if (model_side) {                                                       // :72-74
            launch_gather_points(c.stream, K, x.p, ids.p, P.p);
            QueryBuffers qb = c.query_scratch(K, c.target.T);
            launch_surface_query(c.stream, c.target.T, c.target.verts.p, c.target.tris.p, c.target.spheres.p, K, P.p, hint.p, qb, cp.p,
                                 nullptr, nullptr);
            launch_correspond_plain(c.stream, K, ids.p, cp.p, c.ref.p, c.mean.p, cb);
          } else {                                                                // :76-78
            QueryBuffers qb = c.query_scratch(K, c.N);
            launch_vertex_query(c.stream, c.N, x.p, K, pts.p, hint.p, qb, nullptr, nn.p);
            launch_correspond_plain(c.stream, K, nn.p, pts.p, c.ref.p, c.mean.p, cb);
          }
        }
        int splits = 1;
        launch_regression(c.stream, K, r, c.Q.p, cb, wt, 0.0, Mpart.p, &splits);   // model.posterior(corr, sigma2) (:81)
        PosteriorFactorIO io{Mpart.p, splits, M.p, alpha.p, status.p, fscratch.p};
        launch_posterior_factor(c.stream, r, 1, &io);                              // posterior.mean (:82)
        launch_mean_step(c.stream, r, alpha.p, c.P.p, kSigma2, prm->step_length, coeffs.p);   // :84-85
      }
    }
    HIP_OK(hipMemcpyAsync(c.h_res, coeffs.p, sizeof(double) * r, hipMemcpyDeviceToHost, c.stream));
    HIP_OK(hipMemcpyAsync(c.h_status, status.p, sizeof(int), hipMemcpyDeviceToHost, c.stream));
    c.finish(0, 0);
    if (c.h_status[0] != 0) fail(ICP_ERR_NOT_SPD, "regression normal equations are not positive definite");
    std::memcpy(theta_out, theta_init, sizeof(double) * 10);
    for (int j = 0; j < r; ++j) {
      if (!std::isfinite(c.h_res[j])) fail(ICP_ERR_NOT_FINITE, "fitted coefficients are not finite");
      theta_out[10 + j] = c.h_res[j];
    }
  });
}

// --------------------------------------------------------------------- posterior variability maps (next row 3)

int icp_posterior_variability(icp_ctx* ctx, int32_t n_samples, const double* thetas, int32_t mode, const double* theta_ref, double* out) {
  return guard([&] {
    require(ctx && thetas && out, "null argument");
    require(n_samples >= 2, "at least two samples are needed");
    require(mode >= 0 && mode <= 2, "unknown mode");
    require(mode != 1 || theta_ref, "theta_ref is null");
    icp_ctx& c = *ctx;
    const size_t P = 10 + (size_t)c.r, n3 = 3 * (size_t)c.N;
    for (int s = 0; s < n_samples; ++s) check_theta_finite(&c, thetas + s * P);
    if (mode == 1) check_theta_finite(&c, theta_ref);
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    Bound _b(&c);
    DBuf<double> X, coeffs, nrm, tmp, res;
    X.alloc((size_t)n_samples * n3);
    std::vector<double> hc((size_t)(n_samples + 1) * c.r);
    for (int s = 0; s < n_samples; ++s) std::memcpy(&hc[(size_t)s * c.r], thetas + s * P + 10, sizeof(double) * c.r);
    if (mode == 1) std::memcpy(&hc[(size_t)n_samples * c.r], theta_ref + 10, sizeof(double) * c.r);
    coeffs.upload(hc.data(), hc.size());
    nrm.alloc(n3); tmp.alloc(n3); res.alloc(c.N);
    for (int s = 0; s < n_samples; ++s)   // ModelFittingParameters.transformedMesh of every sample (LogHelper.logSamples2shapes)
      launch_instance(c.stream, c.N, c.r, c.Qp.p, c.ref.p, c.mean.p, c.pose_of(thetas + s * P), coeffs.p + (size_t)s * c.r,
                      X.p + (size_t)s * n3);
    if (mode == 1) {
      launch_instance(c.stream, c.N, c.r, c.Qp.p, c.ref.p, c.mean.p, c.pose_of(theta_ref), coeffs.p + (size_t)n_samples * c.r, tmp.p);
      launch_vertex_normals(c.stream, c.N, tmp.p, c.tris.p, c.adj_off.p, c.adj.p, nrm.p);
    } else if (mode == 2) {
      HIP_OK(hipMemsetAsync(nrm.p, 0, sizeof(double) * n3, c.stream));
      for (int s = 0; s < n_samples; ++s) {
        launch_vertex_normals(c.stream, c.N, X.p + (size_t)s * n3, c.tris.p, c.adj_off.p, c.adj.p, tmp.p);
        launch_accumulate(c.stream, (int)n3, tmp.p, s == n_samples - 1 ? 1.0 / n_samples : 0.0, nrm.p);
      }
    }
    launch_variability(c.stream, c.N, n_samples, X.p, mode, nrm.p, res.p);
    HIP_OK(hipMemcpyAsync(out, res.p, sizeof(double) * c.N, hipMemcpyDeviceToHost, c.stream));
    c.finish(0, 0);
  });
}

// --------------------------------------------------------------------- registration metrics (next row 4)

int icp_mesh_metrics(icp_ctx* ctx, const double* theta, double* out) {
  return guard([&] {
    require(ctx && out, "null argument");
    icp_ctx& c = *ctx;
    check_theta_finite(&c, theta);
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    Bound _b(&c);
    StateSlot& s = c.state(theta);
    double* res = c.d_res.p;
    HIP_OK(hipMemsetAsync(res, 0, sizeof(double) * 16, c.stream));
    // reconstruction -> target: every model vertex against the target surface (shared with the proposals/evaluators of the state)
    c.ensure_surface_prefix(s, c.N);
    launch_dist_stats(c.stream, c.N, s.surf_d2.p, nullptr, nullptr, 0, res + 0);                       // avgDistance, one-sided max
    const bool flags = c.target.n_boundary > 0;
    if (flags) c.ensure_nnv_prefix(s, c.N);
    launch_dist_stats(c.stream, c.N, s.surf_d2.p, flags ? c.target.boundary.p : nullptr, flags ? s.surf_nnv.p : nullptr, c.target.V,
                      res + 4);                                                                        // boundary-aware (:31-42)
    // target -> reconstruction: every target vertex against the current model surface (hausdorffDistance is symmetric)
    DBuf<double> d2;
    DBuf<int> hint;
    d2.alloc(c.target.V); hint.alloc(c.target.V); hint.fill_bytes(0xFF);
    c.ensure_model_spheres(s);
    QueryBuffers qb = c.query_scratch(c.target.V, c.T);
    launch_surface_query(c.stream, c.T, s.x.p, c.tris.p, s.spheres.p, c.target.V, c.target.verts.p, hint.p, qb, nullptr, d2.p, nullptr);
    launch_dist_stats(c.stream, c.target.V, d2.p, nullptr, nullptr, 0, res + 8);
    c.finish(12, 0);
    const double* h = c.h_res;
    out[0] = h[0] / h[2];
    out[1] = std::max(h[1], h[9]);
    out[2] = h[6] > 0.0 ? h[4] / h[6] : NAN;
    out[3] = h[6] > 0.0 ? h[5] : NAN;
    out[4] = h[6];
  });
}

// --------------------------------------------------------------------- fused chain step

} // extern "C" (helper)
namespace {
// 0: never, 1: always, 2: adaptive (ICP_SPECULATION / ICP_NO_SPECULATION)
int speculation_mode() {
  static const int mode = [] {
    if (std::getenv("ICP_NO_SPECULATION")) return 0;
    const char* v = std::getenv("ICP_SPECULATION");
    return v ? (std::atoi(v) != 0 ? 1 : 0) : 2;
  }();
  return mode;
}
}  // namespace
extern "C" {
int icp_chain_eval_step(icp_evaluator* e, int32_t n_props, icp_proposal* const* props, const double* theta_cur,
                        const double* theta_prop, double* log_value_prop, double* fwd, double* bwd) {
  int status = ICP_OK;
  int rc = guard([&] {
    require(e && theta_cur && theta_prop && log_value_prop, "null argument");
    require(n_props >= 0 && n_props <= 8 && (n_props == 0 || (props && fwd && bwd)), "bad proposal list");
    icp_ctx& c = *e->ctx;
    for (int i = 0; i < n_props; ++i) require(props[i] && props[i]->ctx == &c, "proposal belongs to another context");
    check_theta_finite(&c, theta_cur);
    check_theta_finite(&c, theta_prop);
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    Bound _b(&c);
    const int r = c.r;
    icp_evaluator::Memo* m = eval_lookup(e, theta_prop);
    const bool need_eval = m == nullptr;
    const bool shape_only = pose_equal(theta_cur, theta_prop);
    // Ranks above 64 (one workgroup factors, one reduces to tridiagonal form: 0.2 + 0.7 ms at rank 200 with most of the chip idle):
    // the posteriors go first and the proposed state's decomposition starts at once on the eigen stream, BESIDE the evaluator's
    // searches on this one — if the state is accepted, the next proposal finds its basis done or under way; if not, the work
    // was done on CUs nobody needed.  (Acceptance tracked as in the merged step; not worth it when next to nothing is accepted.)
    if (!e->last_prop.empty()) {
      const bool accepted = std::memcmp(e->last_prop.data(), theta_cur, sizeof(double) * (10 + (size_t)r)) == 0;
      e->acc_ema = 0.9 * e->acc_ema + (accepted ? 0.1 : 0.0);
    }
    const int spec_mode = speculation_mode();
    const bool spec_ok = r > 64 && n_props == 1 && need_eval && !c.speculation_off &&
                         (spec_mode == 1 || (spec_mode == 2 && e->acc_ema >= 0.1));
    const bool spec_big = spec_ok && shape_only;
    // a pose move changes the state too: if it is kept, the next ICP proposal draws from the posterior at the NEW state — which
    // nothing on this path computes (the transition probabilities across a pose change are zero).  Started here, ahead, it is
    // done or under way by then: search (the evaluator's own, for a model-sampling proposal), regression, factorisation and
    // decomposition, all beside the evaluator.
    const bool spec_pose = spec_ok && !shape_only;
    if (need_eval && !spec_big && !spec_pose) {
      StateSlot& s = c.state(theta_prop);
      enqueue_eval(e, s, 0);
    }
    PosteriorEntry* ec[8];
    PosteriorEntry* ep[8];
    TransitionTailIO tails[16];
    int n_tails = 0;
    bool eval_enqueued = false;
    // (… and so do the one-workgroup factorisations and the tails: they go to a stream of their own, behind the regression; the
    // decomposition follows them on the eigen stream; this stream goes on with the searches and waits for the tails before the
    // results are copied)
    const hipStream_t side = spec_big || spec_pose ? c.front_stream : nullptr;  // (the merged step's second stream: idle on this path)
    // the decomposition of a posterior whose factorisation has just gone to the side stream: behind that — or, if the posterior was
    // computed just now, beside it: M = I + the summed partials is written by a launch at the head of the decomposition as well
    // (the same values the factorisation's assembly writes)
    hipStream_t es_ahead = nullptr;  // the stream part 1 of a decomposition went to (part 2 follows it there)
    auto decompose_ahead = [&](icp_proposal* p, PosteriorEntry& en, int part = 0) {
      if (part == 2) {
        if (es_ahead) p->ensure_eigen_on(en, es_ahead, 2);
        return;
      }
      if (en.eig_valid) return;
      if (c.eig_last && c.eig_last != c.eig_stream) (void)eigen_stream_for(c, c.eig_stream);  // (a batch's stream was in use: drained)
      c.eig_last = c.eig_stream;
      const hipStream_t es = (c.eig_stream2 && (p->eig_flip++ & 1)) ? c.eig_stream2 : c.eig_stream;  // two under way at a time
      es_ahead = es;
      if (p->side_parts && p->side_parts_entry == &en) {
        // (computed just now, all of it on the side stream: nothing of this entry is on the context stream — which carries the
        // evaluator's searches by now, and the decomposition must not wait for those)
        HIP_OK(hipStreamWaitEvent(es, c.ev_sum, 0));
        launch_assemble_posterior_matrix(es, r, p->side_parts, en.M.p);
        HIP_OK(hipEventRecord(c.ev_asm, es));
        p->side_asm_pending = true;
      } else {
        HIP_OK(hipEventRecord(c.ev_ready, c.stream));  // (whatever of this entry is still in flight on the context stream)
        HIP_OK(hipStreamWaitEvent(es, c.ev_ready, 0));
        HIP_OK(hipStreamWaitEvent(es, c.ev_side, 0));
      }
      p->ensure_eigen_on(en, es, part);
    };
    // Order of issue on this path (the host needs 3-6 µs per launch, the context stream is idle until it gets the evaluator's):
    // the proposed state's instance -> the evaluator's searches (the model ids the posterior's own searches will cover — 0..K — left
    // out: StateSlot's detached ranges) and its target-to-model half -> the posterior on the side stream (searches of ids 0..K,
    // correspondences, regression, factorisation) -> the head of the decomposition -> the evaluator's model-to-target reductions,
    // behind the side stream's searches -> the decomposition's other launches -> the tails.
    // the two coefficient vectors the tails read: the states' own copies on the device (the current state's slot is kept from
    // being recycled for the proposed one), staged from the host only for a current state that has no slot any more — here, ahead of
    // the evaluator's launches: ev_inst covers them
    const double *d_cur = nullptr, *d_prop = nullptr;
    if (shape_only && n_props > 0) {
      StateSlot* sc = c.find_state(theta_cur);
      if (sc) sc->stamp = ++c.clock;
      d_cur = sc ? sc->coeffs.p : c.stage(theta_cur + 10, r);
      d_prop = c.state(theta_prop).coeffs.p;
    }
    bool split_eval = false;
    if (side && need_eval && n_props == 1) {
      icp_proposal* p0 = props[0];
      PosteriorEntry* known = p0->find_entry(theta_prop);
      if (known) known->stamp = ++p0->clock;  // (not the one a posterior of the current state, computed first, recycles)
      const int R = (!known && p0->prm.direction == ICP_MODEL_SAMPLING) ? p0->K : 0;
      const int Rn = (R > 0 && p0->prm.boundary_aware && c.target.n_boundary > 0) ? R : 0;
      StateSlot& s = c.state(theta_prop);
      HIP_OK(hipEventRecord(c.ev_inst, c.stream));
      c.ev_inst_slot = &s;
      enqueue_eval_searches(e, s, 0, R, Rn);
      split_eval = true;
    }
    struct InstGuard { icp_ctx& c; ~InstGuard() { c.ev_inst_slot = nullptr; } } inst_guard{c};
    auto eval_reductions = [&] {
      if (split_eval && !eval_enqueued) { enqueue_eval_reductions(e, c.state(theta_prop), 0); eval_enqueued = true; }
    };
    PosteriorEntry* pose_entry = nullptr;
    if (spec_pose) {
      PosteriorEntry& en = props[0]->posterior(theta_prop, false, side);
      decompose_ahead(props[0], en, 1);  // (its other launches: behind the evaluator's, below)
      pose_entry = &en;
      eval_reductions();
    }
    if (shape_only && n_props > 0) {
      for (int i = 0; i < n_props; ++i) {
        icp_proposal* p = props[i];
        ec[i] = &p->posterior(theta_cur, false, side);
        ep[i] = &p->posterior(theta_prop, false, side);
        // (each tail passes its posterior's status words on to d_status[16 + 3·tail …]: they come back with the step's results)
        tails[n_tails] = TransitionTailIO{ec[i]->alpha.p, ec[i]->M.p, d_cur, d_prop, p->prm.step_length, c.d_res.p + 8 + n_tails,
                                          c.d_status.p + n_tails, p->status.p + ec[i]->status_off, c.d_status.p + 16 + 3 * n_tails};
        ++n_tails;
        tails[n_tails] = TransitionTailIO{ep[i]->alpha.p, ep[i]->M.p, d_prop, d_cur, p->prm.step_length, c.d_res.p + 8 + n_tails,
                                          c.d_status.p + n_tails, p->status.p + ep[i]->status_off, c.d_status.p + 16 + 3 * n_tails};
        ++n_tails;
      }
      if (side) {
        // (the staged coefficients and — for entries found in the memo — everything else the tails read: behind ev_inst, which was
        // recorded ahead of the evaluator's searches, or behind ev_ready)
        if (split_eval) {
          HIP_OK(hipStreamWaitEvent(side, c.ev_inst, 0));
        } else {
          HIP_OK(hipEventRecord(c.ev_ready, c.stream));
          HIP_OK(hipStreamWaitEvent(side, c.ev_ready, 0));
        }
        // Order of issue (the host needs 3-6 µs per launch): factorisation (inside posterior) -> the HEAD of the decomposition (the
        // reduction to tridiagonal form, 0.48 ms on one workgroup) -> the evaluator's ten launches -> the decomposition's other ten
        // launches (they run behind the reduction whenever they are issued) -> the tails (behind the factorisation: 0.1-0.3 ms of
        // slack).  The evaluator's searches used to start 70-120 µs after the regression had ended because they were issued last.
        decompose_ahead(props[0], *ep[0], 1);
        eval_reductions();
        if (need_eval && !eval_enqueued) {
          StateSlot& s = c.state(theta_prop);
          enqueue_eval(e, s, 0);
          eval_enqueued = true;
        }
        decompose_ahead(props[0], *ep[0], 2);
      }
      for (int t0 = 0; t0 < n_tails; t0 += 8)
        launch_transition_tails(side ? side : c.stream, r, std::min(8, n_tails - t0), tails + t0, c.Ginv.p, kSigma2);
      if (side) {
        HIP_OK(hipEventRecord(c.ev_side, side));
        props[0]->side_factor_pending = false;  // (this stream waits for ev_side below)
      }
    }
    eval_reductions();
    if (need_eval && (spec_big || spec_pose) && !eval_enqueued) {
      StateSlot& s = c.state(theta_prop);
      enqueue_eval(e, s, 0);
    }
    if (pose_entry) decompose_ahead(props[0], *pose_entry, 2);
    if (side && shape_only && n_props > 0) HIP_OK(hipStreamWaitEvent(c.stream, c.ev_side, 0));
    e->last_prop.assign(theta_prop, theta_prop + 10 + r);
    c.finish(8 + (size_t)n_tails, (size_t)n_tails);
    for (int t = 0; t < n_tails; ++t)
      if (c.h_status[t] != 0) {  // rare: fixed-point tail did not contract -> direct kernel, one at a time
        std::vector<double> saved(c.h_res, c.h_res + 8 + n_tails);
        icp_proposal* p = props[t / 2];
        TransitionTailIO io = tails[t];
        io.c_from = c.stage((t % 2 == 0 ? theta_cur : theta_prop) + 10, r);
        io.c_to = c.stage((t % 2 == 0 ? theta_prop : theta_cur) + 10, r);
        io.out = c.d_res.p;
        io.status = c.d_status.p + 64;
        io.relay_in = nullptr; io.relay_out = nullptr;
        sync_eigen(c);  // (the direct form borrows the eigen work buffer)
        launch_transition_tail_direct(c.stream, r, io, c.G.p, kSigma2, p->work.p);
        c.finish(1, 96);
        if (c.h_status[64] != 0) fail(ICP_ERR_NOT_SPD, "G + sigma^2 M is not positive definite");
        saved[8 + t] = c.h_res[0];
        std::memcpy(c.h_res, saved.data(), sizeof(double) * saved.size());
      }
    if (need_eval) {
      m = eval_store(e, theta_prop);
      m->status = finish_eval(e, c.h_res, &m->value, m->aux);
    }
    *log_value_prop = m->value;
    status = m->status;
    for (int i = 0; i < n_props; ++i) {
      if (!shape_only) { fwd[i] = -INFINITY; bwd[i] = -INFINITY; continue; }
      for (int k = 0; k < 3; ++k) {
        props[i]->h_status.data()[ec[i]->status_off + k] = c.h_status[16 + 3 * (2 * i) + k];
        props[i]->h_status.data()[ep[i]->status_off + k] = c.h_status[16 + 3 * (2 * i + 1) + k];
      }
      props[i]->check_status(*ec[i]);
      props[i]->check_status(*ep[i]);
      fwd[i] = c.h_res[8 + 2 * i];
      bwd[i] = c.h_res[9 + 2 * i];
      if (std::isnan(fwd[i]) || std::isnan(bwd[i])) fail(ICP_ERR_NOT_FINITE, "NaN transition probability");
    }
  });
  return rc != ICP_OK ? rc : status;
}

// --------------------------------------------------------------------- one Metropolis–Hastings step, one submission

namespace {

// the merged-launch pipeline covers the configurations of the reference's experiments that run on closed targets
// (apps/femur/*): one proposal per ICP direction, model-to-target likelihood; everything else takes the per-stage path
bool step_pipeline_covers(icp_evaluator* e, int n_props, icp_proposal* const* props) {
  icp_ctx& c = *e->ctx;
  if (n_props < 1 || n_props > 2) return false;
  const icp_evaluator_params& ep = e->prm;
  if (ep.kind == ICP_EVAL_HAUSDORFF) return false;
  // (the boundary-aware collective evaluator on a target WITH boundary needs the nearest-vertex pass behind the searches)
  if (ep.kind == ICP_EVAL_COLLECTIVE_AVG_HAUSDORFF_BOUNDARY_AWARE && c.target.n_boundary > 0) return false;
  const bool m2t = ep.mode != ICP_TARGET_TO_MODEL, t2m = ep.mode != ICP_MODEL_TO_TARGET;
  if (m2t && ep.n_model_ids < 1) return false;
  if (t2m && (e->Kt < 1 || c.T < 1 || (size_t)(e->Kt + 4) * (size_t)cand_stride(c.T) > kMaxCandidates)) return false;
  int n_model = 0, n_target = 0, ksurf = m2t ? ep.n_model_ids : 0;
  for (int i = 0; i < n_props; ++i) {
    const icp_proposal* p = props[i];
    if (p->K < 1) return false;
    // (the Cholesky-root sampler above rank 64 gets its factor from the per-stage factorisation: icp_proposal::posterior)
    if (p->sampler == ICP_SAMPLER_CHOLESKY_ROOT && !eigen_speculation_supported(c.r)) return false;
    if (p->prm.direction == ICP_MODEL_SAMPLING) {
      if (p->prm.boundary_aware && c.target.n_boundary > 0) return false;  // needs the nearest-vertex pass (:98-99)
      ++n_model;
      ksurf = std::max(ksurf, p->K);
    } else {
      ++n_target;
    }
  }
  if (n_model > 1 || n_target > 1) return false;
  if (ksurf < 1) return false;  // (a TargetToModel evaluator beside a TargetSampling proposal alone: no model-side surface query at all)
  if (c.target.T < 1 || (size_t)(ksurf + 4) * (size_t)cand_stride(c.target.T) > kMaxCandidates) return false;
  if (n_target && (size_t)(props[0]->K + props[n_props - 1]->K + 8) * (size_t)cand_stride(c.N) > kMaxCandidates) return false;
  return step_finish_supported(c.r);
}

}  // namespace

}  // extern "C"

namespace {

// give back what a front holds without recording anything (its launches, if any, are harmless: they wrote to a state
// slot and memo entries that nobody refers to, to the search scratch and to the hints, which may be stale by design)
void release_front(StepFront& F);
// A half step launched ahead for an outcome that did not happen.  Its launches may still be running — they write the search
// scratch, the hints, a state slot and memo entries that the replacement is about to be given — so the step that replaces it
// must be ordered behind it: it takes the dropped front's own stream (enqueue_front toggles the parity back), where stream
// order does that, instead of the other one, where nothing would.
void drop_front(icp_evaluator* e) {
  if (!e->front.valid) return;
  const int parity = e->front.parity;
  release_front(e->front);
  e->front_parity = parity ^ 1;
}
void release_front(StepFront& F) {
  if (F.s) F.s->reserved = false;
  for (int i = 0; i < F.n_props; ++i)
    if (F.ep[i]) F.ep[i]->reserved = false;
  if (F.valid && F.eigen_first_use && F.generator >= 0 && F.ec[F.generator]) F.ec[F.generator]->eig_checked = false;
  F = StepFront{};
}

// KL bases of a state's posteriors `ec` that are not on record yet: all of them are started now, in ONE launch on the eigen
// stream (they run side by side); a step waits for the one it draws from only (through its completion word) — the other
// is ready when a later step draws from it.  m_in_flight: `stream` may still be writing what they read.
struct EigenCollect {  // the decompositions of a batch of chains, launched together (icp_chain_step_batched)
  hipStream_t stream;                  // the eigen stream of the batch's first context
  std::vector<EigenRequest> rq;
  std::vector<PosteriorEntry*> all;    // every entry with a request: ONE event, recorded behind the launch, stands for them all
};

// Events for the decompositions of whole batches: one ring PER DEVICE (an event belongs to the device that was current when it
// was created) that lives as long as the process — the entries of many chains, many proposals, many contexts, destroyed in any
// order, point at its slots.  A slot is handed out again after 64 batches; the entries of the earlier batch notice by the slot's
// generation counter (PosteriorEntry::eigen_event_stale) and wait on the host instead of on an event that now stands for other work.
// One record per batch instead of one per chain that moved: ≈ 2.3 µs of host time each.
struct BatchEventSlot { hipEvent_t ev = nullptr; uint64_t gen = 0; };
BatchEventSlot& next_batch_event(int device) {
  constexpr int kMaxDevices = 64, kRing = 64;
  static std::mutex mu;
  static BatchEventSlot ring[kMaxDevices][kRing];
  static unsigned turn[kMaxDevices] = {};
  require(device >= 0 && device < kMaxDevices, "device ordinal out of range");
  std::lock_guard<std::mutex> lk(mu);
  BatchEventSlot& e = ring[device][turn[device]++ % kRing];
  if (!e.ev) HIP_OK(hipEventCreateWithFlags(&e.ev, hipEventDisableTiming));  // (the caller has bound `device`)
  ++e.gen;
  return e;
}
// A launch that waits ON THE DEVICE for a word another stream's launch raises is safe as long as the waiting workgroups cannot keep
// the launch they wait for from becoming resident.  One chain's step cannot (its first launch is 14 workgroups, a decomposition
// six), two chains' neither; the batched step orders its chip-wide first launch behind the residency of its decompositions
// explicitly (StepBatchGate).  MANY contexts stepped one by one from many threads could, together, fill the chip with spinning first
// launches (round 2 saw the batched form of this: a 50 ms time-out): from three live contexts on, the single-chain
// step therefore takes its cross-stream order from events and stream order — no device-side wait at all.
bool device_side_waits_allowed() { return g_live_contexts.load(std::memory_order_relaxed) <= 2; }

void start_decompositions(icp_ctx& c, int n_props, icp_proposal* const* props, PosteriorEntry* const* ec, bool m_in_flight,
                          EigenCollect* collect = nullptr) {
  const int r = c.r;
  EigenRequest rqs[2];
  PosteriorEntry* need[2];
  int nn = 0;
  for (int i = 0; i < n_props; ++i)
    if (!ec[i]->eig_valid) { props[i]->prepare_eigen(*ec[i], &rqs[nn]); need[nn++] = ec[i]; }
  if (nn == 0) return;
  need[0]->eig_done_shared = nullptr;
  for (int i = 1; i < nn; ++i) need[i]->eig_done_shared = need[0]->eig_done;
  for (int i = 0; i < nn; ++i) need[i]->eig_shared_gen = nullptr;
  if (collect) {
    if (m_in_flight) HIP_OK(hipStreamSynchronize(c.stream));
    for (int i = 0; i < nn; ++i) need[i]->eig_event_valid = true;  // (recorded by the caller behind the batch's launch)
    (void)eigen_stream_for(c, collect->stream);
    for (int i = 0; i < nn; ++i) collect->rq.push_back(rqs[i]);
    for (int i = 0; i < nn; ++i) collect->all.push_back(need[i]);
    return;
  }
  const hipStream_t es = eigen_stream_for(c, c.eig_stream);
  // M of these entries is complete when they were recorded by a finished chain step (the host has seen its results);
  // only work another entry point has put on `stream` may still be writing them
  if (m_in_flight) {
    HIP_OK(hipEventRecord(c.ev_ready, c.stream));
    HIP_OK(hipStreamWaitEvent(es, c.ev_ready, 0));
  }
  if (launch_posterior_eigen_pair(es, r, c.sqrt_lambda.p, nn, rqs)) {
    if (!device_side_waits_allowed()) {
      // many contexts in the process: an event instead of the completion word (see device_side_waits_allowed)
      HIP_OK(hipEventRecord(need[0]->eig_done, es));
      for (int i = 0; i < nn; ++i) { need[i]->eig_event_valid = true; need[i]->done_value = 0; }
      return;
    }
    // completion words: the step's first launch waits for the one it draws from on the device; no event (host time on the
    // accepted path) — whoever else needs the basis waits for the eigen stream on the host (await_eigen)
    for (int i = 0; i < nn; ++i) need[i]->eig_event_valid = false;
    return;
  }
  for (int i = 0; i < nn; ++i) {  // (ranks > 64: one after the other)
    need[i]->done_value = 0;
    launch_posterior_eigen(es, r, rqs[i].M, c.sqrt_lambda.p, rqs[i].Vwarm, rqs[i].V, rqs[i].Vt, rqs[i].S, rqs[i].work, rqs[i].status,
                           nullptr, rqs[i].host_status);
  }
  HIP_OK(hipEventRecord(need[0]->eig_done, es));  // (one event for two launches in a row)
  for (int i = 0; i < nn; ++i) need[i]->eig_event_valid = true;
}

void front_launches(icp_evaluator* e, int n_props, icp_proposal* const* props, int generator, const double* key, StepFront& F, bool batched,
                    bool two_streams);

// launches 1-3 of the step (theta_cur --generator/key--> proposal); `key` = z or the proposed state (see StepFront)
// (batched: the launches are being captured for icp_chain_step_batched — one stream, nothing to wait for on the device
// but the decomposition)
void enqueue_front(icp_evaluator* e, int n_props, icp_proposal* const* props, int generator, const double* theta_cur,
                   const double* key, StepFront& F, bool batched = false) {
  icp_ctx& c = *e->ctx;
  const int r = c.r;
  F = StepFront{};
  F.n_props = n_props; F.generator = generator;
  for (int i = 0; i < n_props; ++i) F.props[i] = props[i];
  F.theta_cur.assign(theta_cur, theta_cur + 10 + r);
  F.key.assign(key, key + (generator >= 0 ? r : 10 + r));
  F.parity = (e->front_parity ^= 1);
  // ---- cached side: posterior of the current state for every proposal (+ its KL basis for the generating one)
  PosteriorEntry** ec = F.ec;
  PosteriorEntry** ep = F.ep;
  bool missing = false;
  for (int i = 0; i < n_props; ++i) missing = missing || !props[i]->find_entry(theta_cur);
  if (missing && c.front_stream_used) {  // the posteriors are computed on `stream` with the scratch a step in flight may still use
    HIP_OK(hipStreamSynchronize(c.front_stream));
    c.front_stream_used = false;
  }
  for (int i = 0; i < n_props; ++i) ec[i] = &props[i]->posterior(theta_cur, false);  // NonRigidIcpProposal.scala:54,76
  if (missing) c.stream_used_elsewhere = true;  // … and this step reads them
  const bool m_in_flight = c.stream_used_elsewhere;  // `stream` may still be writing what the decompositions below read
  const bool two_streams = !c.pipeline_off && !batched && device_side_waits_allowed();
  F.stream = (F.parity && two_streams) ? c.front_stream : c.stream;
  if (F.stream == c.front_stream) {
    if (c.stream_used_elsewhere) {  // another entry point has work on `stream` that this step may depend on: join once
      HIP_OK(hipEventRecord(c.ev_join, c.stream));
      HIP_OK(hipStreamWaitEvent(c.front_stream, c.ev_join, 0));
    }
    c.front_stream_used = true;
  }
  c.stream_used_elsewhere = false;
  start_decompositions(c, n_props, props, ec, m_in_flight);
  bool eigen_first_use = false;
  if (generator >= 0) {
    // (await_eigen on the front's stream: through the decomposition's own completion word when it has one — see launch 1)
    if (ec[generator]->done_value == 0 && ec[generator]->eigen_event()) HIP_OK(hipStreamWaitEvent(F.stream, ec[generator]->eigen_event(), 0));
    eigen_first_use = !ec[generator]->eig_checked;  // (possibly of an earlier prefetch or speculation): fetch its status
    ec[generator]->eig_checked = true;
  }
  F.eigen_first_use = eigen_first_use;

  // ---- new side: one state slot, one memo entry per proposal
  StateSlot& s = c.fresh_state();
  s.reserved = true;
  F.s = &s;
  s.pose = c.pose_of(generator >= 0 ? theta_cur : key);
  for (int i = 0; i < n_props; ++i) { ep[i] = &props[i]->fresh_entry(); ep[i]->reserved = true; }
  front_launches(e, n_props, props, generator, key, F, batched, two_streams);
  F.valid = true;
}

// launches 1-4 of a merged step with every choice made: F.ec / F.ep (the posterior entries of the current and of the proposed state),
// F.s (the proposed state's slot, its pose set), F.parity, F.stream.  enqueue_front above makes those choices from the memo; the
// on-device chain loop (icp_chains_run_on_device) fixes them once per chain and captures the arguments.
void front_launches(icp_evaluator* e, int n_props, icp_proposal* const* props, int generator, const double* key, StepFront& F, bool batched,
                    bool two_streams) {
  icp_ctx& c = *e->ctx;
  const int r = c.r;
  PosteriorEntry** ec = F.ec;
  PosteriorEntry** ep = F.ep;
  StateSlot& s = *F.s;
  const icp_evaluator_params& evp = e->prm;
  icp_proposal* pm = nullptr;  // ModelSampling proposal
  icp_proposal* pt = nullptr;  // TargetSampling proposal
  int im = -1, it = -1;
  for (int i = 0; i < n_props; ++i) {
    if (props[i]->prm.direction == ICP_MODEL_SAMPLING) { pm = props[i]; im = i; }
    else { pt = props[i]; it = i; }
  }
  const bool ev_m2t = evp.mode != ICP_TARGET_TO_MODEL, ev_t2m = evp.mode != ICP_MODEL_TO_TARGET;
  const int Ksurf = std::max(ev_m2t ? evp.n_model_ids : 0, pm ? pm->K : 0);
  F.Ksurf = Ksurf;
  require(Ksurf <= c.N, "model id count exceeds the number of model points");
  QueryBuffers qs = c.query_scratch(Ksurf, c.target.T);
  QueryBuffers qv{};
  if (pt) qv = c.query_scratch(pt->K, c.N, 1);

  SurfaceTask st_surf = make_surface_task(c.target.T, c.target.verts.p, c.target.tris.p, c.target.spheres.p, Ksurf, s.x.p,
                                          c.hint_surf.p, qs, s.surf_cp.p, s.surf_d2.p, s.surf_tri.p);
  VertexTask st_vert{};
  if (pt) st_vert = make_vertex_task(c.N, s.x.p, pt->K, pt->target_pts.p, pt->hint_nn.p, qv, nullptr, pt->nn_id.p);
  st_vert.thr2 = nullptr;  // bounds are computed by the filter launch itself (see vertex_filter)
  // the evaluator's reverse direction (IndependentPointDistanceEvaluator.scala:49-54, Collective…Evaluator.scala:55-64): its
  // decimated-target points against the surface of the NEW instance — spheres and bounds are taken by the filter launch itself
  SurfaceTask st_t2m{};
  if (ev_t2m) {
    QueryBuffers qt = c.query_scratch(e->Kt, c.T, 2);
    st_t2m = make_surface_task(c.T, s.x.p, c.tris.p, nullptr, e->Kt, e->d_tpts, e->hint_tri.p, qt, e->t2m_cp.p, e->t2m_d2.p, e->t2m_tri.p);
    st_t2m.order = c.tri_order.p;
    st_t2m.thrA = nullptr;
  }

  // 1: coefficients -> instance -> bounds
  StepBeginArgs b{};
  b.N = c.N; b.r = r; b.inst_blocks = (c.N + 255) / 256;
  b.Qp = c.Qp.p; b.ref = c.ref.p; b.mean = c.mean.p; b.pose = s.pose;
  b.propose = generator >= 0 ? 1 : 0;
  const double* src = generator >= 0 ? key : key + 10;
  // (the merged launches cover ranks whose factor fits LDS — step_finish_supported, about 116 — so the r host-drawn numbers
  // always travel inside the kernel arguments)
  require(r <= kStepInlineZ, "internal: merged step at a rank above the inline-argument limit");
  std::memcpy(b.zin, src, sizeof(double) * r);
  if (generator >= 0) {
    PosteriorEntry& g = *ec[generator];
    b.prop = ProposeIn{g.alpha.p, g.V.p, g.S.p, c.inv_sqrt_lambda.p, c.P.p, g.coeffs.p, nullptr, kSigma2,
                       props[generator]->prm.step_length, props[generator]->sampler == ICP_SAMPLER_CHOLESKY_ROOT};
    int t = 0;
    while (t < 6 && (r << (t + 1)) <= 256 && (r >> (t + 1)) >= 8) ++t;  // = matvec_tpr_log2(r, 256) of k_propose
    b.tpr_log2 = t;
  }
  b.n_out = 0;
  b.out[b.n_out++] = s.coeffs.p;
  for (int i = 0; i < n_props; ++i) b.out[b.n_out++] = ep[i]->coeffs.p;
  b.out[b.n_out++] = c.h_res + 16 + F.parity * kCoeffArea;
  b.x = s.x.p;
  b.has_surf = 1; b.surf = st_surf;
  b.has_vert = pt ? 1 : 0; b.vert = st_vert;
  b.zero2 = ev_t2m ? st_t2m.cnt : nullptr; b.n_zero2 = ev_t2m ? st_t2m.Kpad : 0;
  // (nothing to wait for before the first finish launch, nor when every step is on one stream)
  b.wait_flag = (c.last_back_seq > 0 && two_streams) ? c.d_done.p + 2 : nullptr;
  // test hook: the first launch waits for a word that never comes, times out, and the step is repeated unpipelined
  static const int starve_pipeline = dev_env("ICP_TEST_STARVE_PIPELINE") ? (1 << 24) : 0;
  b.wait_seq = c.last_back_seq + starve_pipeline;
  b.wait_error = c.h_wait_error;
  b.wait_ticks = c.profiling ? c.d_wait_ticks.p : nullptr;
  if (generator >= 0 && ec[generator]->done_value != 0) {
    b.wait2_flag = props[generator]->eig_words.p + ec[generator]->status_off / 3;
    b.wait2_seq = ec[generator]->done_value;
    // still in flight (the register-holding variant of launch 1 multiplies with the KL basis: not for the Cholesky-root sampler)
    b.hold_regs = *(volatile int*)(props[generator]->h_eig + ec[generator]->status_off / 3) == -1 && !b.prop.root;
  }
  launch_step_begin(F.stream, b);

  // 2 + 3: searches and correspondences
  StepSearchArgs q{};
  q.n_surf = ev_t2m ? 2 : 1; q.n_vert = pt ? 1 : 0;
  q.s[0] = st_surf;
  q.fstart[0] = 0; q.fstart[1] = filter_grid_blocks(st_surf.tblocks, st_surf.ksplit);
  q.rstart[0] = 0; q.rstart[1] = Ksurf;
  int nt = 1;  // tasks so far (surface tasks first)
  if (ev_t2m) {
    q.s[1] = st_t2m;
    q.fstart[2] = q.fstart[1] + filter_grid_blocks(st_t2m.tblocks, st_t2m.ksplit);
    q.rstart[2] = q.rstart[1] + e->Kt;
    nt = 2;
  }
  q.s_corr[0] = q.s_corr[1] = q.v_corr[0] = q.v_corr[1] = -1;
  int n_corr = 0;
  if (pm) {
    q.corr[n_corr] = CorrTask{pm->K, ep[im]->corr(), s.x.p, nullptr, c.target.boundary.p, nullptr, pm->prm.boundary_aware,
                              s.pose, c.ref.p, c.mean.p, c.tris.p, c.adj_off.p, c.adj.p};
    q.s_corr[0] = n_corr++;
  }
  if (pt) {
    q.v[0] = st_vert;
    q.fstart[nt + 1] = q.fstart[nt] + filter_grid_blocks(st_vert.vblocks, st_vert.ksplit);
    q.rstart[nt + 1] = q.rstart[nt] + pt->K;
    q.corr[n_corr] = CorrTask{pt->K, ep[it]->corr(), s.x.p, pt->target_pts.p, c.boundary.p, nullptr, pt->prm.boundary_aware,
                              s.pose, c.ref.p, c.mean.p, c.tris.p, c.adj_off.p, c.adj.p};
    q.v_corr[0] = n_corr++;
  }
  launch_step_filter(F.stream, q);
  launch_step_resolve(F.stream, q);

  // 4: regressions + likelihood reduction
  StepRegressionArgs g{};
  g.n = n_props; g.r = r;
  g.ntiles = regression_tiles(r);
  g.Q = c.Q.p;
  g.ustart[0] = 0;
  int* splits = F.splits;
  for (int i = 0; i < n_props; ++i) {
    icp_proposal* p = props[i];
    splits[i] = regression_splits(p->K);
    g.K[i] = p->K;
    g.kchunk[i] = std::max(1, (p->K + splits[i] - 1) / splits[i]);
    g.cb[i] = ep[i]->corr();
    g.wt[i] = 1.0 / (p->prm.tangential_noise * p->prm.tangential_noise);
    g.kappa[i] = 1.0 / (p->prm.noise_along_normal * p->prm.noise_along_normal) - g.wt[i];
    p->mpart_half = (p->mpart_half + 1) % icp_proposal::kMpartRing;
    g.Mpart[i] = p->mpart_for_write(p->mpart_half, F.stream);
    g.status[i] = p->status.p + ep[i]->status_off;
    g.ustart[i + 1] = g.ustart[i] + g.ntiles * splits[i];
  }
  if (n_props == 1) g.ustart[2] = g.ustart[1];
  g.reduce_kind = evp.kind == ICP_EVAL_INDEPENDENT_POINT_DISTANCE ? 1 : 2;
  g.Kred = ev_m2t ? evp.n_model_ids : 0; g.d2 = s.surf_d2.p; g.mean = evp.gauss_mean; g.sigma = evp.gauss_sigma;
  g.Kred2 = ev_t2m ? e->Kt : 0; g.d2b = e->t2m_d2.p;
  g.red_out = c.h_res + kReduceArea + F.parity * 8;  // (its own half: the host may still be reading the previous step's)
  for (int i = 0; i < n_props; ++i) { F.mpart[i] = g.Mpart[i]; F.mpart_half[i] = props[i]->mpart_half; }
  launch_step_regression(F.stream, g);
  if (!batched) {  // 4b: many partials are summed by many CUs before the one-workgroup factorisation (see launch_step_reduce)
    bool many = false;
    for (int i = 0; i < n_props; ++i) many = many || splits[i] >= kStepReduceSplits;
    if (many) {
      StepReduceArgs ra{};
      ra.n = n_props; ra.nn = (r + 1) * (r + 1);
      for (int i = 0; i < n_props; ++i) { ra.Mpart[i] = g.Mpart[i]; ra.splits[i] = splits[i]; splits[i] = 1; }
      launch_step_reduce(F.stream, ra);
    }
  }
}

// Host side of a merged step whose results have arrived in the context's pinned memory: status of the decomposition it
// drew from, the proposed state, the memo entries, the rare direct transition tail, likelihood and densities.
// -> false: the step has to be done again (nothing of it has been recorded)
bool chain_step_record(icp_evaluator* e, int n_props, icp_proposal* const* props, int generator, const double* theta_cur, StepFront& F,
                       const StepFinishArgs& f, double* theta_prop, double* log_value_prop, double* fwd, double* bwd, int* status) {
  icp_ctx& c = *e->ctx;
  const int r = c.r;
  PosteriorEntry** ec = F.ec;
  PosteriorEntry** ep = F.ep;
  StateSlot& s = *F.s;
  const int Ksurf = F.Ksurf;
  const bool eigen_first_use = F.eigen_first_use;
  const bool eigen_status_pinned = eigen_speculation_supported(r);
  const double* h_coeffs = c.h_res + 16 + F.parity * kCoeffArea;
  if (eigen_first_use && eigen_status_pinned) {  // this step's first launch waited for that decomposition: its status is in
    icp_proposal* p = props[generator];
    const int st = p->h_eig[ec[generator]->status_off / 3];
    if (st == kEigenGaveUp) {  // a speculative decomposition that never saw its input (see k_posterior_eigen_rr): the
      // step just computed drew from a stale basis — drop it (nothing of it has been recorded) and do it again
      ++c.stats.speculation_giveups; ++g_runtime_stats.speculation_giveups;
      ec[generator]->eig_valid = false;
      ec[generator]->eig_checked = false;
      p->warm_valid = false;  // (it pointed at the basis that was never written)
      return false;
    }
    p->h_status[ec[generator]->status_off + 2] = st;
  }
  const size_t P = 10 + (size_t)r;
  if (generator >= 0) {
    std::memcpy(theta_prop, theta_cur, sizeof(double) * 10);  // :64-66 only the shape changes
    for (int j = 0; j < r; ++j) {
      if (!std::isfinite(h_coeffs[j])) fail(ICP_ERR_NOT_FINITE, "proposed coefficients are not finite");
      theta_prop[10 + j] = h_coeffs[j];
    }
  }
  s.theta.assign(theta_prop, theta_prop + P);
  s.valid = true;
  s.stamp = ++c.clock;
  s.n_surf = Ksurf;
  for (int i = 0; i < n_props; ++i) {
    icp_proposal* p = props[i];
    ep[i]->theta.assign(theta_prop, theta_prop + P);
    ep[i]->valid = true;
    ep[i]->stamp = ++p->clock;
    p->h_status[ep[i]->status_off] = c.h_status[8 + i];
    p->h_status[ep[i]->status_off + 1] = 0;
    p->h_status[ep[i]->status_off + 2] = 0;
    p->check_status(*ec[i]);
    p->check_status(*ep[i]);
  }
  for (int t = 0; t < 2 * n_props; ++t)
    if (c.h_status[t] != 0) {  // rare: the fixed-point tail did not contract -> direct kernel
      std::vector<double> saved(c.h_res, c.h_res + 16);
      icp_proposal* p = props[t / 2];
      TransitionTailIO io = (t % 2 == 0) ? f.fwd[t / 2] : f.bwd[t / 2];
      io.out = c.d_res.p;
      io.status = c.d_status.p + 32;
      sync_eigen(c);  // (the direct form borrows the eigen work buffer)
      launch_transition_tail_direct(c.stream, r, io, c.G.p, kSigma2, p->work.p);
      c.finish(1, 64);
      if (c.h_status[32] != 0) fail(ICP_ERR_NOT_SPD, "G + sigma^2 M is not positive definite");
      saved[8 + t] = c.h_res[0];
      std::memcpy(c.h_res, saved.data(), sizeof(double) * saved.size());
    }
  for (int i = 0; i < 8; ++i) c.h_res[i] = c.h_res[kReduceArea + F.parity * 8 + i];  // launch 4's reductions, where finish_eval looks
  icp_evaluator::Memo* m = eval_store(e, theta_prop);
  m->status = finish_eval(e, c.h_res, &m->value, m->aux);
  *log_value_prop = m->value;
  *status = m->status;
  for (int i = 0; i < n_props; ++i) {
    fwd[i] = c.h_res[8 + 2 * i];
    bwd[i] = c.h_res[9 + 2 * i];
    if (std::isnan(fwd[i]) || std::isnan(bwd[i])) fail(ICP_ERR_NOT_FINITE, "NaN transition probability");
  }
  return true;
}

// ICP_SPECULATION: 0 never, 1 always, unset = adaptive (2): while the chain's running acceptance rate is high

bool front_matches(const StepFront& F, int n_props, icp_proposal* const* props, int generator, const double* theta_cur,
                   const double* key, int r) {
  if (!F.valid || F.n_props != n_props || F.generator != generator) return false;
  for (int i = 0; i < n_props; ++i)
    if (F.props[i] != props[i]) return false;
  return std::memcmp(F.theta_cur.data(), theta_cur, sizeof(double) * (10 + r)) == 0 &&
         std::memcmp(F.key.data(), key, sizeof(double) * F.key.size()) == 0;
}

bool wide_chain_covered(icp_evaluator* e, int n_props, icp_proposal* const* props, int generator, const double* theta_cur,
                        const double* theta_prop_in);  // (the wide step: further down)

// shared argument checks of icp_chain_step and icp_chain_step_prelaunch; -> the merged launches cover this call
bool chain_step_covered(icp_evaluator* e, int n_props, icp_proposal* const* props, int generator, const double* theta_cur,
                        const double* theta_prop_in) {
  icp_ctx& c = *e->ctx;
  bool per_stage = !step_pipeline_covers(e, n_props, props);
  if (!per_stage && generator < 0) {
    // a state the caches already know (or a pose move, whose ICP transition densities are -inf) has nothing to merge
    per_stage = !pose_equal(theta_cur, theta_prop_in) || c.find_state(theta_prop_in) || eval_lookup(e, theta_prop_in);
    for (int i = 0; i < n_props && !per_stage; ++i) per_stage = props[i]->find_entry(theta_prop_in) != nullptr;
  }
  return !per_stage;
}

}  // namespace

extern "C" {

int icp_chain_step_prelaunch(icp_evaluator* e, int32_t n_props, icp_proposal* const* props, int32_t generator, const double* theta_cur,
                             const double* z_or_theta_prop) {
  return guard([&] {
    require(e != nullptr, "null argument");
    if (n_props == 0) {  // "nothing further": drop a pending half step
      std::lock_guard<std::recursive_mutex> lk0(e->ctx->mu);
      drop_front(e);
      return;
    }
    require(theta_cur && z_or_theta_prop, "null argument");
    require(n_props >= 1 && n_props <= 2 && props, "bad proposal list");
    require(generator < n_props, "generator index out of range");
    icp_ctx& c = *e->ctx;
    for (int i = 0; i < n_props; ++i) require(props[i] && props[i]->ctx == &c, "proposal belongs to another context");
    check_theta_finite(&c, theta_cur);
    if (generator < 0) check_theta_finite(&c, z_or_theta_prop);
    else
      for (int j = 0; j < c.r; ++j)
        if (!std::isfinite(z_or_theta_prop[j])) fail(ICP_ERR_NOT_FINITE, "z contains a non-finite value");
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    drop_front(e);
    if (c.pipeline_off) return;
    // with several chains in the process the device is not idle during one chain's turn-around, and the launches of a
    // dropped half step cost the others host time (tools/multichain.py)
    if (g_live_contexts.load(std::memory_order_relaxed) > 1) return;
    if (!chain_step_covered(e, n_props, props, generator, theta_cur, z_or_theta_prop)) return;  // nothing to pre-launch
    // every posterior of the assumed current state must be on record already (the step in flight computed them)
    for (int i = 0; i < n_props; ++i)
      if (!props[i]->find_entry(theta_cur)) return;
    Bound _b(&c, true);
    try {
      enqueue_front(e, n_props, props, generator, theta_cur, z_or_theta_prop, e->front);
    } catch (...) {
      release_front(e->front);
      throw;
    }
  });
}

int icp_chain_step(icp_evaluator* e, int32_t n_props, icp_proposal* const* props, int32_t generator, const double* theta_cur,
                   const double* z, double* theta_prop, double* log_value_prop, double* fwd, double* bwd) {
  int status = ICP_OK;
  bool per_stage = false, redo = false, wide = false;
  static thread_local int wide_depth = 0;  // (a wide step that has to be repeated comes back through this entry point: bounded)
  int rc = guard([&] {
    require(e && theta_cur && theta_prop && log_value_prop, "null argument");
    require(n_props >= 0 && n_props <= 8 && (n_props == 0 || (props && fwd && bwd)), "bad proposal list");
    require(generator < n_props, "generator index out of range");
    require(generator < 0 || z, "z is null");
    icp_ctx& c = *e->ctx;
    for (int i = 0; i < n_props; ++i) require(props[i] && props[i]->ctx == &c, "proposal belongs to another context");
    check_theta_finite(&c, theta_cur);
    if (generator < 0) check_theta_finite(&c, theta_prop);
    else
      for (int j = 0; j < c.r; ++j)
        if (!std::isfinite(z[j])) fail(ICP_ERR_NOT_FINITE, "z contains a non-finite value");
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    const int r = c.r;
    const double* key = generator >= 0 ? z : theta_prop;
    const bool reuse = n_props <= 2 && front_matches(e->front, n_props, props, generator, theta_cur, key, r);
    if (e->front.valid && !reuse) drop_front(e);  // pre-launched for another outcome: dropped
    per_stage = !reuse && !chain_step_covered(e, n_props, props, generator, theta_cur, theta_prop);
    if (per_stage) {
      // what the five merged launches do not cover as a configuration takes the wide step (a batch of one chain)
      wide = wide_depth < 2 && n_props >= 1 && n_props <= 2 && !step_pipeline_covers(e, n_props, props) &&
             wide_chain_covered(e, n_props, props, generator, theta_cur, theta_prop);
      return;
    }
    Bound _b(&c, true);
    g_host_timing.start();

    for (int i = 0; i < n_props; ++i) props[i]->resolve_speculation(theta_cur);
    if (!e->last_prop.empty()) {  // did the caller keep the state the previous step proposed?
      const bool accepted = std::memcmp(e->last_prop.data(), theta_cur, sizeof(double) * (10 + (size_t)r)) == 0;
      e->acc_ema = 0.9 * e->acc_ema + (accepted ? 0.1 : 0.0);
    }
    StepFront F;
    struct FrontGuard {  // whatever happens below, the slots of this step are not left reserved
      StepFront* f;
      ~FrontGuard() { if (f) release_front(*f); }
    } front_guard{&F};
    if (reuse) { F = e->front; e->front = StepFront{}; }
    else enqueue_front(e, n_props, props, generator, theta_cur, key, F);
    g_host_timing.mark(0);
    PosteriorEntry** ec = F.ec;
    PosteriorEntry** ep = F.ep;
    StateSlot& s = *F.s;
    const bool eigen_first_use = F.eigen_first_use;
    // the decompositions of ranks <= 64 leave their status in pinned memory themselves; the others need a copy
    const bool eigen_status_pinned = eigen_speculation_supported(r);
    const bool eigen_enqueued = eigen_first_use && !eigen_status_pinned;
    for (int i = 0; i < 16; ++i) c.h_res[i] = 0.0;
    for (int i = 0; i < 16; ++i) c.h_status[i] = 0;

    // Adaptive (speculation_mode): an accepted step finds its basis ≈ 50 µs earlier; a rejected one has paid one launch
    // (≈ 3 µs of host time, a few CUs for at most one sweep) for nothing — worth it unless next to nothing is accepted
    // (measured: 8.3k against 7.4k it/s over a chain's first 20 steps, 13.6k against 13.7k at one acceptance in three).
    const int spec_mode = speculation_mode();
    const bool speculate = (spec_mode == 1 || (spec_mode == 2 && e->acc_ema >= 0.1)) && !c.speculation_off &&
                           g_live_contexts.load(std::memory_order_relaxed) <= 2 && n_props > 0 && eigen_speculation_supported(r);
    // test hook: the speculative decompositions wait for a word that never comes, time out and are repeated
    static const int starve = dev_env("ICP_TEST_STARVE_SPECULATION") ? (1 << 24) : 0;
    const int step_seq = ++c.step_seq;

    // 5: factorisations + tails (results go straight to pinned host memory)
    StepFinishArgs f{};
    f.n = n_props; f.r = r; f.Ginv = c.Ginv.p; f.sigma2 = kSigma2;
    for (int i = 0; i < n_props; ++i) {
      icp_proposal* p = props[i];
      f.Mpart[i] = F.mpart[i]; f.splits[i] = F.splits[i];
      f.M[i] = ep[i]->M.p; f.alpha[i] = ep[i]->alpha.p;
      f.status[i] = p->status.p + ep[i]->status_off;
      f.host_status[i] = c.h_status + 8 + i;
      f.fwd[i] = TransitionTailIO{ec[i]->alpha.p, ec[i]->M.p, ec[i]->coeffs.p, ep[i]->coeffs.p, p->prm.step_length,
                                  c.h_res + 8 + 2 * i, c.h_status + 2 * i};
      f.bwd[i] = TransitionTailIO{ep[i]->alpha.p, ep[i]->M.p, ep[i]->coeffs.p, ec[i]->coeffs.p, p->prm.step_length,
                                  c.h_res + 9 + 2 * i, c.h_status + 2 * i + 1};
    }
    f.done_counter = c.d_done.p; f.host_flag = c.h_flag; f.seq = step_seq;
    f.ready_flag = c.d_done.p + 2;  // (speculative decompositions and the next step's first launches wait for it)
    c.last_back_seq = step_seq;
    launch_step_finish(F.stream, f);
    g_host_timing.mark(1);
    // the caller's outcome-independent host work runs beside the device — first of all the pre-launch of the next step's
    // first half, which the device can start as soon as the finish launch above has
    // KL bases of the proposed state's posteriors, in case it is accepted: they run on the eigen stream beside the
    // factorisations and the host's round trip; the next call keeps or cancels them (resolve_speculation).  One launch (both
    // directions side by side), issued BEFORE the caller's hook: the accepted path waits for nothing else.
    if (speculate) {
      EigenSpec specs[2];
      EigenRequest rqs[2];
      for (int i = 0; i < n_props; ++i) props[i]->speculate_eigen(*ep[i], *ec[i], F.splits[i], F.mpart_half[i], c.d_done.p + 2, step_seq + starve, &specs[i], &rqs[i]);
      // (developer switch: the tridiagonal route for these decompositions while the running acceptance rate is above a threshold —
      // its time does not depend on how far the chain has moved, the warm-started iteration's does (124 µs on average over a chain's
      // first steps against 84 in the steady state); measured: 7.2k against 8.7k it/s over the first 20 steps — from input to
      // completion word the two launches take ≈ 110 µs, the iteration with its replay beside it ≈ 100 even at four sweeps.)
      static const double direct_above = dev_env("ICP_DIRECT_ABOVE") ? std::atof(dev_env("ICP_DIRECT_ABOVE")) : 2.0;
      for (int i = 0; i < n_props; ++i) rqs[i].direct = e->acc_ema >= direct_above;
      const hipStream_t es = eigen_stream_for(c, c.eig_stream);
      launch_posterior_eigen_pair(es, r, c.sqrt_lambda.p, n_props, rqs);  // (no event: completion words, see start_decompositions)
    }
    // the caller's outcome-independent host work runs beside the device — first of all the pre-launch of the next step's
    // first half (under the rejection assumption), which the device can start as soon as the finish launch above has
    if (c.idle_fn) c.idle_fn(c.idle_arg);
    g_host_timing.mark(2);
    if (eigen_enqueued) {
      if (F.stream != c.stream) HIP_OK(hipStreamSynchronize(F.stream));
      sync_proposal_status_if(props[generator], true);
      c.finish(0, 0);
    } else {
      // results and flag are written into pinned memory by the kernels: poll the flag (≈ 4 µs less than a stream
      // synchronisation); give up after 2 s and let the synchronisation report what went wrong
      volatile int* flag = c.h_flag;
      const auto t_start = std::chrono::steady_clock::now();
      long spins = 0;
      while (*flag != f.seq) {
        if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(2)) break;
      }
      if (*flag != f.seq) {
        if (F.stream != c.stream) HIP_OK(hipStreamSynchronize(F.stream));
        c.finish(0, 0);
      }
      c.stage_used = 0;
    }

    // ---- bookkeeping with the results in hand
    g_host_timing.mark_wait(eigen_first_use);
    if (c.h_wait_error[0]) {
      // A first launch did not see the word it waits for within 50 ms and went ahead unordered.  That is what a tool does
      // that lets one kernel run at a time in an order of its own (rocprofv3 --pmc): drain everything, switch the pipelining
      // off for this context, and do the step again — nothing of it has been recorded.
      HIP_OK(hipStreamSynchronize(c.stream));
      HIP_OK(hipStreamSynchronize(c.front_stream));  // (a half step launched ahead may time out here, too)
      sync_eigen(c);
      c.h_wait_error[0] = 0;
      ++c.stats.wait_timeouts; ++g_runtime_stats.wait_timeouts;
      if (c.pipeline_off) fail(ICP_ERR_DEVICE, "internal: a step's first launch timed out on its word");
      c.pipeline_off = true;
      ++c.stats.pipeline_fallbacks; ++g_runtime_stats.pipeline_fallbacks;
      ++c.stats.step_redos; ++g_runtime_stats.step_redos;
      if (e->front.valid) release_front(e->front);
      redo = true;
      return;
    }
    if (!chain_step_record(e, n_props, props, generator, theta_cur, F, f, theta_prop, log_value_prop, fwd, bwd, &status)) {
      ++c.stats.step_redos; ++g_runtime_stats.step_redos;
      redo = true;
      return;
    }
    s.reserved = false;
    for (int i = 0; i < n_props; ++i) ep[i]->reserved = false;
    front_guard.f = nullptr;
    ++c.paths.n[0]; ++g_step_paths.n[0];
    e->last_prop.assign(theta_prop, theta_prop + 10 + r);
    g_host_timing.mark(4);
    g_host_timing.end();
  });
  if (rc != ICP_OK) return rc;
  if (redo) return icp_chain_step(e, n_props, props, generator, theta_cur, z, theta_prop, log_value_prop, fwd, bwd);
  if (per_stage && wide) {
    int32_t gen = generator, st = ICP_OK;
    const double* tc = theta_cur;
    const double* zz = z;
    double* tp = theta_prop;
    ++wide_depth;
    rc = icp_chain_step_batched(1, &e, n_props, props, &gen, &tc, generator >= 0 ? &zz : nullptr, &tp, log_value_prop, fwd, bwd, &st);
    --wide_depth;
    return rc != ICP_OK ? rc : st;
  }
  if (per_stage) {  // same results through the per-stage kernels
    ++e->ctx->paths.n[2]; ++g_step_paths.n[2];
    if (generator >= 0) {
      rc = icp_proposal_propose(props[generator], theta_cur, z, theta_prop, nullptr);
      if (rc != ICP_OK) return rc;
    }
    return icp_chain_eval_step(e, n_props, props, theta_cur, theta_prop, log_value_prop, fwd, bwd);
  }
  return status;
}

// B chains per launch.  Every chain takes the merged step of icp_chain_step with its own context's buffers; the five
// launches are recorded per chain (StepCapture) and issued ONCE for all of them on the first chain's stream, the
// decompositions of chains that moved run on their own contexts' eigen streams beside it (launch 1 waits for each on the
// device, as in the single-chain step).  Chains this does not cover (another device or rank than the first chain's, a
// context that already has a chain in the batch, a configuration the merged launches do not cover) take icp_chain_step
// one after the other, behind the batch.
// The work is split in two so that a caller can keep two batches in flight (the decompositions of one run beside the
// launches of the other): _issue ends when everything is on the device, _collect waits and records.
} // extern "C" (helpers)

// StepRandom::normal of the C++ harness (host/icp_host.hpp; = orc_rng_normal of the oracle): Box–Muller over the counter-based uniforms
static inline uint64_t harness_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static inline double harness_uniform(uint64_t seed, uint64_t step, uint64_t lane) {
  const uint64_t h = harness_splitmix64(harness_splitmix64(harness_splitmix64(seed) ^ (step * 0xD1342543DE82EF95ull)) ^ (lane * 0x2545F4914F6CDD1Dull));
  return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
static inline double harness_normal(uint64_t seed, uint64_t step, uint64_t lane) {
  const double u1 = harness_uniform(seed, step, 2 * lane + 1000), u2 = harness_uniform(seed, step, 2 * lane + 1001);
  return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
}

// ===================================================================== the wide step (kernels_wide.hip; icp_kernels.hpp "the wide step")
// One Metropolis–Hastings step of the configurations the five merged launches do not cover — reference:
// apps/bfm/BfmFittingPartial.scala:62-83 (open target, boundary-aware ModelSampling, collective / full-mesh Hausdorff evaluator, rank
// 200, pose walks) — without a host round trip inside the step and for B chains per launch sequence.  Host side per chain: the
// choices enqueue_front makes (posterior entries of the current and of the proposed state, the proposed state's slot), the arguments
// of every launch; then ONE sequence of launches for all chains on the launch context's stream, the one-workgroup kernels
// (factorisation, tails) on its second stream, the proposed states' decompositions on the batch's eigen stream.
namespace {

struct WideItem {
  bool on = false;          // the chain takes the wide step of this ticket
  bool shape_only = false;  // only the shape differs between the current and the proposed state (transition densities exist)
  bool do_post = false;     // the proposed state's posteriors are computed (always for a shape move; ahead, for a pose move)
  bool do_spec = false;     // … and decomposed ahead
  bool eigen_first_use = false;
  StateSlot* s = nullptr;
  PosteriorEntry* ec[2] = {nullptr, nullptr};
  PosteriorEntry* ep[2] = {nullptr, nullptr};
  int Ksurf = 0, Knnv = 0;
  bool spheres = false;     // the new instance's bounding spheres are made (a target -> model search)
  int seq = 0;
  int n_tails = 0;
  TransitionTailIO tails[4];  // fwd_0, bwd_0, fwd_1, bwd_1 (as recorded: for the rare direct-tail fall-back)
};

// the rank-dependent kernels of a wide step exist for this rank and sampler
bool wide_rank_covered(int r, int sampler) {
  if (r < 3 || r > 256) return false;
  if (sampler == ICP_SAMPLER_CHOLESKY_ROOT) return r <= kCholMaxRankAbi;
  return eigen_speculation_supported(r) || eigen_tridiag_many_supported(r);
}

// the configuration (proposal set + evaluator) is one a wide step covers
bool wide_pipeline_covers(icp_evaluator* e, int n_props, icp_proposal* const* props) {
  icp_ctx& c = *e->ctx;
  if (n_props < 1 || n_props > 2) return false;
  const icp_evaluator_params& ep = e->prm;
  const bool hd = ep.kind == ICP_EVAL_HAUSDORFF;
  const bool m2t = hd || ep.mode != ICP_TARGET_TO_MODEL, t2m = hd || ep.mode != ICP_MODEL_TO_TARGET;
  const int Km = hd ? c.N : ep.n_model_ids;
  if (m2t && Km < 1) return false;
  if (t2m && (e->Kt < 1 || c.T < 1)) return false;
  if (c.target.T < 1 || c.target.V < 1) return false;
  int n_model = 0, n_target = 0, ksurf = m2t ? Km : 0;
  for (int i = 0; i < n_props; ++i) {
    const icp_proposal* p = props[i];
    if (p->K < 1) return false;
    if (!wide_rank_covered(c.r, p->sampler)) return false;
    if (p->sampler != props[0]->sampler) return false;
    if (p->prm.direction == ICP_MODEL_SAMPLING) { ++n_model; ksurf = std::max(ksurf, p->K); }
    else {
      ++n_target;
      if ((size_t)(p->K + 8) * (size_t)kCandStride > kMaxCandidates) return false;
    }
  }
  if (n_model > 1 || n_target > 1) return false;
  if ((size_t)(ksurf + 8) * (size_t)kCandStride > kMaxCandidates) return false;
  if (t2m && (size_t)(e->Kt + 8) * (size_t)kCandStride > kMaxCandidates) return false;
  return true;
}

// … and so is this call (a proposed state the caches already know has nothing to compute: the per-stage entry points answer it)
bool wide_chain_covered(icp_evaluator* e, int n_props, icp_proposal* const* props, int generator, const double* theta_cur,
                        const double* theta_prop_in) {
  if (!wide_pipeline_covers(e, n_props, props)) return false;
  if (generator < 0) {
    icp_ctx& c = *e->ctx;
    if (c.find_state(theta_prop_in) || eval_lookup(e, theta_prop_in)) return false;
    for (int i = 0; i < n_props; ++i)
      if (props[i]->find_entry(theta_prop_in)) return false;
  }
  return true;
}

}  // namespace

struct BatchItem {
  icp_evaluator* e = nullptr;
  icp_proposal* const* props = nullptr;
  int generator = -1;
  const double* key = nullptr;
  bool batched = false, issued = false, redo = false;
  bool wide = false;  // takes the wide step (kernels_wide.hip) instead of the five merged launches
  WideItem W;
  StepFront F;
  StepFinishArgs f{};
  std::unique_lock<std::recursive_mutex> lk;
};

struct icp_step_ticket {
  int n_chains = 0, n_props = 0, nb = 0;
  icp_ctx* lead = nullptr;
  bool counted = false;  // included in lead->tickets_in_flight
  hipStream_t finish_stream = nullptr;  // where the batch's last launch went, if not lead->stream
  hipStream_t wide_streams[2] = {nullptr, nullptr};  // the streams of the ticket's wide step, if it has one
  std::vector<BatchItem> items;
  std::vector<StepCapture> caps;
  std::vector<icp_proposal*> props;
  std::vector<const double*> theta_cur, z;
  std::vector<double*> theta_prop;
  double* log_value_prop = nullptr;
  double* fwd = nullptr;
  double* bwd = nullptr;
  int32_t* status = nullptr;
};

namespace {
void wide_release(BatchItem& it, bool recorded = false);
// whatever happened, nothing stays reserved or locked; a failed batch leaves its launches to drain
void batch_release(icp_step_ticket& t) {
  if (t.counted && t.lead) { --t.lead->tickets_in_flight; t.counted = false; }
  for (auto& it : t.items) {
    if ((!it.batched && !it.wide) || !it.e) continue;
    icp_ctx& c = *it.e->ctx;
    if (!it.lk.owns_lock()) it.lk = std::unique_lock<std::recursive_mutex>(c.mu);
    if (it.issued) {
      (void)hipSetDevice(c.device);
      if (t.lead) (void)hipStreamSynchronize(t.lead->stream);
      if (t.finish_stream) (void)hipStreamSynchronize(t.finish_stream);
      for (hipStream_t ws : t.wide_streams)
        if (ws) (void)hipStreamSynchronize(ws);
      if (it.wide) wide_release(it);
      else release_front(it.F);
      it.issued = false;
    } else if (it.wide) {
      wide_release(it);
    }
    c.batch_busy = false;
    it.lk.unlock();
  }
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// the wide step's host side: wide_issue (everything onto the device) / wide_collect (results, bookkeeping), called by
// icp_chain_step_batched_issue / _collect for the items marked `wide`
namespace {

void wide_release(BatchItem& it, bool recorded) {  // gives back what the item holds (recorded: its step has been booked)
  WideItem& w = it.W;
  if (w.s) w.s->reserved = false;
  for (int i = 0; i < 2; ++i)
    if (w.ep[i]) w.ep[i]->reserved = false;
  if (!recorded && w.on && w.eigen_first_use && it.generator >= 0 && w.ec[it.generator]) w.ec[it.generator]->eig_checked = false;
  w.on = false; w.s = nullptr;
  w.ec[0] = w.ec[1] = w.ep[0] = w.ep[1] = nullptr;
}

// `S` waits for the decomposition `en` may still be the subject of (an event), or the host does (none on record)
void wide_await_entry(icp_ctx& c, icp_proposal* p, PosteriorEntry& en, hipStream_t S, std::vector<hipEvent_t>& waited) {
  // (a finished decomposition has left its status in pinned memory, −1 while in flight: nothing to wait for then)
  if (!en.eig_valid || *(volatile int*)(p->h_eig + en.status_off / 3) != -1) return;
  if (hipEvent_t ev = en.eigen_event()) {
    if (std::find(waited.begin(), waited.end(), ev) == waited.end()) {
      HIP_OK(hipStreamWaitEvent(S, ev, 0));
      waited.push_back(ev);
    }
  } else if (en.eig_valid && (en.eig_event_valid || en.done_value != 0)) {
    sync_eigen(c);  // (started by another kind of step without an event, or its event slot has been handed out again)
  }
}

void wide_issue(icp_step_ticket& t, icp_ctx& lead, icp_ctx& elead) {
  const int n_props = t.n_props;
  std::vector<int> idx;
  for (int b = 0; b < t.n_chains; ++b)
    if (t.items[b].wide) idx.push_back(b);
  if (idx.empty()) return;
  const int r = elead.r, nW = (int)idx.size();
  std::lock_guard<std::recursive_mutex> lead_lk(lead.mu);  // (its streams, its record ring)
  const hipStream_t S = lead.stream, S2 = lead.front_stream;
  // Two eigen streams: the decompositions of a chain's consecutive steps alternate between them (and between the proposal's two work
  // buffers), so that one started ahead for a state that was then not kept — or whose successor was a pose move that did not wait
  // for it — does not hold the next one back: at rank 200 a decomposition takes 0.6-0.7 ms, a step that does not wait for one half
  // of that.  A lone chain uses its context's own pair (made with the context, on hardware queues of their own), a batch the launch
  // context's pool.  (Ranks <= 64: one stream — the Jacobi iteration is warm-started from the decomposition before it.)
  const bool two_eig = eigen_tridiag_many_supported(r) && elead.eig_stream2 != nullptr;
  const bool lone = t.n_chains == 1 && &lead == &elead;
  hipStream_t Es[2];
  Es[0] = lone ? elead.eig_stream : batch_eigen_stream(lead, &elead, 0);
  Es[1] = !two_eig ? Es[0] : (lone ? elead.eig_stream2 : batch_eigen_stream(lead, &elead, 1));
  const int turn = (lead.wide_turn = (lead.wide_turn + 1) % icp_ctx::kBatchRing);
  {
    Bound _b(&lead, true, true);
    if (!lead.ev_wide_sum[turn]) HIP_OK(hipEventCreateWithFlags(&lead.ev_wide_sum[turn], hipEventDisableTiming));
    if (!lead.ev_wide_fac[turn]) HIP_OK(hipEventCreateWithFlags(&lead.ev_wide_fac[turn], hipEventDisableTiming));
    if (!lead.ev_wide_head[turn]) HIP_OK(hipEventCreateWithFlags(&lead.ev_wide_head[turn], hipEventDisableTiming));
    if (!lead.ev_wide_eval[turn]) HIP_OK(hipEventCreateWithFlags(&lead.ev_wide_eval[turn], hipEventDisableTiming));
    const size_t bytes = wide_batch_bytes(nW);
    if (bytes > lead.wide_bytes[turn]) {
      // (the slot's previous reader was the batch kBatchRing tickets ago: collected — tickets_in_flight —, its launches finished)
      if (lead.wide_pinned[turn]) { pinned_free(lead.wide_pinned[turn]); lead.wide_pinned[turn] = nullptr; }
      const size_t cap = std::max(bytes, wide_batch_bytes(kWideMaxChains));
      pinned_alloc((void**)&lead.wide_pinned[turn], cap);
      lead.wide_device[turn].alloc(cap);
      lead.wide_bytes[turn] = cap;
    }
  }
  std::vector<hipEvent_t> waited;
  std::vector<WideProposeItem> prop_items;
  std::vector<WideChainArgs> chain_args(nW);
  std::vector<double*> sum_parts; std::vector<int> sum_splits;
  std::vector<PosteriorFactorIO> factors;
  std::vector<PosteriorEntry*> root_entries;  // (Cholesky-root sampler above rank 64: the factorisation hands the "basis" out)
  std::vector<icp_proposal*> root_props;
  std::vector<TransitionTailIO> tails;
  std::vector<EigenRequest> spec_rq[2]; std::vector<const double*> spec_parts[2]; std::vector<PosteriorEntry*> spec_entries[2];
  std::vector<EigenRequest> pre_rq[2]; std::vector<PosteriorEntry*> pre_entries[2];
  std::vector<WideDoneItem> dones;
  WideLaunchPlan plan{};
  plan.B = nW; plan.N = elead.N; plan.r = r; plan.Qp = elead.Qp.p; plan.ref = elead.ref.p; plan.mean = elead.mean.p;
  plan.f1_prepared = true;
  bool any_split = false;
  const bool concurrent = t.items[idx[0]].e->prm.kind == ICP_EVAL_HAUSDORFF;  // (how a split step's two sequences are scheduled: below)
  const int spec_mode = speculation_mode();

  for (int k = 0; k < nW; ++k) {
    BatchItem& it = t.items[idx[k]];
    WideItem& w = it.W;
    icp_evaluator* e = it.e;
    icp_ctx& c = *e->ctx;
    const double* theta_cur = t.theta_cur[idx[k]];
    double* theta_prop = t.theta_prop[idx[k]];
    const int generator = it.generator;
    Bound _b(&c, true);
    w = WideItem{};
    w.on = true;
    if (!e->last_prop.empty()) {  // did the caller keep the state the previous step proposed?
      const bool accepted = std::memcmp(e->last_prop.data(), theta_cur, sizeof(double) * (10 + (size_t)r)) == 0;
      e->acc_ema = 0.9 * e->acc_ema + (accepted ? 0.1 : 0.0);
    }
    for (int i = 0; i < n_props; ++i) it.props[i]->resolve_speculation(theta_cur);  // (a merged step's speculation, if the chain changed paths)
    w.shape_only = generator >= 0 || pose_equal(theta_cur, theta_prop);
    const bool spec = (spec_mode == 1 || (spec_mode == 2 && e->acc_ema >= 0.1)) && !c.speculation_off;
    // a pose move changes the state too: if it is kept, the next ICP proposal draws from the posterior at the NEW state, which
    // nothing else on this path would compute (the transition densities across a pose change are −∞): started here, ahead
    w.do_post = w.shape_only || spec;
    w.do_spec = w.do_post && spec;
    const bool root = it.props[0]->sampler == ICP_SAMPLER_CHOLESKY_ROOT;
    const bool root_here = root && !eigen_speculation_supported(r);  // (the factorisation itself hands the factor out)

    // ---- cached side: the current state's posteriors (a shape move's tails and proposal read them)
    PosteriorEntry** ec = w.ec;
    PosteriorEntry** ep = w.ep;
    if (w.shape_only) {
      bool missing = false;
      for (int i = 0; i < n_props; ++i) missing = missing || !it.props[i]->find_entry(theta_cur);
      for (int i = 0; i < n_props; ++i) ec[i] = &it.props[i]->posterior(theta_cur, false);  // NonRigidIcpProposal.scala:54,76 (the per-stage way if not on record)
      if (missing || c.stream_used_elsewhere) { HIP_OK(hipStreamSynchronize(c.stream)); c.stream_used_elsewhere = false; c.stage_used = 0; }
      for (int i = 0; i < n_props; ++i) { ec[i]->reserved = true; }  // (not to be recycled for the proposed state's entries below)
    }
    if (generator >= 0) {
      icp_proposal* pg = it.props[generator];
      PosteriorEntry& g = *ec[generator];
      if (!g.eig_valid) {
        if (root_here) fail(ICP_ERR_DEVICE, "internal: a posterior of the Cholesky-root sampler without its factor");
        EigenRequest rq;
        pg->prepare_eigen(g, &rq);
        rq.sqrt_lambda = c.sqrt_lambda.p;
        g.eig_event_valid = false; g.eig_done_shared = nullptr; g.eig_shared_gen = nullptr;
        eigen_streams_for(c, Es[0], Es[1]);
        const int fl = two_eig ? (int)(pg->eig_flip++ & 1) : 0;
        if (fl) rq.work = pg->work2.p;
        pre_rq[fl].push_back(rq);
        pre_entries[fl].push_back(&g);
      } else {
        wide_await_entry(c, pg, g, S, waited);
      }
      w.eigen_first_use = !g.eig_checked;
      g.eig_checked = true;
    }

    // ---- new side: one state slot, one memo entry per proposal
    StateSlot* same = nullptr;  // a pose move: a state with these coefficients whose deformations are kept
    if (generator < 0 && !w.shape_only)
      for (auto& o : c.slots)
        if (o.valid && o.defo_valid && std::memcmp(o.theta.data() + 10, theta_prop + 10, sizeof(double) * r) == 0) { same = &o; break; }
    if (same) { same->stamp = ++c.clock; same->reserved = true; }
    StateSlot& s = c.fresh_state();
    if (same) same->reserved = false;
    s.reserved = true;
    w.s = &s;
    s.pose = c.pose_of(generator >= 0 ? theta_cur : theta_prop);
    if (w.do_post)
      for (int i = 0; i < n_props; ++i) {
        ep[i] = &it.props[i]->fresh_entry();
        ep[i]->reserved = true;
        wide_await_entry(c, it.props[i], *ep[i], S, waited);  // (a decomposition started ahead for a state that was not kept may still read / write it)
        ep[i]->eig_event_valid = false;
        ep[i]->done_value = 0;
      }
    if (w.shape_only)
      for (int i = 0; i < n_props; ++i) ec[i]->reserved = false;

    // ---- W1: coefficients of the proposed state
    WideProposeItem pi{};
    if (generator >= 0) {
      PosteriorEntry& g = *ec[generator];
      if (!c.h_wide_z) pinned_alloc((void**)&c.h_wide_z, sizeof(double) * kMaxRank);
      std::memcpy(c.h_wide_z, t.z[idx[k]], sizeof(double) * r);  // posterior.sample()'s standard normals (:55)
      pi.kind = 1;
      pi.in = ProposeIn{g.alpha.p, g.V.p, g.S.p, c.inv_sqrt_lambda.p, c.P.p, g.coeffs.p, c.h_wide_z, kSigma2,
                        it.props[generator]->prm.step_length, root ? 1 : 0};
    } else {
      if (!c.h_wide_z) pinned_alloc((void**)&c.h_wide_z, sizeof(double) * kMaxRank);
      std::memcpy(c.h_wide_z, theta_prop + 10, sizeof(double) * r);
      pi.kind = 0;
      pi.src = c.h_wide_z;
    }
    pi.n_out = 0;
    pi.out[pi.n_out++] = s.coeffs.p;
    if (w.do_post)
      for (int i = 0; i < n_props; ++i) pi.out[pi.n_out++] = ep[i]->coeffs.p;
    pi.out[pi.n_out++] = c.h_res + 16;
    prop_items.push_back(pi);

    // ---- searches
    const icp_evaluator_params& evp = e->prm;
    const bool hd = evp.kind == ICP_EVAL_HAUSDORFF, coll = evp.kind == ICP_EVAL_COLLECTIVE_AVG_HAUSDORFF_BOUNDARY_AWARE;
    const bool ev_m2t = hd || evp.mode != ICP_TARGET_TO_MODEL, ev_t2m = hd || evp.mode != ICP_MODEL_TO_TARGET;
    const int Km = hd ? c.N : evp.n_model_ids, Kt = e->Kt;
    icp_proposal* pm = nullptr; icp_proposal* pt = nullptr;
    int im = -1, itx = -1;
    if (w.do_post)
      for (int i = 0; i < n_props; ++i) {
        if (it.props[i]->prm.direction == ICP_MODEL_SAMPLING) { pm = it.props[i]; im = i; }
        else { pt = it.props[i]; itx = i; }
      }
    const bool open_target = c.target.n_boundary > 0;
    const int Ksurf = std::max(ev_m2t ? Km : 0, pm ? pm->K : 0);
    const bool prop_nnv = pm && pm->prm.boundary_aware && open_target;      // NonRigidIcpProposal.scala:98-99
    const bool eval_nnv = coll && open_target && ev_m2t;                     // Collective…Evaluator.scala:44-48
    const bool t2m_nnv = coll && open_target && ev_t2m;                      // :56-60
    const int Knnv = std::max(prop_nnv ? pm->K : 0, eval_nnv ? Km : 0);
    require(Ksurf <= c.N, "model id count exceeds the number of model points");
    w.Ksurf = Ksurf; w.Knnv = Knnv; w.spheres = ev_t2m;
    // Split: the PROPOSAL's chain — its K model ids -> their nearest vertices -> correspondences -> regression -> partial sums: what the
    // factorisation, the tails and the decomposition wait for — is the MAIN sequence; the evaluator's searches and reductions are a
    // sequence of their own behind (or beside) it:
    //   concurrent (the batch's evaluator is the full-mesh Hausdorff distance: every model vertex against the target surface, every
    //     target vertex against the model's, 0.2 ms of chip-wide searches): the evaluator's sequence on S BESIDE the main one on the
    //     second stream; the maxima are order-independent, each sequence reduces its own range (atomic maxima);
    //   serial (any other evaluator): main first, the evaluator's sequence behind it on the same stream — the side streams
    //     (factorisation + tails, decomposition) start as soon as the partial sums exist instead of behind every search of the step
    //     (10 chains of the face configuration: ≈ 0.2 ms earlier); the reductions run at the end, over all ids.
    const bool split = w.do_post && (concurrent ? hd : true);
    const int Kp = split ? (pm ? std::min(pm->K, Ksurf) : 0) : 0;
    any_split = any_split || split;
    int nnv_main = Knnv, nnv_lo = 0, nnv_hi = 0;  // nearest vertices: ids [0, nnv_main) by the main sequence, [nnv_lo, nnv_hi) by the evaluator's
    if (split) {
      nnv_main = prop_nnv ? pm->K : 0;
      if (eval_nnv) { nnv_lo = prop_nnv ? std::min(pm->K, Km) : 0; nnv_hi = Km; }
    }
    QueryBuffers qs{}, qv{}, qt{}, qn{}, qtn{}, qp{}, qen{};
    if (Ksurf - Kp > 0) qs = c.query_scratch(Ksurf - Kp, c.target.T, 0);
    if (split && Kp > 0) qp = c.query_scratch(Kp, c.target.T, 5);
    if (pt) qv = c.query_scratch(pt->K, c.N, 1);
    if (ev_t2m) qt = c.query_scratch(Kt, c.T, 2);
    if (nnv_main > 0) qn = c.query_scratch(nnv_main, c.target.V, 3);
    if (nnv_hi > nnv_lo) qen = c.query_scratch(nnv_hi - nnv_lo, c.target.V, 6);
    if (t2m_nnv) qtn = c.query_scratch(Kt, c.N, 4);

    WideChainArgs& A = chain_args[k];
    std::memset(&A, 0, sizeof(A));
    SurfaceTask st_surf{}, st_t2m{}, st_surfp{};
    VertexTask st_vert{}, st_nnv{}, st_ennv{}, st_tnn{};
    if (Ksurf - Kp > 0)  // (ids Kp..Ksurf; Kp = 0 unless the evaluator has a sequence of its own)
      st_surf = make_surface_task(c.target.T, c.target.verts.p, c.target.tris.p, c.target.spheres.p, Ksurf - Kp, s.x.p + 3 * (size_t)Kp,
                                  c.hint_surf.p + Kp, qs, s.surf_cp.p + 3 * (size_t)Kp, s.surf_d2.p + Kp, s.surf_tri.p + Kp);
    if (split && Kp > 0)
      st_surfp = make_surface_task(c.target.T, c.target.verts.p, c.target.tris.p, c.target.spheres.p, Kp, s.x.p, c.hint_surf.p, qp,
                                   s.surf_cp.p, s.surf_d2.p, s.surf_tri.p);
    if (ev_t2m)
      st_t2m = make_surface_task(c.T, s.x.p, c.tris.p, s.spheres.p, Kt, e->d_tpts, e->hint_tri.p, qt, e->t2m_cp.p, e->t2m_d2.p, e->t2m_tri.p);
    if (pt) { st_vert = make_vertex_task(c.N, s.x.p, pt->K, pt->target_pts.p, pt->hint_nn.p, qv, nullptr, pt->nn_id.p); st_vert.thr2 = nullptr; }
    if (nnv_main > 0) { st_nnv = make_vertex_task(c.target.V, c.target.verts.p, nnv_main, s.surf_cp.p, c.hint_nnv.p, qn, nullptr, s.surf_nnv.p); st_nnv.thr2 = nullptr; }
    if (nnv_hi > nnv_lo) {
      st_ennv = make_vertex_task(c.target.V, c.target.verts.p, nnv_hi - nnv_lo, s.surf_cp.p + 3 * (size_t)nnv_lo, c.hint_nnv.p + nnv_lo, qen, nullptr,
                                 s.surf_nnv.p + nnv_lo);
      st_ennv.thr2 = nullptr;
    }
    if (t2m_nnv) { st_tnn = make_vertex_task(c.N, s.x.p, Kt, e->t2m_cp.p, e->hint_nnv.p, qtn, nullptr, e->t2m_nnv.p); st_tnn.thr2 = nullptr; }

    // W2
    A.inst.kind = same ? 1 : 0;
    A.inst.coeffs = s.coeffs.p;
    A.inst.defo_src = same ? same->defo.p : nullptr;
    A.inst.pose = s.pose;
    A.inst.x = s.x.p; A.inst.defo = s.defo.p;
    if (split) {
      A.inst.has_surf = Kp > 0 ? 1 : 0; A.inst.surf = st_surfp;
      A.inst.has_surf2 = Ksurf - Kp > 0 ? 1 : 0; A.inst.surf2 = st_surf;
    } else {
      A.inst.has_surf = Ksurf > 0 ? 1 : 0; A.inst.surf = st_surf;
    }
    // W3
    A.prep.T = ev_t2m ? c.T : 0; A.prep.x = s.x.p; A.prep.tris = c.tris.p; A.prep.order = c.tri_order.p; A.prep.spheres = s.spheres.p;
    A.prep.has_t2m = ev_t2m ? 1 : 0; A.prep.t2m = st_t2m;
    A.prep.n_cnt = 0;
    auto reset_cnt = [&](const VertexTask& v) { A.prep.cnt[A.prep.n_cnt] = v.cnt; A.prep.cnt_n[A.prep.n_cnt++] = v.Kpad; };
    if (pt) reset_cnt(st_vert);
    if (nnv_main > 0) reset_cnt(st_nnv);
    if (nnv_hi > nnv_lo) reset_cnt(st_ennv);
    if (t2m_nnv) reset_cnt(st_tnn);
    A.prep.zero_d = c.d_res.p; A.prep.n_zero_d = 8;
    // ---- the search sequences: tasks are appended to a StepSearchArgs (surface tasks first)
    struct Seq { StepSearchArgs* q; int nt = 0, n_corr = 0; };
    auto seq_init = [](StepSearchArgs& q) { q.s_corr[0] = q.s_corr[1] = q.v_corr[0] = q.v_corr[1] = -1; q.fstart[0] = 0; q.rstart[0] = 0; return Seq{&q}; };
    auto add_surface = [&](Seq& sq, const SurfaceTask& t, const CorrTask* corr) {
      StepSearchArgs& q = *sq.q;
      q.s[q.n_surf] = t;
      q.fstart[sq.nt + 1] = q.fstart[sq.nt] + filter_grid_blocks(t.tblocks, t.ksplit);
      q.rstart[sq.nt + 1] = q.rstart[sq.nt] + t.K;
      if (corr) { q.corr[sq.n_corr] = *corr; q.s_corr[q.n_surf] = sq.n_corr++; }
      ++q.n_surf; ++sq.nt;
    };
    auto add_vertex = [&](Seq& sq, const VertexTask& t, const CorrTask* corr) {
      StepSearchArgs& q = *sq.q;
      q.v[q.n_vert] = t;
      q.fstart[sq.nt + 1] = q.fstart[sq.nt] + filter_grid_blocks(t.vblocks, t.ksplit);
      q.rstart[sq.nt + 1] = q.rstart[sq.nt] + t.K;
      if (corr) { q.corr[sq.n_corr] = *corr; q.v_corr[q.n_vert] = sq.n_corr++; }
      ++q.n_vert; ++sq.nt;
    };
    auto seq_close = [](Seq& sq) { for (int u = sq.nt + 1; u < 5; ++u) { sq.q->fstart[u] = sq.q->fstart[sq.nt]; sq.q->rstart[u] = sq.q->rstart[sq.nt]; } };
    CorrTask corr_m{}, corr_t{};
    if (pm) corr_m = CorrTask{pm->K, ep[im]->corr(), s.x.p, nullptr, c.target.boundary.p, nullptr, pm->prm.boundary_aware,
                              s.pose, c.ref.p, c.mean.p, c.tris.p, c.adj_off.p, c.adj.p, prop_nnv ? s.surf_cp.p : nullptr};
    if (pt) corr_t = CorrTask{pt->K, ep[itx]->corr(), s.x.p, pt->target_pts.p, c.boundary.p, nullptr, pt->prm.boundary_aware,
                              s.pose, c.ref.p, c.mean.p, c.tris.p, c.adj_off.p, c.adj.p, nullptr};
    // main, stage 1 (W4/W5) and stage 2 (W6/W7: nearest vertices of the surface points, ModelSampling correspondences with their flag)
    Seq m1 = seq_init(A.s1), m2 = seq_init(A.s2), e1 = seq_init(A.s1b), e2 = seq_init(A.s2b);
    const SurfaceTask& st_first = split ? st_surfp : st_surf;
    if (st_first.K > 0) add_surface(m1, st_first, (pm && !prop_nnv) ? &corr_m : nullptr);
    if (ev_t2m && !split) add_surface(m1, st_t2m, nullptr);
    if (pt) add_vertex(m1, st_vert, &corr_t);
    if (nnv_main > 0) add_vertex(m2, st_nnv, prop_nnv ? &corr_m : nullptr);
    if (t2m_nnv && !split) add_vertex(m2, st_tnn, nullptr);
    // the evaluator's own sequence (split): the model ids behind the proposal's, the target -> model direction, their nearest vertices
    if (split) {
      if (st_surf.K > 0) add_surface(e1, st_surf, nullptr);
      if (ev_t2m) add_surface(e1, st_t2m, nullptr);
      if (nnv_hi > nnv_lo) add_vertex(e2, st_ennv, nullptr);
      if (t2m_nnv) add_vertex(e2, st_tnn, nullptr);
    }
    seq_close(m1); seq_close(m2); seq_close(e1); seq_close(e2);
    // W8: regressions + the likelihood's reductions
    StepRegressionArgs& g = A.reg.reg;
    g.n = w.do_post ? n_props : 0; g.r = r; g.ntiles = regression_tiles(r); g.Q = c.Q.p;
    g.ustart[0] = 0; g.ustart[1] = 0; g.ustart[2] = 0;
    int splits[2] = {1, 1};
    double* parts[2] = {nullptr, nullptr};
    for (int i = 0; i < g.n; ++i) {
      icp_proposal* p = it.props[i];
      splits[i] = regression_splits(p->K);
      g.K[i] = p->K;
      g.kchunk[i] = std::max(1, (p->K + splits[i] - 1) / splits[i]);
      g.cb[i] = ep[i]->corr();
      g.wt[i] = 1.0 / (p->prm.tangential_noise * p->prm.tangential_noise);
      g.kappa[i] = 1.0 / (p->prm.noise_along_normal * p->prm.noise_along_normal) - g.wt[i];
      if (p->side_factor_pending || p->side_asm_pending) {  // (a per-stage step of this proposal left work on its side streams)
        HIP_OK(hipStreamSynchronize(c.front_stream)); sync_eigen(c);
        p->side_factor_pending = false; p->side_asm_pending = false;
      }
      p->side_parts = nullptr; p->side_parts_entry = nullptr;
      p->mpart_half = (p->mpart_half + 1) % icp_proposal::kMpartRing;
      parts[i] = g.Mpart[i] = p->mpart_for_write(p->mpart_half, S);
      g.status[i] = p->status.p + ep[i]->status_off;
      g.ustart[i + 1] = g.ustart[i] + g.ntiles * splits[i];
    }
    if (g.n == 1) g.ustart[2] = g.ustart[1];
    // the likelihood's reductions over ALL ids (the layout finish_eval reads) …
    WideRegArgs full{};
    full.eval_kind = evp.kind; full.eval_m2t = ev_m2t ? 1 : 0; full.eval_t2m = ev_t2m ? 1 : 0;
    full.Km = Km; full.d2m = s.surf_d2.p;
    full.flags_m = eval_nnv ? c.target.boundary.p : nullptr; full.idx_m = eval_nnv ? s.surf_nnv.p : nullptr;
    full.Kt = Kt; full.d2t = e->t2m_d2.p;
    full.flags_t = t2m_nnv ? c.target.boundary.p : nullptr; full.idx_t = t2m_nnv ? e->t2m_nnv.p : nullptr;  // (sic: SURVEY App. D5)
    full.n_flags = c.target.V;
    full.mean = evp.gauss_mean; full.sigma = evp.gauss_sigma;
    full.red_out = c.d_res.p;
    const StepRegressionArgs reg_only = g;
    if (!split) {                 // … behind every search of the one sequence
      A.reg = full; A.reg.reg = reg_only;
    } else if (concurrent) {      // … each sequence its own range (the Hausdorff maxima are order-independent: atomic maxima into one word)
      A.reg = full; A.reg.reg = reg_only;
      A.reg.eval_m2t = Kp > 0 ? 1 : 0; A.reg.Km = Kp; A.reg.eval_t2m = 0;
      A.regb = full;
      A.regb.eval_m2t = Km - Kp > 0 ? 1 : 0; A.regb.Km = Km - Kp; A.regb.d2m = s.surf_d2.p + Kp;
    } else {                      // … at the end of the evaluator's sequence, which runs behind the main one on the same stream
      A.reg = full; A.reg.reg = reg_only;
      A.reg.eval_m2t = 0; A.reg.eval_t2m = 0;
      A.regb = full;
    }
    plan.grid_prep = std::max(plan.grid_prep, wide_prep_grid(A.prep));
    plan.grid_f1 = std::max(plan.grid_f1, A.s1.fstart[m1.nt]);
    plan.grid_r1 = std::max(plan.grid_r1, A.s1.rstart[m1.nt]);
    plan.grid_f2 = std::max(plan.grid_f2, A.s2.fstart[m2.nt]);
    plan.grid_r2 = std::max(plan.grid_r2, A.s2.rstart[m2.nt]);
    plan.grid_reg = std::max(plan.grid_reg, wide_reg_blocks(A.reg));
    plan.grid_f1b = std::max(plan.grid_f1b, A.s1b.fstart[e1.nt]);
    plan.grid_r1b = std::max(plan.grid_r1b, A.s1b.rstart[e1.nt]);
    plan.grid_f2b = std::max(plan.grid_f2b, A.s2b.fstart[e2.nt]);
    plan.grid_r2b = std::max(plan.grid_r2b, A.s2b.rstart[e2.nt]);
    if (split) plan.grid_regb = std::max(plan.grid_regb, wide_reg_blocks(A.regb));

    // ---- W9..W12
    for (int i = 0; i < 16; ++i) c.h_res[i] = 0.0;
    for (int i = 0; i < 16; ++i) c.h_status[i] = 0;
    w.seq = ++c.step_seq;
    WideDoneItem di{};
    di.red_src = c.d_res.p; di.red_dst = c.h_res;
    di.host_flag = c.h_flag; di.seq = w.seq;
    for (int i = 0; i < g.n; ++i) {
      icp_proposal* p = it.props[i];
      if (splits[i] > 1) { sum_parts.push_back(parts[i]); sum_splits.push_back(splits[i]); }
      PosteriorFactorIO io{parts[i], 1, ep[i]->M.p, ep[i]->alpha.p, p->status.p + ep[i]->status_off, p->fscratch.p};
      if (root_here) { io.Lout = ep[i]->V.p; io.Sout = ep[i]->S.p; root_entries.push_back(ep[i]); root_props.push_back(p); }
      factors.push_back(io);
      di.st_src[i] = p->status.p + ep[i]->status_off; di.st_dst[i] = c.h_status + 8 + i;
      if (w.shape_only) {
        w.tails[2 * i] = TransitionTailIO{ec[i]->alpha.p, ec[i]->M.p, ec[i]->coeffs.p, ep[i]->coeffs.p, p->prm.step_length,
                                          c.h_res + 8 + 2 * i, c.h_status + 2 * i};
        w.tails[2 * i + 1] = TransitionTailIO{ep[i]->alpha.p, ep[i]->M.p, ep[i]->coeffs.p, ec[i]->coeffs.p, p->prm.step_length,
                                              c.h_res + 9 + 2 * i, c.h_status + 2 * i + 1};
        tails.push_back(w.tails[2 * i]); tails.push_back(w.tails[2 * i + 1]);
        w.n_tails = 2 * (i + 1);
      }
      if (w.do_spec && !root_here) {
        EigenRequest rq;
        p->prepare_eigen(*ep[i], &rq);
        rq.sqrt_lambda = c.sqrt_lambda.p;
        ep[i]->eig_checked = false;
        ep[i]->eig_event_valid = false; ep[i]->eig_done_shared = nullptr; ep[i]->eig_shared_gen = nullptr;
        p->mpart_reader[p->mpart_half] = ep[i];
        eigen_streams_for(c, Es[0], Es[1]);
        const int fl = two_eig ? (int)(p->eig_flip++ & 1) : 0;
        if (fl) rq.work = p->work2.p;
        spec_rq[fl].push_back(rq);
        spec_parts[fl].push_back(parts[i]);
        spec_entries[fl].push_back(ep[i]);
      }
    }
    dones.push_back(di);
    it.issued = true;
  }

  // ---- one sequence of launches for all of them
  Bound _b(&lead, true, true);
  for (int fl = 0; fl < 2; ++fl) {
    if (pre_rq[fl].empty()) continue;  // KL bases of current states that have none yet (a chain's first ICP proposal; speculation off)
    const hipStream_t E = Es[fl];
    if (eigen_tridiag_many_supported(r)) launch_posterior_eigen_tridiag_many(E, r, (int)pre_rq[fl].size(), pre_rq[fl].data(), nullptr);
    else
      for (auto& rq : pre_rq[fl])
        if (!launch_posterior_eigen_pair(E, r, rq.sqrt_lambda, 1, &rq)) fail(ICP_ERR_DEVICE, "internal: wide step at a rank without a decomposition kernel");
    BatchEventSlot* ev_pre = &next_batch_event(elead.device);
    HIP_OK(hipEventRecord(ev_pre->ev, E));
    HIP_OK(hipStreamWaitEvent(S, ev_pre->ev, 0));
    for (PosteriorEntry* en : pre_entries[fl]) {
      en->eig_done_shared = ev_pre->ev; en->eig_shared_gen = &ev_pre->gen; en->eig_shared_gen_value = ev_pre->gen;
      en->eig_event_valid = true; en->done_value = 0;
    }
  }
  for (size_t p0 = 0; p0 < prop_items.size(); p0 += kWideMaxChains) {
    WideProposeArgs pa{};
    pa.n = (int)std::min<size_t>(kWideMaxChains, prop_items.size() - p0);
    for (int i = 0; i < pa.n; ++i) pa.it[i] = prop_items[p0 + i];
    launch_wide_propose(S, r, pa);
  }
  launch_wide_head(S, plan, chain_args.data(), lead.wide_pinned[turn], lead.wide_device[turn].p);
  // the step's searches, regressions and reductions: on `S` — or, where the evaluator's searches are a sequence of their own, those
  // on `S` and the proposals' chain (searches of their K ids, regression, then factorisation and tails) on the second stream beside them
  const hipStream_t Sm = (any_split && concurrent) ? S2 : S;
  if (any_split && concurrent) {
    HIP_OK(hipEventRecord(lead.ev_wide_head[turn], S));
    HIP_OK(hipStreamWaitEvent(S2, lead.ev_wide_head[turn], 0));
  }
  launch_wide_main(Sm, plan, lead.wide_device[turn].p);
  for (size_t p0 = 0; p0 < sum_parts.size(); p0 += kWideMaxChains)
    launch_sum_partials_many(Sm, r, (int)std::min<size_t>(kWideMaxChains, sum_parts.size() - p0), sum_parts.data() + p0, sum_splits.data() + p0);
  HIP_OK(hipEventRecord(lead.ev_wide_sum[turn], Sm));
  // the one-workgroup kernels on the second stream: the evaluator's sequence and the next batch's chip-wide launches on `S` run beside them
  if (Sm != S2) HIP_OK(hipStreamWaitEvent(S2, lead.ev_wide_sum[turn], 0));
  const bool any_spec = !spec_rq[0].empty() || !spec_rq[1].empty();
  const bool jacobi_spec = any_spec && !eigen_tridiag_many_supported(r);  // (ranks <= 64: the iteration reads the finished M)
  // The decompositions and factorisations of this step — the critical path — are handed to the device BEFORE the evaluator's own
  // sequence (chip-wide launches on `S`, which they then run beside), the factorisations in one launch (two took 200 + 880 µs in a
  // 30-chain step: the second one started among the evaluator's searches).  configs[4], 30 chains a step: 11.2k -> 12.6k it/s with
  // the tridiagonalisation's load prologue (icp_tridiag.hpp: tridiag_kernel_body), tools/r4_trace_c4.sh.
  if (any_spec && !jacobi_spec)
    for (int fl = 0; fl < 2; ++fl) {
      if (spec_rq[fl].empty()) continue;
      // the proposed states' KL bases BESIDE their factorisations: M = I + the summed partials is written at the head of the
      // decomposition as well (the values the factorisation's own assembly writes)
      HIP_OK(hipStreamWaitEvent(Es[fl], lead.ev_wide_sum[turn], 0));
      launch_posterior_eigen_tridiag_many(Es[fl], r, (int)spec_rq[fl].size(), spec_rq[fl].data(), spec_parts[fl].data());
    }
  {
    const size_t fmax = (size_t)posterior_factor_max();
    for (size_t p0 = 0; p0 < factors.size(); p0 += fmax)
      launch_posterior_factor(S2, r, (int)std::min(fmax, factors.size() - p0), factors.data() + p0);
  }
  if (any_split) {
    launch_wide_eval(S, plan, lead.wide_device[turn].p);
    HIP_OK(hipEventRecord(lead.ev_wide_eval[turn], S));
  }
  if (!root_entries.empty()) {  // "decomposed" as soon as the factorisation is through: an event behind it stands for the basis
    BatchEventSlot& done = next_batch_event(elead.device);
    HIP_OK(hipEventRecord(done.ev, S2));
    for (size_t q = 0; q < root_entries.size(); ++q) {
      PosteriorEntry* en = root_entries[q];
      en->eig_done_shared = done.ev; en->eig_shared_gen = &done.gen; en->eig_shared_gen_value = done.gen;
      en->eig_event_valid = true; en->done_value = 0; en->eig_valid = true; en->eig_checked = false;
      root_props[q]->h_eig[en->status_off / 3] = 0;
    }
  }
  if (jacobi_spec) {
    HIP_OK(hipEventRecord(lead.ev_wide_fac[turn], S2));
    HIP_OK(hipStreamWaitEvent(Es[0], lead.ev_wide_fac[turn], 0));
    for (size_t q = 0; q < spec_rq[0].size(); ++q)
      if (!launch_posterior_eigen_pair(Es[0], r, spec_rq[0][q].sqrt_lambda, 1, &spec_rq[0][q])) fail(ICP_ERR_DEVICE, "internal: wide step at a rank without a decomposition kernel");
  }
  for (int fl = 0; fl < 2; ++fl) {
    if (spec_rq[fl].empty()) continue;
    BatchEventSlot& done = next_batch_event(elead.device);
    HIP_OK(hipEventRecord(done.ev, Es[fl]));
    for (PosteriorEntry* en : spec_entries[fl]) {
      en->eig_done_shared = done.ev; en->eig_shared_gen = &done.gen; en->eig_shared_gen_value = done.gen;
      en->eig_event_valid = true; en->done_value = 0;
    }
  }
  for (size_t t0 = 0; t0 < tails.size(); t0 += 2 * kWideMaxChains)
    launch_transition_tails(S2, r, (int)std::min<size_t>(2 * kWideMaxChains, tails.size() - t0), tails.data() + t0, elead.Ginv.p, kSigma2);
  if (any_split) HIP_OK(hipStreamWaitEvent(S2, lead.ev_wide_eval[turn], 0));  // (the reductions of the evaluator's own sequence)
  for (size_t p0 = 0; p0 < dones.size(); p0 += kWideMaxChains) {
    WideDoneArgs da{};
    da.n = (int)std::min<size_t>(kWideMaxChains, dones.size() - p0);
    for (int i = 0; i < da.n; ++i) da.it[i] = dones[p0 + i];
    launch_wide_done(S2, da);
  }
  t.wide_streams[0] = S; t.wide_streams[1] = S2;
}

// results of one wide item (its flag has been waited for) -> false: the step has to be done again
bool wide_record(icp_step_ticket& t, int b, double* log_value_prop, double* fwd, double* bwd, int* status) {
  BatchItem& it = t.items[b];
  WideItem& w = it.W;
  icp_evaluator* e = it.e;
  icp_ctx& c = *e->ctx;
  const int r = c.r, n_props = t.n_props, generator = it.generator;
  const size_t P = 10 + (size_t)r;
  const double* theta_cur = t.theta_cur[b];
  double* theta_prop = t.theta_prop[b];
  PosteriorEntry** ec = w.ec;
  PosteriorEntry** ep = w.ep;
  if (w.eigen_first_use) {  // this step drew from a basis whose status nobody has looked at yet (the decomposition left it in pinned memory)
    icp_proposal* p = it.props[generator];
    PosteriorEntry& g = *ec[generator];
    int st = p->h_eig[g.status_off / 3];
    {  // (test-hooks build only: the n-th such look pretends the decomposition reported 2 — tests/test_gpu_wide.py)
      static const int pretend_at = dev_env("ICP_TEST_WIDE_EIGEN_STATUS") ? std::atoi(dev_env("ICP_TEST_WIDE_EIGEN_STATUS")) : 0;
      static std::atomic<int> looks{0};
      if (pretend_at > 0 && st == 0 && ++looks == pretend_at) st = 2;
    }
    if (st != 0) {
      // the multisection could not separate the spectrum (or the iteration did not converge): the per-stage decomposition, which
      // has the Jacobi fall-back in its launch sequence, takes over, and the step is done again from the basis it leaves
      g.eig_valid = false; g.eig_checked = false; g.eig_event_valid = false;
      p->warm_valid = false;
      p->ensure_eigen(g);
      sync_eigen(c);
      st = p->h_eig[g.status_off / 3];
      if (st != 0) {
        p->h_status[g.status_off + 2] = st;
        p->check_status(g);  // throws
      }
      return false;
    }
    p->h_status[g.status_off + 2] = 0;
  }
  const double* h_coeffs = c.h_res + 16;
  if (generator >= 0) {
    std::memcpy(theta_prop, theta_cur, sizeof(double) * 10);  // NonRigidIcpProposal.scala:64-66: only the shape changes
    for (int j = 0; j < r; ++j) {
      if (!std::isfinite(h_coeffs[j])) fail(ICP_ERR_NOT_FINITE, "proposed coefficients are not finite");
      theta_prop[10 + j] = h_coeffs[j];
    }
  }
  StateSlot& s = *w.s;
  s.theta.assign(theta_prop, theta_prop + P);
  s.valid = true;
  s.stamp = ++c.clock;
  s.defo_valid = true;
  s.spheres_valid = w.spheres;
  s.n_surf = w.Ksurf;
  s.n_nnv = w.Knnv;
  s.lo_surf = s.hi_surf = s.lo_nnv = s.hi_nnv = 0;
  if (w.do_post)
    for (int i = 0; i < n_props; ++i) {
      icp_proposal* p = it.props[i];
      ep[i]->theta.assign(theta_prop, theta_prop + P);
      ep[i]->valid = true;
      ep[i]->stamp = ++p->clock;
      p->h_status[ep[i]->status_off] = c.h_status[8 + i];
      p->h_status[ep[i]->status_off + 1] = 0;
      p->h_status[ep[i]->status_off + 2] = 0;
      if (w.shape_only) p->check_status(*ec[i]);
      p->check_status(*ep[i]);
    }
  for (int tl = 0; tl < w.n_tails; ++tl)
    if (c.h_status[tl] != 0) {  // rare: the fixed-point tail did not contract -> direct kernel
      std::vector<double> saved(c.h_res, c.h_res + 16);
      icp_proposal* p = it.props[tl / 2];
      TransitionTailIO io = w.tails[tl];
      io.out = c.d_res.p;
      io.status = c.d_status.p + 32;
      sync_eigen(c);  // (the direct form borrows the eigen work buffer)
      launch_transition_tail_direct(c.stream, r, io, c.G.p, kSigma2, p->work.p);
      c.finish(1, 64);
      if (c.h_status[32] != 0) fail(ICP_ERR_NOT_SPD, "G + sigma^2 M is not positive definite");
      saved[8 + tl] = c.h_res[0];
      std::memcpy(c.h_res, saved.data(), sizeof(double) * saved.size());
    }
  icp_evaluator::Memo* m = eval_store(e, theta_prop);
  m->status = finish_eval(e, c.h_res, &m->value, m->aux);
  *log_value_prop = m->value;
  *status = m->status;
  for (int i = 0; i < n_props; ++i) {
    if (!w.shape_only) { fwd[i] = -INFINITY; bwd[i] = -INFINITY; continue; }
    fwd[i] = c.h_res[8 + 2 * i];
    bwd[i] = c.h_res[9 + 2 * i];
    if (std::isnan(fwd[i]) || std::isnan(bwd[i])) fail(ICP_ERR_NOT_FINITE, "NaN transition probability");
  }
  e->last_prop.assign(theta_prop, theta_prop + P);
  ++c.paths.n[1]; ++g_step_paths.n[1];
  return true;
}

}  // namespace


extern "C" {

int icp_chain_step_batched_issue(int32_t n_chains, icp_evaluator* const* evaluators, int32_t n_props, icp_proposal* const* props,
                                 const int32_t* generator, const double* const* theta_cur_in, const double* const* z_in,
                                 double* const* theta_prop_in, double* log_value_prop, double* fwd, double* bwd, int32_t* status,
                                 icp_ctx* launch_ctx, icp_step_ticket** out) {
  if (out) *out = nullptr;
  icp_step_ticket* tk = nullptr;
  int rc = guard([&] {
    require(out != nullptr, "null argument");
    require(n_chains >= 1 && evaluators && generator && theta_cur_in && theta_prop_in && log_value_prop && status, "null argument");
    tk = new icp_step_ticket();
    icp_step_ticket& t = *tk;
    t.n_chains = n_chains; t.n_props = n_props;
    t.theta_cur.assign(theta_cur_in, theta_cur_in + n_chains);
    t.theta_prop.assign(theta_prop_in, theta_prop_in + n_chains);
    t.z.assign(n_chains, nullptr);
    if (z_in) t.z.assign(z_in, z_in + n_chains);
    t.log_value_prop = log_value_prop; t.fwd = fwd; t.bwd = bwd; t.status = status;
    std::vector<BatchItem>& items = t.items;
    std::vector<StepCapture>& caps = t.caps;
    const double* const* theta_cur = t.theta_cur.data();
    const double* const* z = z_in ? t.z.data() : nullptr;
    double* const* theta_prop = t.theta_prop.data();
    typedef BatchItem Item;
    require(n_props >= 0 && n_props <= 8 && (n_props == 0 || (props && fwd && bwd)), "bad proposal list");
    t.props.assign(props, props + (size_t)n_chains * n_props);
    items.resize(n_chains);
    for (int b = 0; b < n_chains; ++b) {
      Item& it = items[b];
      it.e = evaluators[b];
      it.props = t.props.data() + (size_t)b * n_props;
      it.generator = generator[b];
      require(it.e && theta_cur[b] && theta_prop[b], "null argument");
      require(it.generator < n_props, "generator index out of range");
      require(it.generator < 0 || (z && z[b]), "z is null");
      icp_ctx& c = *it.e->ctx;
      for (int i = 0; i < n_props; ++i) require(it.props[i] && it.props[i]->ctx == &c, "proposal belongs to another context");
      check_theta_finite(&c, theta_cur[b]);
      if (it.generator < 0) check_theta_finite(&c, theta_prop[b]);
      else
        for (int j = 0; j < c.r; ++j)
          if (!std::isfinite(z[b][j])) fail(ICP_ERR_NOT_FINITE, "z contains a non-finite value");
      it.key = it.generator >= 0 ? z[b] : theta_prop[b];
      status[b] = ICP_OK;
    }
    // `lead` carries the launches (its stream, its argument buffers); the decompositions go to the first chain's eigen stream
    icp_ctx& elead = *items[0].e->ctx;
    icp_ctx& lead = launch_ctx ? *launch_ctx : elead;
    require(lead.device == elead.device, "launch context on another device");
    t.lead = &lead;
    // (the launch context's rings hold kBatchRing batches: argument slots, eigen records, gate words, events — a ticket more would
    // rewrite what a batch still on the device reads)
    if (lead.tickets_in_flight.fetch_add(1) >= icp_ctx::kBatchRing) {
      --lead.tickets_in_flight;
      fail(ICP_ERR_BUSY, "the launch context already carries ICP_MAX_BATCHES_IN_FLIGHT uncollected batches: collect or abandon one first");
    }
    t.counted = true;
    // ---- which chains share the launches
    int n_batched = 0, wide_first = -1;
    for (int b = 0; b < n_chains; ++b) {
      Item& it = items[b];
      icp_ctx& c = *it.e->ctx;
      bool ok = n_props >= 1 && n_props <= 2 && c.device == elead.device && c.r == elead.r;
      for (int a = 0; a < b && ok; ++a) ok = !(items[a].batched && items[a].e->ctx == &c);
      if (!ok) continue;
      it.lk = std::unique_lock<std::recursive_mutex>(c.mu);
      if (c.batch_busy) fail(ICP_ERR_BUSY, "a chain's context already belongs to a batch in flight");
      if (it.e->front.valid || c.front_stream_used) {  // half steps launched ahead by the pipelined entry points: drained
        if (it.e->front.valid) release_front(it.e->front);
        c.bind();
        HIP_OK(hipStreamSynchronize(c.stream));
        HIP_OK(hipStreamSynchronize(c.front_stream));
        c.front_stream_used = false;
      }
      if (!chain_step_covered(it.e, n_props, it.props, it.generator, theta_cur[b], theta_prop[b])) {
        // what the five merged launches do not cover takes the wide step (open targets, the Hausdorff evaluator, ranks up to 200, pose
        // moves), side by side with the other such chains of the batch that share the first one's model
        const bool same_model = wide_first < 0 || items[wide_first].e->ctx->Qp.p == c.Qp.p;
        if (!step_pipeline_covers(it.e, n_props, it.props) && same_model &&
            wide_chain_covered(it.e, n_props, it.props, it.generator, theta_cur[b], theta_prop[b])) {
          it.wide = true;
          if (wide_first < 0) wide_first = b;
          continue;  // (its lock stays held until the end of this call, as the batched chains')
        }
        it.lk.unlock();
        continue;
      }
      it.batched = true;
      ++n_batched;
    }
    // ---- per chain: host side of the step, launches captured
    g_batch_timing.start();
    caps.resize(n_batched > 0 ? n_batched : 1);
    // the decompositions of the chains that moved go out first, together, so that they run while the host prepares the
    // launches (a chain whose posteriors are not on record yet starts its own in enqueue_front)
    EigenCollect eigens{[&] { std::lock_guard<std::recursive_mutex> lead_lk(lead.mu); return batch_eigen_stream(lead, &elead); }(), {}, {}};
    for (int b = 0; b < n_chains; ++b) {
      Item& it = items[b];
      if (!it.batched) continue;
      icp_ctx& c = *it.e->ctx;
      PosteriorEntry* ec[2] = {nullptr, nullptr};
      bool all = true;
      for (int i = 0; i < n_props; ++i) { ec[i] = it.props[i]->find_entry(theta_cur[b]); all = all && ec[i]; }
      if (!all) continue;
      Bound _b(&c, true);
      for (int i = 0; i < n_props; ++i) it.props[i]->resolve_speculation(theta_cur[b]);
      // (ranks > 64 decompose through the library, each chain on its own eigen stream, and are awaited on the host below)
      start_decompositions(c, n_props, it.props, ec, c.stream_used_elsewhere, eigen_speculation_supported(c.r) ? &eigens : nullptr);
    }
    StepBatchGate gate{};
    if (!eigens.rq.empty()) {  // … in ONE launch, on the batch's eigen stream
      std::lock_guard<std::recursive_mutex> lead_lk(lead.mu);  // (the record ring and the gate counter are the launch context's)
      Bound _b(&elead, true);
      const int turn = (lead.batch_eig_turn = (lead.batch_eig_turn + 1) % icp_ctx::kBatchRing);
      const size_t bytes = eigen_many_record_bytes((int)eigens.rq.size());
      if (bytes > lead.batch_eig_rec_bytes[turn]) {
        // (the slot's previous reader was the batch kBatchRing tickets ago: collected — tickets_in_flight —, its launches finished)
        if (lead.batch_eig_rec[turn]) { pinned_free(lead.batch_eig_rec[turn]); lead.batch_eig_rec[turn] = nullptr; }
        const size_t cap_bytes = std::max(bytes, eigen_many_record_bytes(128));
        pinned_alloc((void**)&lead.batch_eig_rec[turn], cap_bytes);
        lead.batch_eig_rec_bytes[turn] = cap_bytes;
      }
      if (!lead.batch_gate.p) {
        lead.batch_gate.alloc(16);
        lead.batch_gate.fill_bytes(0);  // (… and waits for the fill: the decompositions launched below count into it)
        pinned_alloc((void**)&lead.h_gate_error, sizeof(int) * 16);
        lead.h_gate_error[0] = 0;
        for (int k = 0; k < icp_ctx::kBatchRing; ++k) lead.batch_gate_expected[k] = 0;
      }
      // (the slot's own counter word: up to kBatchRing batches are in flight per launch context, and the workgroups of a LATER batch's
      // decompositions counting into one shared word could open an earlier batch's gate before its own decompositions are resident)
      const int wgs = launch_posterior_eigen_many(eigens.stream, elead.r, (int)eigens.rq.size(), eigens.rq.data(), lead.batch_eig_rec[turn],
                                                  lead.batch_gate.p + turn);
      require(wgs > 0, "internal: batched decompositions at a rank the kernel does not cover");
      HIP_OK(hipGetLastError());  // (a launch that failed would leave the gate below waiting for arrivals that never come)
      lead.batch_gate_expected[turn] = (int)((unsigned)lead.batch_gate_expected[turn] + (unsigned)wgs);  // (wraps with the counter)
      gate = StepBatchGate{lead.batch_gate.p + turn, lead.batch_gate_expected[turn], lead.h_gate_error};
      // test hook (tools/r3_timeout_repro.py: round 2's schedule, for the record): the launch sequence is not held back
      static const bool no_gate = dev_env("ICP_TEST_NO_GATE") != nullptr;
      if (no_gate) gate = StepBatchGate{};
      g_batch_timing.mark(4);
      BatchEventSlot& done = next_batch_event(elead.device);
      HIP_OK(hipEventRecord(done.ev, eigens.stream));
      for (PosteriorEntry* en : eigens.all) {  // (eig_event_valid is set where the requests were collected)
        en->eig_done_shared = done.ev;
        en->eig_shared_gen = &done.gen;
        en->eig_shared_gen_value = done.gen;
      }
      g_batch_timing.mark(5);
    }
    int nb = 0;
    for (int b = 0; b < n_chains; ++b) {
      Item& it = items[b];
      if (!it.batched) continue;
      icp_ctx& c = *it.e->ctx;
      const int r = c.r;
      Bound _b(&c, true);
      for (int i = 0; i < n_props; ++i) it.props[i]->resolve_speculation(theta_cur[b]);
      bool other_work = c.stream_used_elsewhere;  // another entry point may still be busy on `stream` …
      for (int i = 0; i < n_props; ++i) other_work = other_work || !it.props[i]->find_entry(theta_cur[b]);  // … or is about to be
      StepCapture& cap = caps[nb];
      std::memset(cap.grid, 0, sizeof(cap.grid));
      struct CaptureScope { CaptureScope(StepCapture* c) { step_capture(c); } ~CaptureScope() { step_capture(nullptr); } } scope(&cap);
      enqueue_front(it.e, n_props, it.props, it.generator, theta_cur[b], it.key, it.F, true);
      it.issued = true;
      StepFront& F = it.F;
      // the batch runs on the first chain's stream: what it needs from this chain's own streams is awaited here
      if (other_work) HIP_OK(hipStreamSynchronize(c.stream));
      if (it.generator >= 0 && F.ec[it.generator]->done_value == 0) sync_eigen(c);
      for (int i = 0; i < 16; ++i) c.h_res[i] = 0.0;
      for (int i = 0; i < 16; ++i) c.h_status[i] = 0;
      const int step_seq = ++c.step_seq;
      StepFinishArgs& f = it.f;
      f.n = n_props; f.r = r; f.Ginv = c.Ginv.p; f.sigma2 = kSigma2;
      for (int i = 0; i < n_props; ++i) {
        icp_proposal* p = it.props[i];
        f.Mpart[i] = F.mpart[i]; f.splits[i] = F.splits[i];
        f.M[i] = F.ep[i]->M.p; f.alpha[i] = F.ep[i]->alpha.p;
        f.status[i] = p->status.p + F.ep[i]->status_off;
        f.host_status[i] = c.h_status + 8 + i;
        f.fwd[i] = TransitionTailIO{F.ec[i]->alpha.p, F.ec[i]->M.p, F.ec[i]->coeffs.p, F.ep[i]->coeffs.p, p->prm.step_length,
                                    c.h_res + 8 + 2 * i, c.h_status + 2 * i};
        f.bwd[i] = TransitionTailIO{F.ep[i]->alpha.p, F.ep[i]->M.p, F.ep[i]->coeffs.p, F.ec[i]->coeffs.p, p->prm.step_length,
                                    c.h_res + 9 + 2 * i, c.h_status + 2 * i + 1};
      }
      f.done_counter = c.d_done.p; f.host_flag = c.h_flag; f.seq = step_seq;
      f.ready_flag = c.d_done.p + 2;
      c.last_back_seq = step_seq;
      launch_step_finish(c.stream, f);  // (captured)
      it.f = cap.finish;                // as finalised by the launcher
      ++nb;
    }
    // ---- one sequence of launches for all of them
    g_batch_timing.mark(0);
    if (nb > 0) {
      // (the launch context may itself be busy — a member of an earlier batch on the same stream — but its stream and its
      // argument ring are used under its lock)
      std::lock_guard<std::recursive_mutex> lead_lk(lead.mu);
      Bound _b(&lead, true, true);
      const size_t bytes = step_batch_bytes(nb);
      const int turn = (lead.batch_turn = (lead.batch_turn + 1) % icp_ctx::kBatchRing);
      if (bytes > lead.batch_bytes[turn]) {
        HIP_OK(hipStreamSynchronize(lead.stream));
        if (lead.batch_pinned[turn]) { pinned_free(lead.batch_pinned[turn]); lead.batch_pinned[turn] = nullptr; }
        const size_t cap_bytes = std::max(bytes, step_batch_bytes(16));
        pinned_alloc((void**)&lead.batch_pinned[turn], cap_bytes);
        lead.batch_device[turn].alloc(cap_bytes);
        lead.batch_bytes[turn] = cap_bytes;
      }
      static const bool finish_aside = dev_env("ICP_BATCH_FINISH_INLINE") == nullptr;  // (A/B switch)
      launch_step_batch(lead.stream, nb, caps.data(), lead.batch_pinned[turn], lead.batch_device[turn].p,
                        finish_aside ? lead.front_stream : nullptr, lead.ev_join, gate);
      if (finish_aside) t.finish_stream = lead.front_stream;
    }
    t.nb = nb;
    t.lead = &lead;
    if (wide_first >= 0) wide_issue(t, lead, *items[wide_first].e->ctx);
    g_batch_timing.mark(1);
    // no mutex is held across the API boundary: the member contexts are marked busy instead (other entry points fail with
    // ICP_ERR_BUSY until the ticket is collected or abandoned — by any thread)
    for (auto& it : items)
      if (it.lk.owns_lock()) { it.e->ctx->batch_busy = true; it.lk.unlock(); }
  });
  if (rc != ICP_OK) {
    if (tk) { batch_release(*tk); delete tk; }
    return rc;
  }
  *out = tk;
  return ICP_OK;
}

int icp_chain_step_batched_collect(icp_step_ticket* tk) {
  if (!tk) return ICP_ERR_INVALID_ARG;
  icp_step_ticket& t = *tk;
  const int n_chains = t.n_chains, n_props = t.n_props, nb = t.nb;
  std::vector<BatchItem>& items = t.items;
  const double* const* theta_cur = t.theta_cur.data();
  const double* const* z = t.z.data();
  double* const* theta_prop = t.theta_prop.data();
  double* log_value_prop = t.log_value_prop;
  double* fwd = t.fwd;
  double* bwd = t.bwd;
  int32_t* status = t.status;
  typedef BatchItem Item;
  int rc = guard([&] {
    icp_ctx& lead = *t.lead;
    g_batch_timing.start();
    if (lead.h_gate_error && lead.h_gate_error[0]) {  // (counted; the launches went ahead and their own time-outs take it from there)
      lead.h_gate_error[0] = 0;
      ++lead.stats.gate_timeouts; ++g_runtime_stats.gate_timeouts;
    }
    // ---- results, chain by chain
    bool first_wait = true;
    for (int b = 0; b < n_chains; ++b) {
      Item& it = items[b];
      if (!it.batched) continue;
      icp_ctx& c = *it.e->ctx;
      it.lk = std::unique_lock<std::recursive_mutex>(c.mu);
      Bound _b(&c, true, true);
      volatile int* flag = c.h_flag;
      const auto t_start = std::chrono::steady_clock::now();
      long spins = 0;
      while (*flag != it.f.seq) {
        if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(2)) break;
      }
      if (*flag != it.f.seq) {
        HIP_OK(hipStreamSynchronize(lead.stream));
        if (t.finish_stream) HIP_OK(hipStreamSynchronize(t.finish_stream));
      }
      if (first_wait) { g_batch_timing.mark(2); first_wait = false; }
      if (it.F.eigen_first_use && !eigen_speculation_supported(c.r)) {  // (its status is not written to pinned memory)
        sync_proposal_status_if(it.props[it.generator], true);
        c.finish(0, 0);
      }
      c.stage_used = 0;
      if (c.h_wait_error[0]) {  // the decomposition this chain draws from did not finish in time (a tool that serialises kernels)
        HIP_OK(hipStreamSynchronize(lead.stream));
        if (t.finish_stream) HIP_OK(hipStreamSynchronize(t.finish_stream));
        sync_eigen(c);
        c.h_wait_error[0] = 0;
        ++c.stats.wait_timeouts; ++g_runtime_stats.wait_timeouts;
        it.redo = true;
      } else {
        int st = ICP_OK;
        it.redo = !chain_step_record(it.e, n_props, it.props, it.generator, theta_cur[b], it.F, it.f, theta_prop[b], log_value_prop + b,
                                     fwd + (size_t)b * n_props, bwd + (size_t)b * n_props, &st);
        status[b] = st;
      }
      if (it.redo) { ++c.stats.step_redos; ++g_runtime_stats.step_redos; }
      if (it.redo) release_front(it.F);
      else {
        ++c.paths.n[0]; ++g_step_paths.n[0];
        it.F.s->reserved = false;
        for (int i = 0; i < n_props; ++i) it.F.ep[i]->reserved = false;
      }
      it.issued = false;
      c.batch_busy = false;
      it.lk.unlock();
    }
    // ---- the chains that took the wide step
    for (int b = 0; b < n_chains; ++b) {
      Item& it = items[b];
      if (!it.wide) continue;
      icp_ctx& c = *it.e->ctx;
      it.lk = std::unique_lock<std::recursive_mutex>(c.mu);
      Bound _b(&c, true, true);
      volatile int* flag = c.h_flag;
      const auto t_start = std::chrono::steady_clock::now();
      long spins = 0;
      while (*flag != it.W.seq) {
        if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(5)) break;
      }
      if (*flag != it.W.seq) {
        for (hipStream_t ws : t.wide_streams)
          if (ws) HIP_OK(hipStreamSynchronize(ws));
        if (*flag != it.W.seq) fail(ICP_ERR_DEVICE, "internal: a wide step's completion flag did not arrive");
      }
      c.stage_used = 0;
      int st = ICP_OK;
      it.redo = !wide_record(t, b, log_value_prop + b, fwd + (size_t)b * n_props, bwd + (size_t)b * n_props, &st);
      status[b] = st;
      if (it.redo) { ++c.stats.step_redos; ++g_runtime_stats.step_redos; }
      wide_release(it, !it.redo);
      it.issued = false;
      c.batch_busy = false;
      it.lk.unlock();
    }
    g_batch_timing.mark(3);
    if (g_batch_timing.on) { ++g_batch_timing.calls; g_batch_timing.chains += nb; }
  });
  batch_release(t);
  if (rc != ICP_OK) { delete tk; return rc; }
  // ---- the others, and whatever has to be done again, one after the other
  int first_bad = ICP_OK;
  for (int b = 0; b < n_chains; ++b) {
    Item& it = items[b];
    if ((it.batched || it.wide) && !it.redo) continue;
    if (g_batch_timing.on) ++g_batch_timing.stepped_alone;
    const int st = icp_chain_step(it.e, n_props, it.props, it.generator, theta_cur[b], it.generator >= 0 ? z[b] : nullptr, theta_prop[b],
                                  log_value_prop + b, fwd + (size_t)b * n_props, bwd + (size_t)b * n_props);
    status[b] = st;
  }
  for (int b = 0; b < n_chains; ++b)
    if (status[b] != ICP_OK && status[b] != ICP_ERR_EMPTY && first_bad == ICP_OK) first_bad = status[b];
  delete tk;
  return first_bad;
}

int icp_chain_step_batched_abandon(icp_step_ticket* tk) {
  if (!tk) return ICP_ERR_INVALID_ARG;
  batch_release(*tk);  // waits for the batch's launches, gives back what they hold; nothing of the step is recorded
  delete tk;
  return ICP_OK;
}

// --------------------------------------------------------------------- the whole MH loop on the device (MhChain, kernels_step.hip)
int icp_chains_run_on_device(int32_t n_chains, icp_evaluator* const* evaluators, int32_t n_props, icp_proposal* const* props_in,
                             const icp_mh_mixture* mix, const uint64_t* seeds, const int64_t* first_step, double* const* theta,
                             double* log_value, int32_t n_steps, double* const* records, int64_t* accepted) {
  struct Chain {
    icp_evaluator* e = nullptr;
    icp_proposal* props[2] = {nullptr, nullptr};
    PosteriorEntry* set[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [sel][proposal]
    StateSlot* slot = nullptr;
    std::unique_lock<std::recursive_mutex> lk;
    bool busy = false;
  };
  std::vector<Chain> chains;
  auto release = [&]() {
    for (auto& ch : chains) {
      if (!ch.e) continue;
      icp_ctx& c = *ch.e->ctx;
      if (!ch.lk.owns_lock()) ch.lk = std::unique_lock<std::recursive_mutex>(c.mu);
      for (int sel = 0; sel < 2; ++sel)
        for (int i = 0; i < 2; ++i)
          if (ch.set[sel][i]) ch.set[sel][i]->reserved = false;
      if (ch.slot) ch.slot->reserved = false;
      if (ch.busy) c.batch_busy = false;
      ch.lk.unlock();
    }
  };
  int rc = guard([&] {
    require(n_chains >= 1 && evaluators && props_in && mix && seeds && first_step && theta && log_value && n_steps >= 0, "null argument");
    require(n_props >= 1 && n_props <= 2, "the on-device loop takes one or two ICP proposals per chain");
    require(mix->struct_size == sizeof(icp_mh_mixture), "icp_mh_mixture::struct_size does not match this library's header");
    require(mix->w_icp > 0.0 && mix->w_rw >= 0.0 && mix->rw_sigma > 0.0 && mix->w_pose >= 0.0, "bad mixture");
    if (mix->w_pose > 0.0)
      for (int a = 0; a < 3; ++a) require(mix->pose_rot_sigma[a] > 0.0 && mix->pose_trans_sigma[a] > 0.0, "pose walk sigmas must be positive");
    chains.resize(n_chains);
    icp_ctx& lead = *evaluators[0]->ctx;
    const int r = lead.r, P = 10 + r;
    require(eigen_speculation_supported(r), "the on-device loop covers ranks 3..64");
    // ---- claim the chains' contexts, fix their posterior entries and state slot
    for (int b = 0; b < n_chains; ++b) {
      Chain& ch = chains[b];
      require(evaluators[b] && theta[b], "null argument");
      icp_ctx& c = *evaluators[b]->ctx;
      require(c.device == lead.device && c.r == r, "chains of one run share a device and a rank");
      for (int a = 0; a < b; ++a) require(chains[a].e->ctx != &c, "every chain needs a context of its own");
      ch.e = evaluators[b];
      for (int i = 0; i < n_props; ++i) {
        ch.props[i] = props_in[(size_t)b * n_props + i];
        require(ch.props[i] && ch.props[i]->ctx == &c, "proposal belongs to another context");
        require(ch.props[i]->sampler == ch.props[0]->sampler && ch.props[i]->sampler == chains[0].props[0]->sampler, "one sampler per run");
      }
      check_theta_finite(&c, theta[b]);
      ch.lk = std::unique_lock<std::recursive_mutex>(c.mu);
      if (c.batch_busy) fail(ICP_ERR_BUSY, "a chain's context already belongs to a batch in flight");
      require(step_pipeline_covers(ch.e, n_props, ch.props), "configuration not covered by the merged launches");
      // (the device makes the proposed pose's matrix with the library's own convention: open to a host whose supplied matrices have
      // all agreed with it — icp_ctx_set_rotation's check —, closed to one whose convention is another)
      if (mix->w_pose > 0.0)
        require(c.rotations_mismatched == 0, "pose walks on the device: a caller-supplied rotation matrix disagreed with the library's Rz·Ry·Rx (icp_ctx_rotation_convention)");
      Bound _b(&c);
      if (ch.e->front.valid) release_front(ch.e->front);
      for (int i = 0; i < n_props; ++i) {
        icp_proposal* p = ch.props[i];
        p->resolve_speculation(theta[b]);
        PosteriorEntry& cur = p->posterior(theta[b], false);  // the current state's posterior and its basis, the ordinary way
        p->ensure_eigen(cur);
        cur.reserved = true;
        ch.set[0][i] = &cur;
        PosteriorEntry& other = p->fresh_entry();
        other.reserved = true;
        ch.set[1][i] = &other;
      }
      HIP_OK(hipStreamSynchronize(c.stream));
      HIP_OK(hipStreamSynchronize(c.front_stream));
      sync_eigen(c);
      for (int i = 0; i < n_props; ++i) sync_proposal_status(ch.props[i]);
      HIP_OK(hipStreamSynchronize(c.stream));
      for (int i = 0; i < n_props; ++i) {
        ch.props[i]->check_status(*ch.set[0][i]);
        if (ch.props[i]->h_eig[ch.set[0][i]->status_off / 3] != 0) fail(ICP_ERR_NOT_FINITE, "posterior eigen-decomposition did not converge");
      }
      StateSlot& s = c.fresh_state();
      s.reserved = true;
      s.pose = c.pose_of(theta[b]);
      ch.slot = &s;
      c.batch_busy = true;
      ch.busy = true;
      ch.lk.unlock();
    }
    // ---- groups: each its own stream, everything of a group in order on it; the groups overlap each other's launches
    // (three since the token moved forward, §5.1c: 64 chains 202k it/s in two groups, 211k in three; 128 chains 251k / 257k; 32 chains 130k /
    // 135k; 24 chains 104k / 108k.  Four — a fourth stream made with every context — measured 218k / 266k / 137k / 109k, but the extra
    // stream shifts every context's streams over the runtime's hardware queues, and the host-stepped lockstep path's decompositions, which
    // wait on the device for launches of other streams, then ran into their time-outs: 15 of them in a 50-chain test.  Not adopted.)
    const int n_groups = n_chains >= 16 ? 3 : 1;
    struct Group {
      int b0 = 0, B = 0;
      hipStream_t st = nullptr;
      int grid[5] = {0, 0, 0, 0, 0};
      DBuf<StepBeginArgs> begin_alt, begin_live;
      DBuf<StepSearchArgs> search_alt, search_live;
      bool filter_prepared = true;  // (step_filter_prepared of every captured step)
      DBuf<StepRegressionArgs> regression_alt, regression_live;
      DBuf<StepFinishArgs> finish_alt, finish_live;
      DBuf<MhChain> mh;
      DBuf<EigenProblem> eig_live;
      DBuf<int> eig_skip;
      DBuf<double> normals[2], theta, rec;
      DBuf<double> res;   // per chain 32 doubles: [0..7] launch 4's reductions, [8..11] the tails fwd_i / bwd_i — in DEVICE memory
      DBuf<int> stat;     // per chain 16 ints: [0..3] the tails' status, [8..9] the factorisations' (the decide kernel reads them: pinned
                          // host memory, where the host-stepped paths want them, would cost it a bus round trip per number)
      double* h_normals[2] = {nullptr, nullptr};
      hipEvent_t ev_copy[2] = {nullptr, nullptr};
      hipEvent_t ev_big = nullptr;  // recorded behind launch 4: the next group's chip-wide launches may start
    };
    std::vector<Group> groups(n_groups);
    constexpr int kChunk = 64;  // steps per block of standard normals
    const int root = chains[0].props[0]->sampler == ICP_SAMPLER_CHOLESKY_ROOT;
    struct GroupGuard {
      std::vector<Group>& g;
      ~GroupGuard() {
        for (auto& gr : g) {
          for (int k = 0; k < 2; ++k) {
            if (gr.h_normals[k]) pinned_free(gr.h_normals[k]);
            if (gr.ev_copy[k]) (void)hipEventDestroy(gr.ev_copy[k]);
          }
          if (gr.ev_big) (void)hipEventDestroy(gr.ev_big);
        }
      }
    } group_guard{groups};
    lead.bind();
    std::vector<double> zero_key(P, 0.0);
    for (int g = 0; g < n_groups; ++g) {
      Group& gr = groups[g];
      gr.b0 = (int)((long long)g * n_chains / n_groups);
      gr.B = (int)((long long)(g + 1) * n_chains / n_groups) - gr.b0;
      // the first chain's three streams: created one after the other with the context, they sit on different hardware queues and run
      // beside each other (streams of DIFFERENT contexts, or streams made later, may share a queue: the runtime multiplexes streams
      // onto a handful of them, and two groups on one queue alternate in ~55 µs slices — every kernel of the step then "takes" a
      // multiple of that: eight streams made for the purpose on first use ran two groups at 116k it/s instead of 202k)
      gr.st = g == 0 ? lead.stream : g == 1 ? lead.front_stream : lead.eig_stream;
      const int B = gr.B;
      gr.begin_alt.alloc(2 * B); gr.begin_live.alloc(B);
      gr.search_alt.alloc(2 * B); gr.search_live.alloc(B);
      gr.regression_alt.alloc(2 * B); gr.regression_live.alloc(B);
      gr.finish_alt.alloc(2 * B); gr.finish_live.alloc(B);
      gr.mh.alloc(B);
      gr.eig_live.alloc((size_t)B * n_props);
      gr.eig_skip.alloc((size_t)B * n_props);
      gr.theta.alloc((size_t)B * P);
      gr.res.alloc((size_t)B * 32); gr.res.fill_bytes(0);
      gr.stat.alloc((size_t)B * 16); gr.stat.fill_bytes(0);
      HIP_OK(hipEventCreateWithFlags(&gr.ev_big, hipEventDisableTiming));
      gr.rec.alloc(records ? (size_t)B * std::max(n_steps, 1) * (4 + P) : 1);
      for (int k = 0; k < 2; ++k) {
        gr.normals[k].alloc((size_t)B * kChunk * r);
        pinned_alloc((void**)&gr.h_normals[k], sizeof(double) * (size_t)B * kChunk * r);
        HIP_OK(hipEventCreateWithFlags(&gr.ev_copy[k], hipEventDisableTiming));
      }
      std::vector<StepBeginArgs> hb(2 * B);
      std::vector<StepSearchArgs> hs(2 * B);
      std::vector<StepRegressionArgs> hr(2 * B);
      std::vector<StepFinishArgs> hf(2 * B);
      std::vector<MhChain> hm(B);
      std::vector<int> hskip((size_t)B * n_props, 1);
      std::vector<double> hth((size_t)B * P);
      for (int k = 0; k < B; ++k) {
        Chain& ch = chains[gr.b0 + k];
        icp_ctx& c = *ch.e->ctx;
        std::lock_guard<std::recursive_mutex> lk(c.mu);
        Bound _b(&c, true, true);
        MhChain& m = hm[k];
        std::memset(&m, 0, sizeof(m));
        for (int sel = 0; sel < 2; ++sel) {
          // the step's launches with the current state in set `sel` and the proposed one in the other, captured
          StepCapture cap;
          std::memset(cap.grid, 0, sizeof(cap.grid));
          StepFront F;
          F.n_props = n_props; F.generator = -1; F.parity = 0; F.stream = c.stream; F.s = ch.slot;
          for (int i = 0; i < n_props; ++i) { F.props[i] = ch.props[i]; F.ec[i] = ch.set[sel][i]; F.ep[i] = ch.set[1 - sel][i]; }
          {
            struct CaptureScope { CaptureScope(StepCapture* cp) { step_capture(cp); } ~CaptureScope() { step_capture(nullptr); } } scope(&cap);
            front_launches(ch.e, n_props, ch.props, -1, zero_key.data(), F, true, false);
            StepFinishArgs f{};
            f.n = n_props; f.r = r; f.Ginv = c.Ginv.p; f.sigma2 = kSigma2;
            for (int i = 0; i < n_props; ++i) {
              icp_proposal* p = ch.props[i];
              f.Mpart[i] = F.mpart[i]; f.splits[i] = F.splits[i];
              f.M[i] = F.ep[i]->M.p; f.alpha[i] = F.ep[i]->alpha.p;
              f.status[i] = p->status.p + F.ep[i]->status_off;
              f.host_status[i] = gr.stat.p + (size_t)k * 16 + 8 + i;
              f.fwd[i] = TransitionTailIO{F.ec[i]->alpha.p, F.ec[i]->M.p, F.ec[i]->coeffs.p, F.ep[i]->coeffs.p, p->prm.step_length,
                                          gr.res.p + (size_t)k * 32 + 8 + 2 * i, gr.stat.p + (size_t)k * 16 + 2 * i};
              f.bwd[i] = TransitionTailIO{F.ep[i]->alpha.p, F.ep[i]->M.p, F.ep[i]->coeffs.p, F.ec[i]->coeffs.p, p->prm.step_length,
                                          gr.res.p + (size_t)k * 32 + 9 + 2 * i, gr.stat.p + (size_t)k * 16 + 2 * i + 1};
            }
            f.done_counter = c.d_done.p; f.host_flag = c.h_flag; f.seq = 0;
            f.ready_flag = nullptr;
            launch_step_finish(c.stream, f);  // (captured; finalised by the launcher)
          }
          cap.begin.wait_flag = nullptr; cap.begin.wait2_flag = nullptr; cap.begin.wait_ticks = nullptr; cap.begin.hold_regs = 0;
          cap.regression.red_out = gr.res.p + (size_t)k * 32;  // (device memory instead of the context's pinned area)
          {  // (launch 1's matvec layout: set by enqueue_front only when it knows the generator)
            int t = 0;
            while (t < 6 && (r << (t + 1)) <= 256 && (r >> (t + 1)) >= 8) ++t;
            cap.begin.tpr_log2 = t;
          }
          hb[(size_t)sel * B + k] = cap.begin; hs[(size_t)sel * B + k] = cap.search; hr[(size_t)sel * B + k] = cap.regression;
          hf[(size_t)sel * B + k] = cap.finish;
          for (int q = 0; q < 5; ++q) gr.grid[q] = std::max(gr.grid[q], cap.grid[q]);
          gr.filter_prepared = gr.filter_prepared && step_filter_prepared(cap.search);
          m.begin_alt[sel] = gr.begin_alt.p + (size_t)sel * B + k;
          m.search_alt[sel] = gr.search_alt.p + (size_t)sel * B + k;
          m.regression_alt[sel] = gr.regression_alt.p + (size_t)sel * B + k;
          m.finish_alt[sel] = gr.finish_alt.p + (size_t)sel * B + k;
          for (int i = 0; i < n_props; ++i) {
            icp_proposal* p = ch.props[i];
            PosteriorEntry& cur = *ch.set[sel][i];
            m.prop_alt[sel][i] = ProposeIn{cur.alpha.p, cur.V.p, cur.S.p, c.inv_sqrt_lambda.p, c.P.p, cur.coeffs.p, nullptr, kSigma2,
                                           p->prm.step_length, root};
            // the decomposition of set `sel`'s posterior (an accepted state arrives there), warm-started from the other set's basis
            PosteriorEntry& other = *ch.set[1 - sel][i];
            EigenRequest rq{cur.M.p, root ? nullptr : other.V.p, cur.V.p, cur.Vt.p, cur.S.p, p->work.p, p->status.p + cur.status_off + 2, nullptr,
                            p->h_eig + cur.status_off / 3, p->eig_words.p + cur.status_off / 3, 0, c.sqrt_lambda.p};
            rq.root = root != 0;
            m.eig_alt[sel][i] = eigen_problem_of(r, rq);
          }
        }
        m.begin_live = gr.begin_live.p + k; m.search_live = gr.search_live.p + k;
        m.regression_live = gr.regression_live.p + k; m.finish_live = gr.finish_live.p + k;
        m.eig_live = gr.eig_live.p + (size_t)k * n_props;
        m.eig_skip = gr.eig_skip.p + (size_t)k * n_props;
        m.pw_id_mask = 2046;
        m.seed = seeds[gr.b0 + k];
        m.r = r; m.n_icp = n_props;
        {  // MixtureProposal weights, normalised as the harness normalises them (host/icp_host.hpp: pick_component)
          double ws = 0.0;
          for (int i = 0; i < n_props; ++i) ws += mix->icp_weight[i];
          for (int i = 0; i < n_props; ++i) m.icp_w[i] = mix->icp_weight[i] / ws;
          double raw[3];
          int no = 0;
          if (mix->w_pose > 0.0) { m.outer_kind[no] = 0; raw[no++] = mix->w_pose; }  // (BfmFittingPartial.scala:70: pose, ICP, shape walk)
          m.outer_kind[no] = 1; raw[no++] = mix->w_icp;
          if (mix->w_rw > 0.0) { m.outer_kind[no] = 2; raw[no++] = mix->w_rw; }
          double wsum = 0.0;
          for (int o = 0; o < no; ++o) wsum += raw[o];
          for (int o = 0; o < no; ++o) m.outer_w[o] = raw[o] / wsum;
          m.n_outer = no;
        }
        if (mix->w_pose > 0.0) {  // MixedProposalDistributions.scala:29-39 / host/icp_host.cpp: mixed_random_pose_proposal
          static const int param_index[6] = {6, 5, 4, 1, 2, 3};  // yaw = rotation._3, pitch = _2, roll = _1 (PoseProposals.scala:39-41); x, y, z
          m.n_pose = 6;
          double wsum = 0.0;
          for (int a = 0; a < 6; ++a) wsum += 0.5;
          for (int a = 0; a < 6; ++a) {
            const double sd = a < 3 ? mix->pose_rot_sigma[a] : mix->pose_trans_sigma[a - 3];
            m.pose_index[a] = param_index[a];
            m.pose_w[a] = 0.5 / wsum;
            m.pose_sigma[a] = sd;
            m.pose_logc[a] = std::log(std::sqrt(2.0 * M_PI)) + std::log(sd);  // breeze Gaussian(0, σ).logPdf's normaliser
          }
          m.front_every_step = 1;
        }
        m.rw_sigma = mix->rw_sigma;
        m.rw_logc = 0.5 * (r * std::log(2.0 * M_PI) + r * std::log(mix->rw_sigma * mix->rw_sigma));
        m.prior_c = 0.5 * r * std::log(2.0 * M_PI);
        const icp_evaluator_params& ep = ch.e->prm;
        m.eval_kind = ep.kind == ICP_EVAL_INDEPENDENT_POINT_DISTANCE ? 0 : 2;
        m.eval_mode = ep.mode;
        m.gauss_mean = ep.gauss_mean; m.gauss_sigma = ep.gauss_sigma;
        m.gauss_logn = std::log(std::sqrt(2.0 * M_PI)) + std::log(ep.gauss_sigma);
        m.exp_rate = ep.exp_rate; m.exp_lograte = std::log(ep.exp_rate);
        m.coeff_prop = ch.slot->coeffs.p;
        m.red = gr.res.p + (size_t)k * 32;
        m.tails = gr.res.p + (size_t)k * 32 + 8;
        m.tail_status = gr.stat.p + (size_t)k * 16;
        m.chol_status = gr.stat.p + (size_t)k * 16 + 8;
        m.normals = nullptr; m.normals_first = 0; m.normals_rows = 0;
        m.records = records && records[gr.b0 + k] ? gr.rec.p + (size_t)k * n_steps * (4 + P) : nullptr;
        m.rec_first = first_step[gr.b0 + k];
        m.theta = gr.theta.p + (size_t)k * P;
        m.cur_p = log_value[gr.b0 + k];
        m.step = first_step[gr.b0 + k];
        m.accepted = 0; m.cur_sel = 0; m.gen = -1; m.leaf = -1; m.error = 0;
        for (int i = 0; i < n_props; ++i) m.eig_seq[i] = ch.props[i]->eig_seq;
        std::memcpy(hth.data() + (size_t)k * P, theta[gr.b0 + k], sizeof(double) * P);
        for (int i = 0; i < 16; ++i) { c.h_res[i] = 0.0; c.h_status[i] = 0; }
      }
      HIP_OK(hipMemcpy(gr.begin_alt.p, hb.data(), sizeof(StepBeginArgs) * hb.size(), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.search_alt.p, hs.data(), sizeof(StepSearchArgs) * hs.size(), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.regression_alt.p, hr.data(), sizeof(StepRegressionArgs) * hr.size(), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.finish_alt.p, hf.data(), sizeof(StepFinishArgs) * hf.size(), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.eig_skip.p, hskip.data(), sizeof(int) * hskip.size(), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.theta.p, hth.data(), sizeof(double) * hth.size(), hipMemcpyHostToDevice));
      // (normals: the two buffers are addressed through MhChain::normals / normals_first, re-pointed per block of steps below)
      HIP_OK(hipMemcpy(gr.mh.p, hm.data(), sizeof(MhChain) * hm.size(), hipMemcpyHostToDevice));
      HIP_OK(hipStreamSynchronize(nullptr));  // (as DBuf::upload: the copies have reached the device before a non-blocking stream's launch reads them)
    }
    // ---- the loop: per block of kChunk steps the chains' standard normals (the harness' own expression, drawn here on the host
    // while the device works on the block before), then per step and group eight launches, nothing waited for
    auto draw_block = [&](Group& gr, int blk, int buf) {
      const int s0 = blk * kChunk, ns = std::min(kChunk, n_steps - s0);
      double* out = gr.h_normals[buf];
      for (int k = 0; k < gr.B; ++k) {
        const uint64_t seed = seeds[gr.b0 + k];
        const uint64_t f0 = (uint64_t)first_step[gr.b0 + k] + (uint64_t)s0;
        for (int s_ = 0; s_ < ns; ++s_)
          for (int j = 0; j < r; ++j) out[((size_t)k * kChunk + s_) * r + j] = harness_normal(seed, f0 + (uint64_t)s_, (uint64_t)j);
      }
    };
    const int n_blocks = (n_steps + kChunk - 1) / kChunk;
    // (per-kernel event timing, if the first chain's context is being profiled: icp_ctx_profile_start — its event pool, both streams)
    struct ProfBind { ProfBind(icp_ctx& c) { g_prof = c.profiling ? &c.prof : nullptr; } ~ProfBind() { g_prof = nullptr; } } prof_bind(lead);
    for (int blk = 0; blk < n_blocks; ++blk) {
      const int buf = blk & 1, s0 = blk * kChunk, ns = std::min(kChunk, n_steps - s0);
      for (auto& gr : groups) {
        HIP_OK(hipEventSynchronize(gr.ev_copy[buf]));  // (the staging buffer's previous upload has left it)
        draw_block(gr, blk, buf);
        HIP_OK(hipMemcpyAsync(gr.normals[buf].p, gr.h_normals[buf], sizeof(double) * (size_t)gr.B * kChunk * r, hipMemcpyHostToDevice, gr.st));
        HIP_OK(hipEventRecord(gr.ev_copy[buf], gr.st));
        launch_mh_set_normals(gr.st, gr.B, gr.mh.p, gr.normals[buf].p, kChunk * r, s0, ns);
      }
      // The chip-wide launches of two groups side by side slow each other down more than the overlap gains; what
      // should run beside a group's chip-wide launches is the OTHER group's small ones (launch 5 on four CUs per chain, the
      // decide kernel, the decompositions on three CUs each).  So the launches pass a token from group to group: a group's first
      // launch waits for an event of the previous group's.  Which one: behind launch 4 while the filter launch held five workgroups per
      // CU; with eight (§5.1c) behind launch 2 — the other group's begin and filter beside this group's resolve and regression, two
      // chains of latencies that leave the CUs room — measures best (two groups of 32 chains: 202.0k it/s against 197.3k behind launch
      // 4, 199.0k without a token; four groups of 16, on a build with a fourth stream: 216.2k against 215.0k behind launch 1, 206.3k
      // without, 199.6k behind launch 3, 167.9k behind launch 4); behind launch 1 with 64 per group (two groups of 64: 249.5k against 244.7k / 241.5k).
      for (int s_ = 0; s_ < ns; ++s_)
        for (int g = 0; g < n_groups; ++g) {
          Group& gr = groups[g];
          if (n_groups > 1) {
            Group& prev = groups[(g + n_groups - 1) % n_groups];
            if (blk > 0 || s_ > 0 || g > 0) HIP_OK(hipStreamWaitEvent(gr.st, prev.ev_big, 0));
          }
          // (later steps of a block: prepared by the decide kernel of the step before — except with pose walks, whose proposed pose is
          // made by the front kernel)
          if (s_ == 0 || mix->w_pose > 0.0) launch_mh_front(gr.st, gr.B, gr.mh.p);
          const int token_at = gr.B >= 64 ? 1 : 2;
          {
            int ga[5] = {0, 0, 0, 0, 0}, gb[5] = {0, 0, 0, 0, 0};
            for (int q = 0; q < 4; ++q) (q < token_at ? ga : gb)[q] = gr.grid[q];
            if (token_at > 0)
              launch_step_batch_resident(gr.st, gr.B, ga, r, gr.begin_live.p, gr.search_live.p, gr.regression_live.p, gr.finish_live.p, gr.filter_prepared);
            if (n_groups > 1) HIP_OK(hipEventRecord(gr.ev_big, gr.st));
            if (token_at < 4)
              launch_step_batch_resident(gr.st, gr.B, gb, r, gr.begin_live.p, gr.search_live.p, gr.regression_live.p, gr.finish_live.p, gr.filter_prepared);
          }
          int g5[5] = {0, 0, 0, 0, gr.grid[4]};
          launch_step_batch_resident(gr.st, gr.B, g5, r, gr.begin_live.p, gr.search_live.p, gr.regression_live.p, gr.finish_live.p);
          launch_mh_decide(gr.st, gr.B, gr.mh.p);
          launch_posterior_eigen_resident(gr.st, r, gr.B * n_props, gr.eig_live.p, gr.eig_skip.p, root);
        }
    }
    // ---- results: nothing is handed out unless every chain came through
    std::vector<std::vector<MhChain>> hms(n_groups);
    int first_error = 0;
    for (int g = 0; g < n_groups; ++g) {
      Group& gr = groups[g];
      HIP_OK(hipStreamSynchronize(gr.st));
      hms[g].resize(gr.B);
      HIP_OK(hipMemcpy(hms[g].data(), gr.mh.p, sizeof(MhChain) * hms[g].size(), hipMemcpyDeviceToHost));
      for (const MhChain& m : hms[g])
        if (m.error != 0 && first_error == 0) first_error = m.error;
    }
    if (first_error != 0) {
      for (auto& ch : chains) {  // whatever the sets hold now belongs to no state on record
        std::lock_guard<std::recursive_mutex> lk(ch.e->ctx->mu);
        for (int i = 0; i < n_props; ++i) {
          for (int sel = 0; sel < 2; ++sel) { ch.set[sel][i]->valid = false; ch.set[sel][i]->eig_valid = false; ch.set[sel][i]->eig_checked = false; }
          ch.props[i]->warm_valid = false;
          ch.props[i]->spec_entry = nullptr;
        }
        ch.slot->valid = false;
      }
      fail(first_error == 3 ? ICP_ERR_NOT_SPD : first_error == 5 ? ICP_ERR_EMPTY : ICP_ERR_NOT_FINITE,
           first_error == 2 ? "on-device loop: a transition tail did not contract (step these chains through icp_chain_step_batched)"
           : first_error == 6 ? "on-device loop: posterior eigen-decomposition did not converge"
                              : "on-device loop: a chain stopped on a non-finite, empty or non-positive-definite result");
    }
    // the decompositions behind the LAST step's decisions have no decide kernel behind them: their status (pinned, written by the
    // decomposition itself) is looked at here, before the sets are booked as decomposed and checked
    for (int g = 0; g < n_groups; ++g)
      for (int k = 0; k < groups[g].B; ++k) {
        Chain& ch = chains[groups[g].b0 + k];
        const MhChain& m = hms[g][k];
        for (int i = 0; i < n_props; ++i) {
          const int st = ch.props[i]->h_eig[ch.set[m.cur_sel][i]->status_off / 3];
          if (st != 0) {
            std::lock_guard<std::recursive_mutex> lk(ch.e->ctx->mu);
            for (int sel = 0; sel < 2; ++sel) { ch.set[sel][i]->valid = false; ch.set[sel][i]->eig_valid = false; ch.set[sel][i]->eig_checked = false; }
            ch.props[i]->warm_valid = false;
            ch.props[i]->spec_entry = nullptr;
            fail(ICP_ERR_NOT_FINITE, "on-device loop: posterior eigen-decomposition did not converge");
          }
        }
      }
    for (int g = 0; g < n_groups; ++g) {
      Group& gr = groups[g];
      std::vector<double> hth((size_t)gr.B * P);
      HIP_OK(hipMemcpy(hth.data(), gr.theta.p, sizeof(double) * hth.size(), hipMemcpyDeviceToHost));
      for (int k = 0; k < gr.B; ++k) {
        const int b = gr.b0 + k;
        Chain& ch = chains[b];
        icp_ctx& c = *ch.e->ctx;
        const MhChain& m = hms[g][k];
        std::memcpy(theta[b], hth.data() + (size_t)k * P, sizeof(double) * P);
        log_value[b] = m.cur_p;
        if (accepted) accepted[b] = m.accepted;
        if (records && records[b] && n_steps > 0)
          HIP_OK(hipMemcpy(records[b], gr.rec.p + (size_t)k * n_steps * (4 + P), sizeof(double) * (size_t)n_steps * (4 + P), hipMemcpyDeviceToHost));
        // the contexts' own bookkeeping: the set that holds the current state is on record again, decomposed
        std::lock_guard<std::recursive_mutex> lk(c.mu);
        for (int i = 0; i < n_props; ++i) {
          icp_proposal* p = ch.props[i];
          for (int sel = 0; sel < 2; ++sel) {
            PosteriorEntry& en = *ch.set[sel][i];
            const bool cur = sel == m.cur_sel;
            en.valid = cur; en.eig_valid = cur; en.eig_checked = cur; en.eig_event_valid = false; en.done_value = 0;
            if (cur) {
              en.theta.assign(theta[b], theta[b] + P);
              en.stamp = ++p->clock;
              p->h_eig[en.status_off / 3] = 0;
              p->warm_ptr = en.V.p;
              p->warm_valid = true;
            }
          }
          p->eig_seq = m.eig_seq[i];
          p->spec_entry = nullptr;
        }
        ch.slot->valid = false;
        ch.e->last_prop.clear();
        c.paths.n[3] += n_steps; g_step_paths.n[3] += n_steps;
      }
    }
  });
  release();
  return rc;
}

int icp_proposal_basis_state(icp_proposal* p, const double* theta) {
  if (!p || !theta) return ICP_ERR_INVALID_ARG;
  icp_ctx& c = *p->ctx;
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  if (c.batch_busy) return 1;  // (its context is part of a batch in flight: whatever it is doing, it is not ready)
  PosteriorEntry* e = p->find_entry(theta);
  if (!e) return 0;
  if (!e->eig_valid) return 0;
  return *(volatile int*)(p->h_eig + e->status_off / 3) == -1 ? 1 : 2;
}

int icp_chain_step_path(icp_evaluator* e, int32_t n_props, icp_proposal* const* props) {
  if (!e || n_props < 0 || (n_props > 0 && !props)) return ICP_ERR_INVALID_ARG;
  for (int i = 0; i < n_props; ++i)
    if (!props[i] || props[i]->ctx != e->ctx) return ICP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(e->ctx->mu);
  if (step_pipeline_covers(e, n_props, props)) return 0;
  if (n_props >= 1 && n_props <= 2 && wide_pipeline_covers(e, n_props, props)) return 1;
  return 2;
}

int icp_chain_step_batched(int32_t n_chains, icp_evaluator* const* evaluators, int32_t n_props, icp_proposal* const* props,
                           const int32_t* generator, const double* const* theta_cur, const double* const* z,
                           double* const* theta_prop, double* log_value_prop, double* fwd, double* bwd, int32_t* status) {
  icp_step_ticket* tk = nullptr;
  const int rc = icp_chain_step_batched_issue(n_chains, evaluators, n_props, props, generator, theta_cur, z, theta_prop, log_value_prop, fwd,
                                              bwd, status, nullptr, &tk);
  if (rc != ICP_OK) return rc;
  return icp_chain_step_batched_collect(tk);
}



}  // extern "C"
