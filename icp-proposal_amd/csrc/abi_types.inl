// abi_types.inl — part of icp_abi.hip (one translation unit; included there, in order).
// common types: errors, device buffers, shared model / target data, state slots, icp_ctx
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
// Scope of an object's CREATION: the fills and uploads of its new buffers (DBuf::fill_bytes / upload: null-stream work, which is not
// ordered against this library's non-blocking streams) are waited for ONCE, when the scope ends — before the creating call returns
// and anybody can launch on the buffers — instead of once per buffer (a proposal makes five, an evaluator four: 40 µs each).
struct NullStreamBatch {
  static thread_local int depth;
  NullStreamBatch() { ++depth; }
  ~NullStreamBatch() { if (--depth == 0) (void)hipStreamSynchronize(nullptr); }
};
thread_local int NullStreamBatch::depth = 0;

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
      if (NullStreamBatch::depth == 0) HIP_OK(hipStreamSynchronize(nullptr));  // (as fill_bytes: the copy has reached the device before any launch can read it)
    }
  }
  // (hipMemset may return before the device has filled device memory, and what it enqueues on the null stream is not ordered against
  // this library's non-blocking streams: a launch issued right behind it could see — or, worse, count into — the buffer before the
  // fill lands.  Seen once the device was busy with another thread's batches: a batch gate's arrival counter zeroed AFTER the first
  // arrivals, every later gate of that slot two seconds late.  The null stream is waited for here.)
  void fill_bytes(int v) {
    HIP_OK(hipMemset(p, v, sizeof(T) * (n ? n : 1)));
    if (NullStreamBatch::depth == 0) HIP_OK(hipStreamSynchronize(nullptr));
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
// Search hints to start from (round 5).  A search's hint — the previous winner of each query — is only an upper bound: any valid
// element index gives exact results, a good one gives them after a handful of tests.  A context's FIRST searches have none: every
// query lists thousands of candidates, overflows its list and scans the whole mesh (0.4-3 ms per search at the face model's size; a
// batch registration paid it for each of its 100 chains when the chain's initial state is evaluated: 2 ms per chain object).  The first
// evaluation that completes against a (model, target) pair therefore files its hints with the target's shared data, and every later
// context / evaluator of the same pair starts from a copy: another chain's shape is a few millimetres from this one's, its winners are
// tight bounds.  Immutable once filed.
struct HintSeed {
  uint64_t model_uid = 0;
  DBuf<int> surf, nnv;      // [N]: icp_ctx::hint_surf / hint_nnv after a first evaluation
  struct Eval { int Kt; uint64_t points_hash; DBuf<int> tri, nnv; };
  std::vector<std::unique_ptr<Eval>> evals;  // per evaluator point set: icp_evaluator::hint_tri / hint_nnv
};
struct SharedModel {
  uint64_t uid = 0;  // (unique per process: the key of a target's HintSeed)
  DBuf<double> ref, mean, Q, Qp, sqrt_lambda, inv_sqrt_lambda, G, Ginv, P;
  DBuf<int> tris, adj_off, adj, tri_order;
  DBuf<uint8_t> boundary;
  int n_boundary = 0;
  int device = 0;
  // host: ring_prefix_max[k] = the largest vertex id among the vertices k' <= k and the corners of their triangles — what the vertex
  // normals of the model ids 0..k read (the instance launch's head: wide_issue)
  std::vector<int> ring_prefix_max;
};
struct SharedTarget {
  DeviceMesh mesh;
  std::vector<std::unique_ptr<HintSeed>> seeds;  // one per model that has searched this target (guarded by g_shared_mu)
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

// One of a context's side streams (the merged step's second stream, the eigen streams), made on first use where the context asks for
// that: hipStreamCreateWithPriority takes 3-4 ms, a context has three of them beside its own stream, and a context that is only ever
// stepped as a MEMBER of batches (icp_chain_step_batched / the on-device loop: the launch context's streams carry everything) never
// enqueues on them — 57 of the 80 stream creations of a 10 targets x 10 chains job, a fifth of its wall time with chains of 50 steps.
// The first two contexts alive in a process make theirs at creation, next to the context stream (the runtime maps streams to its
// hardware queues in creation order: a latecomer shared a queue with the context stream, NOTES round 2).  No implicit conversion:
// get() where something is enqueued, peek() where a stream is compared, sync() where one is drained (a stream never made has nothing).
struct LazyStream {
  hipStream_t s = nullptr;
  int device = -1, priority = 0;
  bool greatest = false, eigen = false;
  void arm(int dev, bool great, int prio, bool is_eigen, bool now);
  hipStream_t get();
  hipStream_t peek() const { return s; }
  bool armed() const { return device >= 0; }
  explicit operator bool() const { return armed(); }  // "the context has such a stream"
  void sync();                                        // HIP_OK(hipStreamSynchronize) if it exists
  void sync_quiet() { if (s) (void)hipStreamSynchronize(s); }
  hipStream_t release() { hipStream_t r = s; s = nullptr; device = -1; return r; }
};

struct icp_ctx {
  int device = 0;
  RuntimeStats stats;
  StepPaths paths;
  hipStream_t stream = nullptr;
  // icp_chain_step alternates between two streams: the five launches of a step go to one of them in order, the next step's
  // to the other.  Launches 1-3 of a step do not depend on the finish launch of the step before it and run beside it; what
  // they must not overtake is that step's searches (same scratch, same hints), so launch 1 waits on the device for the word
  // the finish launch of that step raises when it starts (StepBeginArgs::wait_flag).  No event crosses the two streams.
  LazyStream front_stream;                       // the second of the two (`stream` is the first, and everybody else's)
  // every eigen-decomposition of the context runs on this stream, beside the chain (launch order = execution order, so the
  // decompositions of one proposal never overlap each other; the two directions of a step share ONE launch)
  LazyStream eig_stream;
  LazyStream eig_stream2;  // ranks above 64: decompositions started ahead alternate between the two (each with a work buffer of its own)
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
  bool hints_filed = false;  // this context's hints have been offered to the target's HintSeed (or came from it)
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
      front_stream.sync();
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
      front_stream.sync();
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
      if (c->front_stream_used) { c->front_stream.sync_quiet(); c->front_stream_used = false; }
      c->stream_used_elsewhere = true;
    }
  }
  ~Bound() { g_prof = nullptr; }
};
}  // namespace
