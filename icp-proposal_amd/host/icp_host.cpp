// icp_host.cpp — SamplingRegistration.runfitting mirrored over the C ABI (see icp_host.hpp / icp_host.h).
#include "icp_host.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "icp_host.hpp"

// java.lang.Double.toString for the magnitudes proposal names are made of (api/sampling/MixedProposalDistributions.scala:31-54
// interpolates Scala Doubles: "RandomShape-0.1", "RotationYaw-0.01"): the shortest decimal that round-trips, with a
// fractional part; computerised scientific notation below 1e-3 and from 1e7 ("1.0E-4")
static std::string scala_double(double x) {
  if (x != x) return "NaN";
  if (x == 0.0) return std::signbit(x) ? "-0.0" : "0.0";
  if (std::isinf(x)) return x > 0 ? "Infinity" : "-Infinity";
  char buf[64];
  int prec = 1;
  for (; prec <= 17; ++prec) {  // shortest mantissa that reproduces x
    std::snprintf(buf, sizeof(buf), "%.*e", prec - 1, x);
    if (std::strtod(buf, nullptr) == x) break;
  }
  std::string m(buf);
  const size_t epos = m.find('e');
  const int ex = std::atoi(m.c_str() + epos + 1);
  m = m.substr(0, epos);
  if (m.find('.') == std::string::npos) m += ".0";
  const double ax = std::fabs(x);
  if (ax >= 1e-3 && ax < 1e7) {
    std::string digits;
    bool neg = false;
    for (char c : m) { if (c == '-') neg = true; else if (c != '.') digits += c; }
    std::string out;
    if (ex >= 0) {
      while ((int)digits.size() < ex + 2) digits += '0';
      out = digits.substr(0, ex + 1) + "." + digits.substr(ex + 1);
    } else {
      out = "0." + std::string(-ex - 1, '0') + digits;
    }
    while (out.size() > 1 && out.back() == '0' && out[out.size() - 2] != '.') out.pop_back();
    return (neg ? "-" : "") + out;
  }
  return m + "E" + std::to_string(ex);
}

using namespace icphost;

namespace {
thread_local std::string g_host_err;

struct RecordLogger : AcceptRejectLogger {
  double* out = nullptr;
  int64_t index = 0, n_accept = 0;
  int P = 0;
  void write(int status, const ModelFittingParameters& state, int leaf, double logp) {
    if (out) {
      out[0] = (double)index;
      out[1] = status;
      out[2] = leaf;
      out[3] = logp;
      std::memcpy(out + ICP_HOST_RECORD_HEADER, state.data(), sizeof(double) * P);
      out += ICP_HOST_RECORD_HEADER + P;
    }
    ++index;
  }
  void accept(const ModelFittingParameters&, const ModelFittingParameters& sample, int leaf, double logp) override {
    ++n_accept;
    write(1, sample, leaf, logp);
  }
  void reject(const ModelFittingParameters& current, const ModelFittingParameters&, int leaf, double logp) override {
    write(0, current, leaf, logp);
  }
};
}  // namespace

// MixedProposalDistributions.mixedRandomPoseProposal (api/sampling/MixedProposalDistributions.scala:29-39): six walks of weight 0.5
// in the reference's order.  rot_sigma = (rotYaw, rotPitch, rotRoll), trans_sigma = (transX, transY, transZ) — the argument
// order of :29.  YawAxis perturbs rotation._3 = theta[6], PitchAxis _2 = theta[5], RollAxis _1 = theta[4]
// (api/sampling/proposals/PoseProposals.scala:39-41).  Leaf ids 3..8 in this order.
template <class Own>
static MixtureProposal* mixed_random_pose_proposal(Own&& own, const double rot_sigma[3], const double trans_sigma[3]) {
  MixtureProposal* poseMix = static_cast<MixtureProposal*>(own(new MixtureProposal()));
  static const char* names[6] = {"RotationYaw", "RotationPitch", "RotationRoll", "TranslationX", "TranslationY", "TranslationZ"};
  static const int param_index[6] = {6, 5, 4, 1, 2, 3};
  for (int a = 0; a < 6; ++a) {
    const double sd = a < 3 ? rot_sigma[a] : trans_sigma[a - 3];
    auto* p = new GaussianAxisPoseProposal(param_index[a], sd, std::string(names[a]) + "-" + scala_double(sd));
    p->leafId = 3 + a;
    own(p);
    poseMix->add(0.5, p);
  }
  return poseMix;
}

struct icp_host_chain {
  icp_ctx* ctx = nullptr;
  int r = 0;
  uint64_t seed = 0;
  icp_host_chain_config cfg{};  // (as given; the pointers inside are the caller's and are not used after creation)
  std::vector<std::unique_ptr<ProposalGeneratorWithTransition>> owned;
  std::vector<NonRigidIcpProposal*> icp;
  MixtureProposal* root = nullptr;
  std::unique_ptr<ModelPriorEvaluator> prior;
  std::unique_ptr<NativeLikelihoodEvaluator> likelihood;
  ProductEvaluator product;
  std::unique_ptr<MetropolisHastings> mh;
  ChainPrefetcher prefetcher;
  ModelFittingParameters current;
  RecordLogger logger;
  double current_p = 0.0;
  // the standard normals of step `ahead_step`, drawn by the idle hook of the native library while the step before it is
  // on the device (counter-based RNG: a pure function of (seed, step, lane), so drawing early changes nothing)
  std::vector<double> ahead, ahead2;  // … and of the step after that
  uint64_t ahead_step = ~0ull, ahead2_step = ~0ull;
  std::thread::id runner;  // the thread inside icp_host_chain_run (chains that share a context share its hook slot)
  void draw_normals(uint64_t step, std::vector<double>& out) const {
    StepRandom rnd{seed, step};
    for (size_t j = 0; j < out.size(); ++j) out[j] = rnd.normal(j);
  }
  static void draw_ahead(void* self) {
    icp_host_chain* ch = static_cast<icp_host_chain*>(self);
    if (std::this_thread::get_id() != ch->runner) return;  // another chain's step on a shared context: not our turn
    const uint64_t next = (uint64_t)ch->logger.index + 1;
    // the normals of step `next` were drawn one step earlier (below), so that the pre-launch can go out at once
    if (ch->ahead_step != next) {
      if (ch->ahead2_step == next) { ch->ahead.swap(ch->ahead2); ch->ahead2_step = ~0ull; }
      else ch->draw_normals(next, ch->ahead);
      ch->ahead_step = next;
      // Two steps out of three are rejected: the next step then starts from the SAME state, and its proposal is a pure
      // function of that state and of the numbers just drawn.  Its first launches are issued now, behind the step in
      // flight, so the device does not idle through the host's turn-around; if this step is accepted instead they are
      // dropped (icp_chain_step_prelaunch).
      if (ch->prefetcher.whole_step && !ch->icp.empty() && ch->icp.size() <= 2) {
        StepRandom rnd{ch->seed, next};
        rnd.ahead = ch->ahead.data(); rnd.n_ahead = (int)ch->ahead.size();
        ProposalGeneratorWithTransition* leaf = ch->root->peek(rnd, 0);
        icp_proposal* hs[2] = {nullptr, nullptr};
        for (size_t i = 0; i < ch->icp.size(); ++i) hs[i] = ch->icp[i]->h;
        if (auto* ip = dynamic_cast<NonRigidIcpProposal*>(leaf)) {
          if (ip->stepper) (void)icp_chain_step_prelaunch(ch->likelihood->h, (int)ch->icp.size(), hs, ip->stepperIndex, ch->current.data(), ch->ahead.data());
        } else if (auto* rw = dynamic_cast<RandomShapeUpdateProposal*>(leaf)) {
          const ModelFittingParameters prop = rw->propose(ch->current, rnd, 0);
          (void)icp_chain_step_prelaunch(ch->likelihood->h, (int)ch->icp.size(), hs, -1, ch->current.data(), prop.data());
        }
      }
    }
    if (ch->ahead2_step != next + 1) { ch->draw_normals(next + 1, ch->ahead2); ch->ahead2_step = next + 1; }
  }
};

template <class F>
static int host_guard(F&& f) {
  try {
    f();
    return ICP_OK;
  } catch (const NativeError& e) {
    g_host_err = e.what();
    return e.status;
  } catch (const std::exception& e) {
    g_host_err = e.what();
    return ICP_ERR_NOT_FINITE;
  }
}

extern "C" {

const char* icp_host_last_error(void) { return g_host_err.c_str(); }

int icp_host_chain_create(icp_ctx* ctx, const icp_host_chain_config* cfg, const double* theta0, uint64_t seed,
                          icp_host_chain** out) {
  if (out) *out = nullptr;
  icp_host_chain* ch = nullptr;
  int rc = host_guard([&] {
    if (!ctx || !cfg || !theta0 || !out || cfg->n_icp < 0 || cfg->n_icp > 2) throw NativeError(ICP_ERR_INVALID_ARG, "icp_host_chain_create");
    static const bool timing = std::getenv("ICP_CREATE_TIMING") != nullptr;  // (developer aid: where a chain object's creation goes)
    const auto t_begin = std::chrono::steady_clock::now();
    auto t_prop = t_begin, t_eval = t_begin;
    ch = new icp_host_chain();
    ch->ctx = ctx;
    ch->r = icp_ctx_rank(ctx);
    ch->seed = seed;
    ch->cfg = *cfg;
    auto own = [&](ProposalGeneratorWithTransition* p) { ch->owned.emplace_back(p); return p; };
    // MixedProposalDistributions.mixedProposalICP (MixedProposalDistributions.scala:48-68)
    MixtureProposal* icpMix = nullptr;
    if (cfg->n_icp > 0 && cfg->w_icp > 0) {
      icpMix = static_cast<MixtureProposal*>(own(new MixtureProposal()));
      for (int i = 0; i < cfg->n_icp; ++i) {
        const char* dir = cfg->icp[i].direction == ICP_TARGET_SAMPLING ? "TargetSampling" : "ModelSampling";
        auto* p = new NonRigidIcpProposal(ctx, cfg->icp[i], std::string("IcpProposal-") + dir + "-" + scala_double(cfg->icp[i].step_length) + "Step");
        p->leafId = i;
        own(p);
        if (cfg->sampler != 0) check(icp_proposal_set_sampler(p->h, cfg->sampler), "icp_proposal_set_sampler");
        ch->icp.push_back(p);
        icpMix->add(cfg->icp_weight[i], p);
      }
    }
    // MixedProposalDistributions.mixedRandomShapeProposal (:41-46): a one-component mixture
    MixtureProposal* rwMix = nullptr;
    if (cfg->w_rw > 0) {
      rwMix = static_cast<MixtureProposal*>(own(new MixtureProposal()));
      auto* p = new RandomShapeUpdateProposal(cfg->rw_sigma, "RandomShape-" + scala_double(cfg->rw_sigma));
      p->leafId = 2;
      own(p);
      rwMix->add(0.5, p);
    }
    // MixedProposalDistributions.mixedRandomPoseProposal (:29-39): six equally weighted 1-D walks
    MixtureProposal* poseMix = nullptr;
    if (cfg->w_pose > 0) poseMix = mixed_random_pose_proposal(own, cfg->pose_rot_sigma, cfg->pose_trans_sigma);
    // outer mixture (IcpProposalRegistration.scala:72 / BfmFittingPartial.scala:70)
    ch->root = static_cast<MixtureProposal*>(own(new MixtureProposal()));
    if (poseMix) ch->root->add(cfg->w_pose, poseMix);
    if (icpMix) ch->root->add(cfg->w_icp, icpMix);
    if (rwMix) ch->root->add(cfg->w_rw, rwMix);
    if (ch->root->generators.empty()) throw NativeError(ICP_ERR_INVALID_ARG, "icp_host_chain_create: no proposals");
    // ProductEvaluators.proximityAnd* (ProductEvaluators.scala:38-94): prior × likelihood
    ch->prior.reset(new ModelPriorEvaluator(ch->r));
    t_prop = std::chrono::steady_clock::now();
    ch->likelihood.reset(new NativeLikelihoodEvaluator(ctx, cfg->eval));
    t_eval = std::chrono::steady_clock::now();
    ch->product.parts = {ch->prior.get(), ch->likelihood.get()};
    ch->mh.reset(new MetropolisHastings(ch->root, &ch->product));
    if (cfg->fused == 3) {
      // the drop-in path as Scalismo drives it: per-method calls, every one handed to the native side, over a chain bound ONCE
      // (icp_chain_bind: the first call of a step submits the whole step, the others find their values parked)
      std::vector<icp_proposal*> hs;
      for (auto* p : ch->icp) hs.push_back(p->h);
      check(icp_chain_bind(ch->likelihood->h, (int)hs.size(), hs.data()), "icp_chain_bind");
      ch->mh->pass_current_through = true;
    } else if (cfg->fused) {
      ch->prefetcher.evaluator = ch->likelihood.get();
      ch->prefetcher.icp = ch->icp;
      ch->prefetcher.whole_step = cfg->fused >= 2;
      ch->mh->prefetcher = &ch->prefetcher;
      if (cfg->fused >= 2)
        for (size_t i = 0; i < ch->icp.size(); ++i) { ch->icp[i]->stepper = &ch->prefetcher; ch->icp[i]->stepperIndex = (int)i; }
    }
    ch->ahead.assign(ch->r, 0.0);
    ch->ahead2.assign(ch->r, 0.0);
    if (cfg->fused == 2) check(icp_ctx_set_idle_hook(ctx, &icp_host_chain::draw_ahead, ch), "icp_ctx_set_idle_hook");
    ch->current.allParameters.assign(theta0, theta0 + 10 + ch->r);
    ch->logger.P = 10 + ch->r;
    ch->current_p = ch->product.logValue(ch->current);
    if (timing) {
      const auto t_end = std::chrono::steady_clock::now();
      auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
      std::fprintf(stderr, "[icp create timing] proposals + walks %.0f us, evaluator %.0f us, initial log value %.0f us\n", us(t_begin, t_prop), us(t_prop, t_eval), us(t_eval, t_end));
    }
    *out = ch;
  });
  if (rc != ICP_OK && ch) delete ch;
  return rc;
}

int icp_host_chain_run(icp_host_chain* ch, int32_t n_steps, double* records) {
  return host_guard([&] {
    if (!ch || n_steps < 0) throw NativeError(ICP_ERR_INVALID_ARG, "icp_host_chain_run");
    ch->logger.out = records;
    ch->runner = std::this_thread::get_id();
    for (int s = 0; s < n_steps; ++s) {  // SamplingRegistration.scala:58-85: chain.iterator(...).take(n)
      StepRandom rnd{ch->seed, (uint64_t)ch->logger.index};
      if (ch->ahead_step == rnd.step) { rnd.ahead = ch->ahead.data(); rnd.n_ahead = (int)ch->ahead.size(); }
      ch->current = ch->mh->next(ch->current, rnd, &ch->logger);
      ch->current_p = ch->mh->cached_current_p;
    }
    ch->logger.out = nullptr;
    // no further step for now: whatever was launched ahead for it is dropped
    if (ch->prefetcher.whole_step && ch->likelihood) (void)icp_chain_step_prelaunch(ch->likelihood->h, 0, nullptr, -1, nullptr, nullptr);
  });
}

namespace {
// A set of chains stepped in lockstep: per step ONE icp_chain_step_batched submission for all members whose proposal is an
// ICP or a random-walk shape proposal, then every member's MetropolisHastings.next with the results parked for it.
// The standard normals of the chains' NEXT step (posterior.sample(), 2·r uniforms, r logarithms, square roots and cosines per
// chain: 2-3 µs of the ≈ 7 µs of host time a chain's step costs in a batch), drawn by a helper thread while the calling thread
// submits and records: they depend on (seed, step) only.  The same StepRandom::normal calls: the same values.
struct NormalsAhead {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  bool stop = false, pending = false;
  std::atomic<bool> done{true};
  int r = 0;
  std::vector<uint64_t> seed, step;      // the job: per chain
  std::vector<std::vector<double>> z;    // its result
  std::vector<uint64_t> z_step;          // the step z[b] belongs to (~0: none)
  void start(size_t B, int rank) {
    r = rank;
    seed.assign(B, 0); step.assign(B, 0); z.assign(B, std::vector<double>(rank)); z_step.assign(B, ~0ull);
    th = std::thread([this] {
      std::unique_lock<std::mutex> lk(mu);
      for (;;) {
        cv.wait(lk, [this] { return stop || pending; });
        if (stop) return;
        pending = false;
        lk.unlock();
        for (size_t b = 0; b < z.size(); ++b) {
          const StepRandom rn{seed[b], step[b]};
          for (int j = 0; j < r; ++j) z[b][j] = rn.normal(j);
          z_step[b] = step[b];
        }
        done.store(true, std::memory_order_release);
        lk.lock();
      }
    });
  }
  void request() {  // (seed / step filled in by the caller, who has seen done == true)
    done.store(false, std::memory_order_relaxed);
    { std::lock_guard<std::mutex> lk(mu); pending = true; }
    cv.notify_one();
  }
  bool ready() const { return done.load(std::memory_order_acquire); }
  ~NormalsAhead() {
    if (th.joinable()) {
      { std::lock_guard<std::mutex> lk(mu); stop = true; }
      cv.notify_one();
      th.join();
    }
  }
};

struct LockstepGroup {
  std::vector<icp_host_chain*> chains;
  NormalsAhead ahead;
  size_t n_icp = 0;
  int r = 0;
  std::vector<StepRandom> rnd;
  std::vector<std::vector<double>> z, prop;
  std::vector<int> member;  // chains of the submission in flight
  std::vector<icp_evaluator*> ev;
  std::vector<icp_proposal*> props;
  std::vector<int32_t> gen, status;
  std::vector<const double*> cur_p, z_p;
  std::vector<double*> prop_p;
  std::vector<double> value, fwd, bwd;
  icp_step_ticket* ticket = nullptr;
  ModelFittingParameters scratch_prop;
  icp_ctx* launch_ctx = nullptr;  // whose stream carries the group's launches (nullptr: the first member's)
  // Chains are independent, so they need not advance at the same pace: a chain whose next ICP proposal would wait for a KL basis
  // that is still being computed on the device (icp_proposal_basis_state: started ahead by the step that proposed its state, 0.6-0.7 ms
  // at rank 200) sits out the round instead of holding the whole submission back, for at most kMaxDefer rounds and never when no
  // other chain could step.  Opt-in (ICP_DEFERRAL=1) for chains of the wide step: measured neutral to harmful, see icp_host_chains_run_batched.
  bool defer_waiting = false;
  static constexpr int kMaxDefer = 3;
  std::vector<int> left, sat_out;   // steps this run still owes per chain; consecutive rounds the chain has sat out
  std::vector<char> active;         // takes part in the round in flight
  bool in_flight = false;
  bool any_left() const { for (int v : left) if (v > 0) return true; return false; }

  void init(int n_steps) {
    left.assign(chains.size(), n_steps);
    sat_out.assign(chains.size(), 0);
    active.assign(chains.size(), 0);
    const size_t B = chains.size();
    n_icp = chains[0]->icp.size();
    r = chains[0]->r;
    rnd.resize(B);
    z.assign(B, std::vector<double>(r));
    prop.assign(B, std::vector<double>(10 + r));
#ifdef ICP_DEV_SWITCHES
    static const bool no_ahead = std::getenv("ICP_NO_NORMALS_AHEAD") != nullptr;  // (developer A/B)
#else
    static const bool no_ahead = false;
#endif
    if (B >= 8 && n_icp > 0 && !no_ahead) ahead.start(B, r);
  }
  // the next step of every member: random numbers, proposal kind, arguments; submission of those that share launches
  void issue() {
    member.clear(); ev.clear(); props.clear(); gen.clear(); cur_p.clear(); z_p.clear(); prop_p.clear();
    in_flight = true;
    // ---- who takes part in this round
    int n_ready = 0;
    std::vector<char> waiting(chains.size(), 0);
    for (size_t b = 0; b < chains.size(); ++b) {
      active[b] = left[b] > 0;
      if (!active[b]) continue;
      icp_host_chain* ch = chains[b];
      if (defer_waiting) {
        const StepRandom r0{ch->seed, (uint64_t)ch->logger.index};
        if (auto* ip = dynamic_cast<NonRigidIcpProposal*>(ch->root->peek(r0, 0)))
          waiting[b] = icp_proposal_basis_state(ip->h, ch->current.data()) == 1;
      }
      if (!waiting[b]) ++n_ready;
    }
    for (size_t b = 0; b < chains.size(); ++b)
      if (active[b] && waiting[b]) {
        if (n_ready > 0 && sat_out[b] < kMaxDefer) { active[b] = 0; ++sat_out[b]; }
        else sat_out[b] = 0;
      } else if (active[b]) sat_out[b] = 0;
    for (size_t b = 0; b < chains.size(); ++b) {
      if (!active[b]) continue;
      icp_host_chain* ch = chains[b];
      rnd[b] = StepRandom{ch->seed, (uint64_t)ch->logger.index};
      ch->prefetcher.submitted_index = -1;
      ProposalGeneratorWithTransition* leaf = ch->root->peek(rnd[b], 0);
      int g = -2;
      if (auto* ip = dynamic_cast<NonRigidIcpProposal*>(leaf)) {
        if (ip->stepper) {
          g = ip->stepperIndex;
          // posterior.sample() (NonRigidIcpProposal.scala:55): drawn ahead by the helper thread if it got that far
          if (ahead.th.joinable() && ahead.ready() && ahead.z_step[b] == rnd[b].step) z[b] = ahead.z[b];
          else for (int j = 0; j < r; ++j) z[b][j] = rnd[b].normal(j);
        }
      } else if (auto* rw = dynamic_cast<RandomShapeUpdateProposal*>(leaf)) {
        g = -1;
        prop[b] = rw->propose(ch->current, rnd[b], 0).allParameters;
      } else if (auto* gp = dynamic_cast<GaussianAxisPoseProposal*>(leaf)) {
        // PoseProposals.scala:31-90: the proposed state is the current one with one pose parameter moved — submitted with the others
        // (the library's wide step takes pose moves side by side with the chains' ICP proposals; configurations it does not cover
        // are stepped one after the other by the same call)
        g = -1;
        prop[b] = gp->propose(ch->current, rnd[b], 0).allParameters;
      }
      if (g == -2 || n_icp == 0) continue;  // anything else: MetropolisHastings::next submits it itself
      member.push_back((int)b);
      ev.push_back(ch->likelihood->h);
      for (auto* p : ch->icp) props.push_back(p->h);
      gen.push_back(g);
      cur_p.push_back(ch->current.data());
      z_p.push_back(z[b].data());
      prop_p.push_back(prop[b].data());
    }
    const int nb = (int)member.size();
    ticket = nullptr;
    if (nb == 0) { draw_ahead(); return; }
    value.assign(nb, 0.0); fwd.assign((size_t)nb * n_icp + 1, 0.0); bwd.assign((size_t)nb * n_icp + 1, 0.0); status.assign(nb, 0);
    check(icp_chain_step_batched_issue(nb, ev.data(), (int)n_icp, props.data(), gen.data(), cur_p.data(), z_p.data(), prop_p.data(),
                                       value.data(), fwd.data(), bwd.data(), status.data(), launch_ctx, &ticket),
          "icp_chain_step_batched_issue");
    draw_ahead();
  }
  // the normals of every member's next step, while this one is on the device
  void draw_ahead() {
    if (!ahead.th.joinable() || !ahead.ready() || defer_waiting) return;
    for (size_t b = 0; b < chains.size(); ++b) { ahead.seed[b] = chains[b]->seed; ahead.step[b] = (uint64_t)chains[b]->logger.index + 1; }
    ahead.request();
  }
  // results of the submission in flight, then SamplingRegistration.scala:58-85 for every member
  void finish() {
    if (ticket) {
      icp_step_ticket* t = ticket;
      ticket = nullptr;
      const int st = icp_chain_step_batched_collect(t);
      if (st != ICP_OK) check(st, "icp_chain_step_batched_collect");
      for (size_t k = 0; k < member.size(); ++k) {
        icp_host_chain* ch = chains[member[k]];
        ModelFittingParameters& pr = scratch_prop;  // (keeps its storage from chain to chain)
        pr.allParameters = prop[member[k]];
        if (gen[k] >= 0) pr.generatedBy = ch->icp[gen[k]]->generatedBy;
        else {
          ProposalGeneratorWithTransition* leaf = ch->root->peek(rnd[member[k]], 0);
          if (auto* rw = dynamic_cast<RandomShapeUpdateProposal*>(leaf)) pr.generatedBy = rw->generatedBy;
          else if (auto* gp = dynamic_cast<GaussianAxisPoseProposal*>(leaf)) pr.generatedBy = gp->generatedBy;
        }
        ch->prefetcher.park(ch->current, pr, status[k], value[k], fwd.data() + k * n_icp, bwd.data() + k * n_icp);
        if (gen[k] >= 0) { ch->prefetcher.submitted_index = gen[k]; ch->prefetcher.submitted_z = z[member[k]]; }
      }
    }
    for (size_t b = 0; b < chains.size(); ++b) {
      if (!active[b]) continue;  // (sat this round out, or has taken all its steps)
      icp_host_chain* ch = chains[b];
      ch->current = ch->mh->next(ch->current, rnd[b], &ch->logger);
      ch->current_p = ch->mh->cached_current_p;
      --left[b];
    }
    in_flight = false;
  }
  void abandon() {  // after a failure elsewhere: the submission in flight is waited for and dropped
    if (ticket) { (void)icp_chain_step_batched_collect(ticket); ticket = nullptr; }
  }
};
}  // namespace

// -> ICP_OK: the chains have advanced n_steps on the device; anything else: nothing has happened (the caller steps them on the host)
static int run_on_device(icp_host_chain* const* chains, int32_t n_chains, int32_t n_steps, double* const* records, int32_t* steps_done_out) {
  // ICP_HOST_DEVICE_LOOP: 0 never, 1 whenever covered, unset: from 24 chains on — a group's step is a serial chain of ≈ 300 µs on the
  // device whatever its size (launches 1-5, the decide kernel, the decompositions of the chains that moved), which two groups overlap:
  // 64 chains ≈ 200k it/s against ≈ 150k host-stepped, 128 chains ≈ 245k, 32 chains 128k against 113k; 16 chains 75k against 74k, 8 chains
  // 41k against 43k (the host-stepped form lets every chain's first launch wait for its own decomposition only)
  // Chains of the five merged launches ABOVE rank 64 (apps/femur/StdIcpVsChainICPrandomInitComparisonAll.scala at rank 101) take the
  // loop from two chains on: host-stepped, each of their steps waits for a decomposition the batch cannot start ahead (3 chains at rank
  // 100: 3.7k it/s in the loop, 2.5k host-stepped; 24 chains 7.3k against 5.0k — tools/r5_wide_loop.py).  Chains of the wide step (open
  // targets, rank 200): 30 chains 1.7 ms per step in the loop against 2.0 host-stepped, 10 chains 5.2k it/s against 6.8k — at first from 24 on.
  static const int mode = std::getenv("ICP_HOST_DEVICE_LOOP") ? std::atoi(std::getenv("ICP_HOST_DEVICE_LOOP")) : -1;
  icp_host_chain* c0 = chains[0];
  if (!c0) return ICP_ERR_INVALID_ARG;
  // (… measured again at the end of round 5 — restate pass, operand-row regression, matrix-core back-transformation —, 200 steps, whole
  // runs: the face configuration at N = 28,561 2 chains 2.5k it/s in the loop against 2.0k host-stepped, 3: 3.5k / 3.3k, 4: 4.4k / 4.7k,
  // 6: 6.0k / 4.4k, 10: 8.8k / 6.6k, 16: 11.7k / 11.9k, 20: 13.1k / 11.3k; an open target at rank 40: 4 chains 16.9k / 7.4k, 10 chains
  // 34.8k / 12.3k.)
  int from = 24;
  if (mode < 0 && !c0->icp.empty()) {
    std::vector<icp_proposal*> hs;
    for (auto* p : c0->icp) hs.push_back(p->h);
    const int path = icp_chain_step_path(c0->likelihood->h, (int)hs.size(), hs.data());
    // (ranks <= 64 on the wide step: from eight — the loop's warm-started Jacobi iteration starts from another basis than the
    // host-stepped step's, which decomposes ahead: states equal to 1e-11, not bit for bit, and small submissions are what the tests
    // compare chain by chain)
    if (c0->r > 64 && (path == 0 || path == 1)) from = 2;
    else if (path == 1) from = 8;
  }
  if (mode == 0 || (mode < 0 && n_chains < from)) return ICP_ERR_INVALID_ARG;
  const size_t n_icp = c0->icp.size();
  for (int b = 0; b < n_chains; ++b) {
    icp_host_chain* ch = chains[b];
    if (!ch || !ch->prefetcher.whole_step || ch->icp.size() != n_icp || n_icp < 1 || n_icp > 2 || ch->r != c0->r) return ICP_ERR_INVALID_ARG;
    const icp_host_chain_config &a = ch->cfg, &z = c0->cfg;
    if (!(a.w_icp > 0)) return ICP_ERR_INVALID_ARG;
    if (a.w_icp != z.w_icp || a.w_rw != z.w_rw || a.rw_sigma != z.rw_sigma || a.w_pose != z.w_pose) return ICP_ERR_INVALID_ARG;
    for (int k = 0; k < 3 && a.w_pose > 0; ++k)
      if (a.pose_rot_sigma[k] != z.pose_rot_sigma[k] || a.pose_trans_sigma[k] != z.pose_trans_sigma[k]) return ICP_ERR_INVALID_ARG;
    for (size_t i = 0; i < n_icp; ++i)
      if (a.icp_weight[i] != z.icp_weight[i]) return ICP_ERR_INVALID_ARG;
  }
  icp_mh_mixture mix{};
  mix.struct_size = sizeof(icp_mh_mixture);
  for (size_t i = 0; i < n_icp; ++i) mix.icp_weight[i] = c0->cfg.icp_weight[i];
  mix.w_icp = c0->cfg.w_icp; mix.w_rw = c0->cfg.w_rw; mix.rw_sigma = c0->cfg.rw_sigma;
  mix.w_pose = c0->cfg.w_pose > 0 ? c0->cfg.w_pose : 0.0;  // (the six pose walks: on the device too, include/icp_sincos.h)
  for (int k = 0; k < 3; ++k) { mix.pose_rot_sigma[k] = c0->cfg.pose_rot_sigma[k]; mix.pose_trans_sigma[k] = c0->cfg.pose_trans_sigma[k]; }
  std::vector<icp_evaluator*> ev(n_chains);
  std::vector<icp_proposal*> props((size_t)n_chains * n_icp);
  std::vector<uint64_t> seeds(n_chains);
  std::vector<int64_t> first(n_chains), acc(n_chains, 0);
  std::vector<std::vector<double>> th(n_chains);
  std::vector<double*> thp(n_chains);
  std::vector<double> logp(n_chains);
  for (int b = 0; b < n_chains; ++b) {
    icp_host_chain* ch = chains[b];
    ev[b] = ch->likelihood->h;
    for (size_t i = 0; i < n_icp; ++i) props[(size_t)b * n_icp + i] = ch->icp[i]->h;
    seeds[b] = ch->seed;
    first[b] = ch->logger.index;
    th[b] = ch->current.allParameters;
    thp[b] = th[b].data();
    logp[b] = ch->current_p;
    (void)icp_chain_step_prelaunch(ch->likelihood->h, 0, nullptr, -1, nullptr, nullptr);  // (a half step launched ahead: dropped)
  }
  // (in pieces of at most 4,096 steps: the run keeps its records — 14 + r doubles per chain and step — in device memory until it ends)
  constexpr int32_t kPiece = 4096;
  std::vector<double*> recp(n_chains, nullptr);
  for (int32_t done = 0; done < n_steps; done += kPiece) {
    const int32_t n = std::min(kPiece, n_steps - done);
    std::vector<int64_t> acc_piece(n_chains, 0);
    if (records)
      for (int b = 0; b < n_chains; ++b) recp[b] = records[b] ? records[b] + (size_t)done * (ICP_HOST_RECORD_HEADER + 10 + c0->r) : nullptr;
    const int rc = icp_chains_run_on_device(n_chains, ev.data(), (int32_t)n_icp, props.data(), &mix, seeds.data(), first.data(), thp.data(),
                                            logp.data(), n, records ? recp.data() : nullptr, acc_piece.data());
    if (rc != ICP_OK) {
      if (done == 0) return rc;  // nothing has happened: the caller steps the chains on the host
      // a later piece stopped (a tail that did not contract, say): the chains stand where the last complete piece left them;
      // book that, and let the host path take the rest
      n_steps = done;
      break;
    }
    for (int b = 0; b < n_chains; ++b) { first[b] += n; acc[b] += acc_piece[b]; }
  }
  const int32_t steps_done = n_steps;
  *steps_done_out = steps_done;
  for (int b = 0; b < n_chains; ++b) {
    icp_host_chain* ch = chains[b];
    ch->current.allParameters = th[b];
    // generatedBy of the state = the proposal that produced the last ACCEPTED sample; the records carry the leaf ids
    ch->current_p = logp[b];
    ch->logger.index += steps_done;
    ch->logger.n_accept += acc[b];
    ch->mh->cached_current = ch->current;
    ch->mh->cached_current_p = logp[b];
    ch->mh->have_current = true;
    ch->prefetcher.have = false;
    ch->prefetcher.submitted_index = -1;
    ch->likelihood->has_prefetch = false;
    for (auto* p : ch->icp) { p->prefetched[0].valid = false; p->prefetched[1].valid = false; }
    ch->ahead_step = ~0ull; ch->ahead2_step = ~0ull;
  }
  return ICP_OK;
}

int icp_host_chains_run_batched(icp_host_chain* const* chains, int32_t n_chains, int32_t n_steps, double* const* records) {
  // a lone chain is better off with the pipelined single-chain step (launches of the next step issued ahead)
  if (chains && n_chains == 1 && chains[0]) return icp_host_chain_run(chains[0], n_steps, records ? records[0] : nullptr);
  // pose-free mixtures: the whole loop on the device (icp_chains_run_on_device: mixture draw, proposals' inputs, MetropolisHastings.next
  // and the records by kernels of the step's own stream; the host only enqueues).  What it does not cover (and, by default, fewer
  // than 24 chains: see run_on_device) comes back with ICP_ERR_INVALID_ARG and takes the lockstep path below.
  std::vector<double*> rest;  // (records of the steps the device loop did not take)
  if (chains && n_chains >= 1 && n_steps > 0) {
    int32_t n_dev = 0;
    const int rc_dev = run_on_device(chains, n_chains, n_steps, records, &n_dev);
    if (rc_dev == ICP_OK) {
      if (n_dev >= n_steps) return ICP_OK;
      if (records) {
        rest.assign(records, records + n_chains);
        for (int b = 0; b < n_chains; ++b)
          if (rest[b]) rest[b] += (size_t)n_dev * (ICP_HOST_RECORD_HEADER + 10 + chains[b]->r);
        records = rest.data();
      }
      n_steps -= n_dev;
    }
  }
  // (every group submits through groups[0]'s launch context, whose rings hold ICP_MAX_BATCHES_IN_FLIGHT batches: never more groups)
  constexpr int kMaxGroups = ICP_MAX_BATCHES_IN_FLIGHT;
  LockstepGroup groups[kMaxGroups];
  int rc = host_guard([&] {
    if (!chains || n_chains < 1 || n_steps < 0) throw NativeError(ICP_ERR_INVALID_ARG, "icp_host_chains_run_batched");
    for (int b = 0; b < n_chains; ++b) {
      icp_host_chain* ch = chains[b];
      if (!ch || !ch->prefetcher.whole_step || ch->icp.size() != chains[0]->icp.size() || ch->r != chains[0]->r)
        throw NativeError(ICP_ERR_INVALID_ARG, "icp_host_chains_run_batched: chains must share one configuration with fused = 2");
      ch->logger.out = records ? records[b] : nullptr;
      ch->runner = std::this_thread::get_id();
    }
    // Several groups a fraction of a step apart: while one group's launches run, the other groups' decompositions do, and
    // the host prepares their submissions (a step's first launch waits ≈ 100 µs for the decompositions of the chains that
    // moved).  Few chains stay in one group: the launches of a part of them would not fill the device.
    static const int forced = std::getenv("ICP_LOCKSTEP_GROUPS") ? std::atoi(std::getenv("ICP_LOCKSTEP_GROUPS")) : 0;  // (operational switch)
    // (measured, tools/ab_groups64.sh: two groups from 8 chains to 64 — 16 chains: 70k against 56k it/s with one or three, 32: 108k against
    // 90k with three, 64: 134k against 117k; beyond that groups of about 32: 96 chains 135k, 128 chains 142k with four)
    int n_groups = forced > 0 ? forced : (n_chains > 80 ? (n_chains + 31) / 32 : n_chains >= 8 ? 2 : 1);
    bool wide = false;
    // Chains that take the WIDE step (open targets, the Hausdorff evaluator, rank 200, pose walks: a step of 0.5-1.5 ms whose
    // one-workgroup factorisations and decompositions run side by side for all chains of a submission) are better off in ONE group
    // per 32 chains: 10 chains of the face configuration 6.65k it/s in one group, 5.6k in two, 4.9k in three, 4.1k in four — every
    // launch costs the same whatever it carries, and the groups' chip-wide launches share one stream anyway (20 chains in one group
    // 9.1k it/s, 30: 10.5k, 40 in two groups of 20: 10.3k; tools/r4_many.sh).
    if (!chains[0]->icp.empty()) {
      std::vector<icp_proposal*> hs;
      for (auto* p : chains[0]->icp) hs.push_back(p->h);
      wide = icp_chain_step_path(chains[0]->likelihood->h, (int)hs.size(), hs.data()) == 1;
      static const int wide_group = std::getenv("ICP_WIDE_GROUP") ? std::max(1, std::atoi(std::getenv("ICP_WIDE_GROUP"))) : 32;
      if (wide && forced <= 0) n_groups = (n_chains + wide_group - 1) / wide_group;
    }
    n_groups = std::max(1, std::min(std::min(n_groups, kMaxGroups), n_chains));
    for (int b = 0; b < n_chains; ++b) groups[(size_t)b * n_groups / n_chains].chains.push_back(chains[b]);
    for (int g = 0; g < n_groups; ++g) groups[g].init(n_steps);
    // all groups' launches on ONE stream, one group behind the other: side by side the big launches of two groups slow each
    // other down more than the overlap gains (regression 18 -> 80 µs beside the other group's filter); the decompositions
    // keep their own streams
    for (int g = 1; g < n_groups; ++g) groups[g].launch_ctx = groups[0].chains[0]->ctx;
    // steady state: every group has a submission in flight; they are collected and renewed in turn.  (A host thread per group
    // was tried — the kernel trace shows the launch stream idle a third of the time at 64 chains: sequences of 152 µs every 236 µs,
    // the host needs ≈ 7 µs per chain and step — and measured slower at every size, 16 chains 58k against 71k it/s, 64 chains 126k
    // against 129k: the threads' launches and their waits meet in the runtime.)
    // (measured, tools/r4_defer.sh, 10 chains of the face configuration: 6.5-6.9k it/s either way in the steady state, 4.0k against 6.5k
    // over a chain's first 50 steps, where most chains wait most rounds and the rounds of the few that do not cost as much as full ones —
    // a round's cost is its launches, not its chains.  Off unless asked for.)
    static const bool defer = std::getenv("ICP_DEFERRAL") != nullptr && std::atoi(std::getenv("ICP_DEFERRAL")) != 0;
    for (int g = 0; g < n_groups; ++g) groups[g].defer_waiting = wide && defer && groups[g].chains.size() > 1;
    if (n_steps > 0)
      for (int g = 0; g < n_groups; ++g) groups[g].issue();
    for (bool busy = n_steps > 0; busy;) {  // (rounds: a group's chains may advance at different paces, see LockstepGroup::defer_waiting)
      busy = false;
      for (int g = 0; g < n_groups; ++g) {
        if (groups[g].in_flight) groups[g].finish();
        if (groups[g].any_left()) { groups[g].issue(); busy = true; }
      }
    }
  });
  for (auto& g : groups) g.abandon();
  if (chains)
    for (int b = 0; b < n_chains; ++b)
      if (chains[b]) chains[b]->logger.out = nullptr;
  return rc;
}

int icp_host_chain_log_transition(icp_host_chain* ch, const double* theta_from, const double* theta_to, double* out) {
  return host_guard([&] {
    if (!ch || !theta_from || !theta_to || !out) throw NativeError(ICP_ERR_INVALID_ARG, "icp_host_chain_log_transition");
    ModelFittingParameters a, b;
    a.allParameters.assign(theta_from, theta_from + 10 + ch->r);
    b.allParameters.assign(theta_to, theta_to + 10 + ch->r);
    *out = ch->root->logTransitionProbability(a, b);
  });
}

int icp_host_pose_mixture_log_transition(int32_t n_params, const double* rot_sigma, const double* trans_sigma, const double* theta_from,
                                         const double* theta_to, double* out) {
  return host_guard([&] {
    if (n_params < 10 || !rot_sigma || !trans_sigma || !theta_from || !theta_to || !out)
      throw NativeError(ICP_ERR_INVALID_ARG, "icp_host_pose_mixture_log_transition");
    std::vector<std::unique_ptr<ProposalGeneratorWithTransition>> owned;
    MixtureProposal* mix = mixed_random_pose_proposal([&](ProposalGeneratorWithTransition* p) { owned.emplace_back(p); return p; }, rot_sigma, trans_sigma);
    ModelFittingParameters a, b;
    a.allParameters.assign(theta_from, theta_from + n_params);
    b.allParameters.assign(theta_to, theta_to + n_params);
    *out = mix->logTransitionProbability(a, b);
  });
}

int icp_host_pose_mixture_propose(int32_t n_params, const double* rot_sigma, const double* trans_sigma, const double* theta, uint64_t seed,
                                  uint64_t step, double* theta_out, int32_t* leaf_out, char* name_out, int32_t name_len) {
  return host_guard([&] {
    if (n_params < 10 || !rot_sigma || !trans_sigma || !theta || !theta_out)
      throw NativeError(ICP_ERR_INVALID_ARG, "icp_host_pose_mixture_propose");
    std::vector<std::unique_ptr<ProposalGeneratorWithTransition>> owned;
    MixtureProposal* mix = mixed_random_pose_proposal([&](ProposalGeneratorWithTransition* p) { owned.emplace_back(p); return p; }, rot_sigma, trans_sigma);
    ModelFittingParameters cur;
    cur.allParameters.assign(theta, theta + n_params);
    const StepRandom rnd{seed, step};
    const ModelFittingParameters prop = mix->propose(cur, rnd, 1);  // depth 1: the pose mixture sits inside the chain's outer mixture
    std::memcpy(theta_out, prop.data(), sizeof(double) * n_params);
    if (leaf_out) *leaf_out = mix->lastLeaf();
    if (name_out && name_len > 0) std::snprintf(name_out, (size_t)name_len, "%s", prop.generatedBy.c_str());
  });
}

int icp_host_scala_double(double x, char* out, int32_t out_len) {
  if (!out || out_len <= 0) return ICP_ERR_INVALID_ARG;
  std::snprintf(out, (size_t)out_len, "%s", scala_double(x).c_str());
  return ICP_OK;
}

int icp_host_chain_state(icp_host_chain* ch, double* theta_out, double* logp_out, int64_t* steps_done, int64_t* accepted) {
  if (!ch) return ICP_ERR_INVALID_ARG;
  if (theta_out) std::memcpy(theta_out, ch->current.data(), sizeof(double) * (10 + ch->r));
  if (logp_out) *logp_out = ch->current_p;
  if (steps_done) *steps_done = ch->logger.index;
  if (accepted) *accepted = ch->logger.n_accept;
  return ICP_OK;
}

int icp_host_chain_native_calls(icp_host_chain* ch, int64_t* out) {
  if (!ch || !out) return ICP_ERR_INVALID_ARG;
  out[0] = out[1] = 0;
  for (auto* p : ch->icp) out[0] += p->native_calls;
  if (ch->likelihood) out[1] = ch->likelihood->native_calls;
  out[2] = out[3] = out[4] = 0;
  if (ch->likelihood && ch->cfg.fused == 3) (void)icp_chain_bind_stats(ch->likelihood->h, out + 2);
  return ICP_OK;
}

void icp_host_chain_destroy(icp_host_chain* ch) {
  if (ch && ch->ctx) (void)icp_ctx_set_idle_hook(ch->ctx, nullptr, nullptr);  // (several chains may share a context: last one wins)
  delete ch;
}

}  // extern "C"
