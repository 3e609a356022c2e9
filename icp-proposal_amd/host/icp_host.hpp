// icp_host.hpp — C++ host side above the C ABI: the reference's plug-in interface and its caller, mirrored.
//
// The reference's host language is Scala (no JVM in this image), so the host side is C++ with the reference's
// names, argument meaning and error behaviour (paths relative to the reference's src/main/scala/):
//
//   ModelFittingParameters            api/sampling/ModelFittingParameters.scala:47-66
//   ProposalGenerator / TransitionProbability / DistributionEvaluator   Scalismo traits implemented at
//                                     api/sampling/proposals/NonRigidIcpProposal.scala:42-43, evaluators/*.scala
//   NonRigidIcpProposal               api/sampling/proposals/NonRigidIcpProposal.scala:30-155       (-> icp_proposal_*)
//   RandomShapeUpdateProposal         api/sampling/proposals/RandomShapeUpdateProposal.scala:25-46  (host, O(r))
//   GaussianAxisRotationProposal / GaussianAxisTranslationProposal   api/sampling/proposals/PoseProposals.scala:31-90
//   MixtureProposal                   Scalismo; built at api/sampling/MixedProposalDistributions.scala:29-68
//   ModelPriorEvaluator, ProductEvaluator, likelihood evaluators     api/sampling/ProductEvaluators.scala:28-94
//   MetropolisHastings                Scalismo; constructed at api/sampling/SamplingRegistration.scala:54
//   SamplingRegistration::runfitting  api/sampling/SamplingRegistration.scala:45-93
//
// Everything numerical about the hot path happens behind include/icp_proposal.h; this file only sequences calls,
// draws random numbers (counter-based, reproducible) and does the O(r) host arithmetic the reference also does
// on the host.  It is the measuring harness of bench.py and the template for the Scala adapters.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/icp_proposal.h"

namespace icphost {

// ---------------------------------------------------------------- randomness (counter-based; bit-identical to oracle/icp_oracle.c)

inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

struct StepRandom {  // the random numbers of ONE Metropolis–Hastings step
  uint64_t seed = 0, step = 0;
  // standard normals 0..n_ahead-1 of this step drawn ahead of time (icp_host.cpp: while the previous step was on the device)
  const double* ahead = nullptr;
  int n_ahead = 0;
  double uniform(uint64_t lane) const {
    uint64_t h = splitmix64(splitmix64(splitmix64(seed) ^ (step * 0xD1342543DE82EF95ull)) ^ (lane * 0x2545F4914F6CDD1Dull));
    return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  }
  double normal(uint64_t lane) const {
    if (lane < (uint64_t)n_ahead) return ahead[lane];
    double u1 = uniform(2 * lane + 1000), u2 = uniform(2 * lane + 1001);
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
  }
  // lanes: mixture draw at nesting depth d -> d (d < 2) or d + 1; accept/reject draw -> 2; standard normals -> normal(j)
  static uint64_t mixture_lane(int depth) { return depth < 2 ? (uint64_t)depth : (uint64_t)depth + 1; }
  static constexpr uint64_t kAcceptLane = 2;
};

// ---------------------------------------------------------------- chain state

struct ModelFittingParameters {
  std::vector<double> allParameters;  // [s | t(3) | phi,theta,psi | centre(3) | c(r)]  (ModelFittingParameters.scala:64)
  std::string generatedBy = "Anonymous";
  int rank() const { return (int)allParameters.size() - 10; }
  const double* data() const { return allParameters.data(); }
  const double* shape() const { return allParameters.data() + 10; }
  bool sameButShape(const ModelFittingParameters& o) const {  // NonRigidIcpProposal.scala:72
    return std::memcmp(allParameters.data(), o.allParameters.data(), sizeof(double) * 10) == 0;
  }
  bool operator==(const ModelFittingParameters& o) const {
    return allParameters.size() == o.allParameters.size() &&
           std::memcmp(allParameters.data(), o.allParameters.data(), sizeof(double) * allParameters.size()) == 0;
  }
};

struct NativeError : std::runtime_error {
  int status;
  NativeError(int s, const std::string& where)
      : std::runtime_error(where + ": " + icp_status_string(s) + " (" + icp_last_error() + ")"), status(s) {}
};
inline void check(int status, const char* where) {
  if (status != ICP_OK) throw NativeError(status, where);
}

// ---------------------------------------------------------------- Scalismo traits

struct DistributionEvaluator {
  virtual ~DistributionEvaluator() = default;
  virtual double logValue(const ModelFittingParameters& sample) = 0;
};

struct ProposalGeneratorWithTransition {
  virtual ~ProposalGeneratorWithTransition() = default;
  // depth = nesting level inside MixtureProposals (selects the random lane)
  virtual ModelFittingParameters propose(const ModelFittingParameters& current, const StepRandom& rnd, int depth) = 0;
  virtual double logTransitionProbability(const ModelFittingParameters& from, const ModelFittingParameters& to) = 0;
  // Scalismo TransitionProbability.logTransitionRatio
  double logTransitionRatio(const ModelFittingParameters& from, const ModelFittingParameters& to) {
    double fw = logTransitionProbability(from, to), bw = logTransitionProbability(to, from);
    if (std::isnan(fw) || std::isnan(bw)) throw std::runtime_error("NaN transition Probability!");
    if (std::isinf(fw) && fw < 0 && std::isinf(bw) && bw < 0) return 0.0;
    return fw - bw;
  }
  // the leaf proposal that propose(·, rnd, depth) would draw from, without drawing (mixtures descend; leaves are themselves)
  virtual ProposalGeneratorWithTransition* peek(const StepRandom&, int) { return this; }
  // index of the leaf proposal that generated the last sample (for logs)
  virtual int lastLeaf() const { return leafId; }
  int leafId = -1;
};

// ---------------------------------------------------------------- evaluators

struct ModelPriorEvaluator : DistributionEvaluator {  // evaluators/ModelPriorEvaluator.scala:24-31
  explicit ModelPriorEvaluator(int rank) : rank(rank) {}
  double logValue(const ModelFittingParameters& theta) override {
    double out;
    check(icp_prior_log_value(rank, theta.data(), &out), "icp_prior_log_value");
    return out;
  }
  int rank;
};

// IndependentPointDistanceEvaluator / HausdorffDistanceEvaluator / CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator
struct NativeLikelihoodEvaluator : DistributionEvaluator {
  NativeLikelihoodEvaluator(icp_ctx* ctx, const icp_evaluator_params& prm) {
    check(icp_evaluator_create(ctx, &prm, &h), "icp_evaluator_create");
  }
  ~NativeLikelihoodEvaluator() override { icp_evaluator_destroy(h); }
  double logValue(const ModelFittingParameters& sample) override {
    if (has_prefetch && prefetched_for == sample) return prefetched_value;
    double out;
    ++native_calls;
    check(icp_evaluator_log_value(h, sample.data(), &out, nullptr), "icp_evaluator_log_value");
    return out;
  }
  icp_evaluator* h = nullptr;
  int64_t native_calls = 0;  // icp_evaluator_log_value calls made (bench.py: calls per step of the per-method modes)
  bool has_prefetch = false;
  ModelFittingParameters prefetched_for;
  double prefetched_value = 0.0;
};

struct ProductEvaluator : DistributionEvaluator {  // Scalismo ProductEvaluator: sum of log values (ProductEvaluators.scala:45-48)
  std::vector<DistributionEvaluator*> parts;
  double logValue(const ModelFittingParameters& s) override {
    double v = 0.0;
    for (auto* p : parts) v += p->logValue(s);
    return v;
  }
};

// ---------------------------------------------------------------- proposals

struct ChainPrefetcher;

struct NonRigidIcpProposal : ProposalGeneratorWithTransition {  // NonRigidIcpProposal.scala:30-41
  NonRigidIcpProposal(icp_ctx* ctx, const icp_proposal_params& prm, std::string generatedBy)
      : generatedBy(std::move(generatedBy)) {
    check(icp_proposal_create(ctx, &prm, &h), "icp_proposal_create");
  }
  ~NonRigidIcpProposal() override { icp_proposal_destroy(h); }
  ModelFittingParameters propose(const ModelFittingParameters& theta, const StepRandom& rnd, int) override;
  double logTransitionProbability(const ModelFittingParameters& from, const ModelFittingParameters& to) override {
    for (auto& pf : prefetched)
      if (pf.valid && pf.from == from && pf.to == to) return pf.value;
    double out;
    ++native_calls;
    check(icp_proposal_log_transition(h, from.data(), to.data(), &out), "icp_proposal_log_transition");
    return out;
  }
  icp_proposal* h = nullptr;
  int64_t native_calls = 0;  // icp_proposal_propose + icp_proposal_log_transition calls made
  std::string generatedBy;
  ChainPrefetcher* stepper = nullptr;  // set: propose() submits the WHOLE step (icp_chain_step) and parks the other results
  int stepperIndex = -1;
  struct Prefetched {
    bool valid = false;
    ModelFittingParameters from, to;
    double value = 0.0;
  } prefetched[2];
  std::vector<double> z_scratch;  // the standard normals of the sample being drawn (one chain = one thread)
};

struct RandomShapeUpdateProposal : ProposalGeneratorWithTransition {  // RandomShapeUpdateProposal.scala:25-46
  RandomShapeUpdateProposal(double stdev, std::string generatedBy) : stdev(stdev), generatedBy(std::move(generatedBy)) {}
  ModelFittingParameters propose(const ModelFittingParameters& theta, const StepRandom& rnd, int) override {
    ModelFittingParameters out = theta;
    for (int j = 0; j < theta.rank(); ++j) out.allParameters[10 + j] = theta.allParameters[10 + j] + stdev * rnd.normal(j);  // :31-35
    out.generatedBy = generatedBy;
    return out;
  }
  double logTransitionProbability(const ModelFittingParameters& from, const ModelFittingParameters& to) override {
    if (!from.sameButShape(to)) return -std::numeric_limits<double>::infinity();  // :38-40
    const int r = from.rank();
    double nn = 0.0;
    for (int j = 0; j < r; ++j) { double d = to.allParameters[10 + j] - from.allParameters[10 + j]; nn += d * d; }
    return -0.5 * nn / (stdev * stdev) - 0.5 * (r * std::log(2.0 * M_PI) + r * std::log(stdev * stdev));  // MVN(0, σ²I).logpdf
  }
  double stdev;
  std::string generatedBy;
};

// PoseProposals.scala:31-62 (rotation about one Euler axis) and :64-90 (translation along one axis): 1-D Gaussian walks.
// allParameters = [s | t(3) | rotation._1, _2, _3 | centre(3) | c] (ModelFittingParameters.scala:28-36,64); RollAxis perturbs
// rotation._1 = theta[4], PitchAxis _2 = theta[5], YawAxis _3 = theta[6] (PoseProposals.scala:39-41); translation axis a theta[1 + a].
struct GaussianAxisPoseProposal : ProposalGeneratorWithTransition {
  GaussianAxisPoseProposal(int param_index, double stdev, std::string generatedBy)
      : index(param_index), group_begin(param_index >= 4 ? 4 : 1), stdev(stdev), generatedBy(std::move(generatedBy)) {}
  ModelFittingParameters propose(const ModelFittingParameters& theta, const StepRandom& rnd, int) override {
    ModelFittingParameters out = theta;
    out.allParameters[index] = theta.allParameters[index] + stdev * rnd.normal(0);  // :39-41, :72-74
    out.generatedBy = generatedBy;
    return out;
  }
  // :46-60 / :77-88.  The reference resets the WHOLE rotation triple (:47) or the WHOLE translation (:78) of `to` to `from`'s before
  // it compares the parameter vectors: -inf only when something OUTSIDE the proposal's own group differs.  Inside the group
  // only this proposal's axis enters the residual — the Yaw walk evaluated on a Roll move returns logPdf(0), finite, and
  // takes part in the mixture's log-sum-exp.
  double logTransitionProbability(const ModelFittingParameters& from, const ModelFittingParameters& to) override {
    for (size_t i = 0; i < from.allParameters.size(); ++i)
      if (((int)i < group_begin || (int)i >= group_begin + 3) && from.allParameters[i] != to.allParameters[i])
        return -std::numeric_limits<double>::infinity();
    double d = (to.allParameters[index] - from.allParameters[index]) / stdev;  // breeze Gaussian(0, σ).logPdf(residual)
    return -d * d / 2.0 - (std::log(std::sqrt(2.0 * M_PI)) + std::log(stdev));
  }
  int index;        // 1..3 translation x/y/z, 4..6 rotation._1/_2/_3 (roll / pitch / yaw)
  int group_begin;  // first parameter of the group the reference resets before comparing (1: translation, 4: rotation)
  double stdev;
  std::string generatedBy;
};

// Scalismo MixtureProposal: component by a uniform draw against the cumulative normalised weights;
// transition density = log-sum-exp over ALL components.
struct MixtureProposal : ProposalGeneratorWithTransition {
  void add(double weight, ProposalGeneratorWithTransition* g) { weights.push_back(weight); generators.push_back(g); }
  size_t pick_component(const StepRandom& rnd, int depth) const {
    double wsum = 0.0;
    for (double w : weights) wsum += w;
    const double u = rnd.uniform(StepRandom::mixture_lane(depth));
    double acc = 0.0;
    for (size_t i = 0; i < generators.size(); ++i) {
      acc += weights[i] / wsum;
      if (acc >= u) return i;
    }
    return generators.size() - 1;
  }
  ProposalGeneratorWithTransition* peek(const StepRandom& rnd, int depth) override {
    return generators[pick_component(rnd, depth)]->peek(rnd, depth + 1);
  }
  ModelFittingParameters propose(const ModelFittingParameters& current, const StepRandom& rnd, int depth) override {
    const size_t pick = pick_component(rnd, depth);
    ModelFittingParameters out = generators[pick]->propose(current, rnd, depth + 1);
    last = generators[pick];
    return out;
  }
  double logTransitionProbability(const ModelFittingParameters& from, const ModelFittingParameters& to) override {
    double wsum = 0.0, mx = -std::numeric_limits<double>::infinity();
    for (double w : weights) wsum += w;
    double t_small[8];
    std::vector<double> t_big;
    if (generators.size() > 8) t_big.resize(generators.size());
    double* t = generators.size() > 8 ? t_big.data() : t_small;
    for (size_t i = 0; i < generators.size(); ++i) {
      t[i] = generators[i]->logTransitionProbability(from, to);
      if (std::isnan(t[i])) throw std::runtime_error("NaN transition probability encountered!");
      if (t[i] > mx) mx = t[i];
    }
    if (std::isinf(mx) && mx < 0) return mx;
    double s = 0.0;
    for (size_t i = 0; i < generators.size(); ++i) s += (weights[i] / wsum) * std::exp(t[i] - mx);
    return std::log(s) + mx;
  }
  int lastLeaf() const override { return last ? last->lastLeaf() : -1; }
  std::vector<double> weights;
  std::vector<ProposalGeneratorWithTransition*> generators;
  ProposalGeneratorWithTransition* last = nullptr;
};

// ---------------------------------------------------------------- the caller

struct AcceptRejectLogger {  // api/sampling/loggers/JSONAcceptRejectLogger.scala:93-106 (record sink only)
  virtual ~AcceptRejectLogger() = default;
  virtual void accept(const ModelFittingParameters& current, const ModelFittingParameters& sample, int leaf, double logp) = 0;
  virtual void reject(const ModelFittingParameters& current, const ModelFittingParameters& sample, int leaf, double logp) = 0;
};

// Optional accelerator: ONE native submission (icp_chain_eval_step) computes the likelihood of the proposal and the
// forward/backward transition densities of every ICP proposal, and parks them where the per-method calls of
// MetropolisHastings::next find them.  Semantics are unchanged; only host<->device round trips are saved.
struct ChainPrefetcher {
  NativeLikelihoodEvaluator* evaluator = nullptr;
  std::vector<NonRigidIcpProposal*> icp;
  bool whole_step = false;  // true: icp_chain_step (propose + evaluation in one submission); false: icp_chain_eval_step
  bool have = false;        // results of (parked_cur -> parked_prop) are parked
  ModelFittingParameters parked_cur, parked_prop;
  // a whole step submitted ahead of MetropolisHastings::next by the batched runner (icp_host_chains_run_batched): step()
  // hands its proposal out when asked for exactly that step
  int submitted_index = -1;
  std::vector<double> submitted_z;

  void park(const ModelFittingParameters& cur, const ModelFittingParameters& prop, int st, double value, const std::vector<double>& fwd,
            const std::vector<double>& bwd) {
    park(cur, prop, st, value, fwd.data(), bwd.data());
  }
  // (field by field: the parked copies keep their storage from step to step)
  void park(const ModelFittingParameters& cur, const ModelFittingParameters& prop, int st, double value, const double* fwd, const double* bwd) {
    evaluator->has_prefetch = st == ICP_OK;
    evaluator->prefetched_for = prop;
    evaluator->prefetched_value = value;
    for (size_t i = 0; i < icp.size(); ++i) {
      NonRigidIcpProposal::Prefetched& f = icp[i]->prefetched[0];
      f.valid = true; f.from = cur; f.to = prop; f.value = fwd[i];
      NonRigidIcpProposal::Prefetched& b = icp[i]->prefetched[1];
      b.valid = true; b.from = prop; b.to = cur; b.value = bwd[i];
    }
    have = true;
    parked_cur = cur;
    parked_prop = prop;
  }
  void prefetch(const ModelFittingParameters& cur, const ModelFittingParameters& prop) {
    if (have && parked_cur == cur && parked_prop == prop) return;  // already submitted by the generating proposal
    std::vector<icp_proposal*> hs;
    for (auto* p : icp) hs.push_back(p->h);
    std::vector<double> fwd(icp.size() + 1), bwd(icp.size() + 1);
    double value;
    int st;
    if (whole_step) {
      ModelFittingParameters tmp = prop;
      st = icp_chain_step(evaluator->h, (int)icp.size(), hs.data(), -1, cur.data(), nullptr, tmp.allParameters.data(), &value,
                          fwd.data(), bwd.data());
    } else {
      st = icp_chain_eval_step(evaluator->h, (int)icp.size(), hs.data(), cur.data(), prop.data(), &value, fwd.data(), bwd.data());
    }
    if (st != ICP_OK && st != ICP_ERR_EMPTY) check(st, "icp_chain_eval_step");
    park(cur, prop, st, value, fwd, bwd);
  }
  // propose from icp[index] AND evaluate the proposal, one native call
  ModelFittingParameters step(int index, const ModelFittingParameters& cur, const double* z) {
    if (have && submitted_index == index && parked_cur == cur && !submitted_z.empty() &&
        std::memcmp(submitted_z.data(), z, sizeof(double) * submitted_z.size()) == 0) {
      submitted_index = -1;
      return parked_prop;
    }
    submitted_index = -1;
    std::vector<icp_proposal*> hs;
    for (auto* p : icp) hs.push_back(p->h);
    std::vector<double> fwd(icp.size() + 1), bwd(icp.size() + 1);
    double value;
    ModelFittingParameters prop;
    prop.allParameters.resize(cur.allParameters.size());
    int st = icp_chain_step(evaluator->h, (int)icp.size(), hs.data(), index, cur.data(), z, prop.allParameters.data(), &value,
                            fwd.data(), bwd.data());
    if (st != ICP_OK && st != ICP_ERR_EMPTY) check(st, "icp_chain_step");
    prop.generatedBy = icp[index]->generatedBy;
    park(cur, prop, st, value, fwd, bwd);
    return prop;
  }
};

inline ModelFittingParameters NonRigidIcpProposal::propose(const ModelFittingParameters& theta, const StepRandom& rnd, int) {
  const int r = theta.rank();
  std::vector<double>& z = z_scratch;
  z.resize(r);
  for (int j = 0; j < r; ++j) z[j] = rnd.normal(j);  // posterior.sample() (:55)
  if (stepper) return stepper->step(stepperIndex, theta, z.data());
  ModelFittingParameters out;
  out.allParameters.resize(theta.allParameters.size());
  ++native_calls;
  check(icp_proposal_propose(h, theta.data(), z.data(), out.allParameters.data(), nullptr), "icp_proposal_propose");
  out.generatedBy = generatedBy;  // :66
  return out;
}

// Scalismo MetropolisHastings.next
struct MetropolisHastings {
  MetropolisHastings(ProposalGeneratorWithTransition* generator, DistributionEvaluator* evaluator)
      : generator(generator), evaluator(evaluator) {}
  ModelFittingParameters next(const ModelFittingParameters& current, const StepRandom& rnd, AcceptRejectLogger* logger,
                              bool* accepted_out = nullptr) {
    // (pass_current_through: the Scala adapters hand EVERY logValue to the native side, whose Memoize(3) answers for the current state
    // — bindings/scala/api/gpu/GpuLikelihoodEvaluator.scala; a chain bound with icp_chain_bind relies on seeing that call)
    const double currentP = !pass_current_through && have_current && cached_current == current ? cached_current_p : evaluator->logValue(current);
    ModelFittingParameters proposal = generator->propose(current, rnd, 0);
    if (prefetcher) prefetcher->prefetch(current, proposal);
    const double proposalP = evaluator->logValue(proposal);
    const double t = generator->logTransitionRatio(current, proposal);
    const double a = proposalP - currentP - t;
    const bool acc = a > 0.0 || rnd.uniform(StepRandom::kAcceptLane) < std::exp(a);
    if (accepted_out) *accepted_out = acc;
    const int leaf = generator->lastLeaf();
    if (acc) {
      if (logger) logger->accept(current, proposal, leaf, proposalP);
      cached_current = proposal; cached_current_p = proposalP; have_current = true;
      return proposal;
    }
    if (logger) logger->reject(current, proposal, leaf, currentP);
    cached_current = current; cached_current_p = currentP; have_current = true;
    return current;
  }
  ProposalGeneratorWithTransition* generator;
  DistributionEvaluator* evaluator;
  ChainPrefetcher* prefetcher = nullptr;
  bool pass_current_through = false;
  // the reference gets this from Memoize(computeLogValue, 3) (evaluators/EvaluationCaching.scala:32)
  bool have_current = false;
  ModelFittingParameters cached_current;
  double cached_current_p = 0.0;
};

}  // namespace icphost
