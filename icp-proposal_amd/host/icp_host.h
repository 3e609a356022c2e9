/* icp_host.h — C entry points of libicp_host.so, the C++ host harness (see icp_host.hpp).
 * NOT part of the drop-in boundary (that is include/icp_proposal.h): this is the build's own caller, mirroring
 * SamplingRegistration.runfitting (api/sampling/SamplingRegistration.scala:45-93) so bench.py and the tests can
 * drive whole chains without a Python interpreter in the per-step loop. */
#ifndef ICP_HOST_H
#define ICP_HOST_H
#include <stdint.h>
#include "../../include/icp_proposal.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct icp_host_chain icp_host_chain;

typedef struct {
  int32_t n_icp;                 /* 0..2 NonRigidIcpProposal components (MixedProposalDistributions.scala:48-68) */
  icp_proposal_params icp[2];
  double icp_weight[2];          /* inner mixture weights (0.5 / 0.5) */
  double w_icp;                  /* outer mixture: ICP mixture            (IcpProposalRegistration.scala:72: 0.90) */
  double w_rw;                   /*                random shape walk       (0.10) */
  double w_pose;                 /*                6-component pose walk   (BfmFittingPartial.scala:70: 0.4; 0 = none) */
  double rw_sigma;               /* RandomShapeUpdateProposal stdev (0.1) */
  double pose_rot_sigma[3];      /* rotYaw, rotPitch, rotRoll (MixedProposalDistributions.scala:29: 0.01): the walks on
                                    rotation._3 = theta[6], _2 = theta[5], _1 = theta[4] (PoseProposals.scala:39-41) */
  double pose_trans_sigma[3];    /* x/y/z stdevs (0.1) */
  icp_evaluator_params eval;     /* likelihood; the shape prior is always multiplied in (ProductEvaluators.scala:38-55) */
  int32_t fused;                 /* 0 = per-method calls; 1 = icp_chain_eval_step prefetch after propose; 2 = the whole step
                                    (propose + evaluation) as ONE icp_chain_step submission; 3 = per-method calls as Scalismo's
                                    MetropolisHastings.next makes them (every logValue handed to the native side) over a chain bound
                                    once with icp_chain_bind — the drop-in path of INTEGRATION.md §2 */
  int32_t sampler;               /* icp_sampler of the ICP proposals: 0 = eigen (the reference's posterior.sample()), 1 = opt-in
                                    Cholesky root (same distribution, no eigen-decomposition; NOT the reference's arithmetic) */
} icp_host_chain_config;

/* fixed-size per-step record (the layout the multi-GPU log gather ships; mirrors jsonLogFormat,
 * api/sampling/loggers/JSONAcceptRejectLogger.scala:35): [index, status(1 accept/0 reject), leaf proposal id,
 * log product value of the chain state after the step, theta(10+r)]  =>  14 + r doubles */
#define ICP_HOST_RECORD_HEADER 4

ICP_API int icp_host_chain_create(icp_ctx *ctx, const icp_host_chain_config *cfg, const double *theta0, uint64_t seed,
                                  icp_host_chain **out);
/* runs n_steps more steps; records [n_steps * (4 + 10 + r)] may be NULL */
ICP_API int icp_host_chain_run(icp_host_chain *chain, int32_t n_steps, double *records);
/* n_steps more steps of n_chains chains in lockstep: per step ONE icp_chain_step_batched submission for all chains
 * whose proposal is an ICP or a random-walk shape proposal (the others step on their own), then every chain's
 * MetropolisHastings.next with those results.  Chain by chain the records are those of icp_host_chain_run.  Every chain
 * needs its own context and fused = 2.  records[b] may be NULL.  From 8 chains on they form two groups (from 24:
 * three) a fraction of a step apart (ICP_LOCKSTEP_GROUPS = 1..4 overrides), so that one group's decompositions run beside
 * the other's launches. */
ICP_API int icp_host_chains_run_batched(icp_host_chain *const *chains, int32_t n_chains, int32_t n_steps,
                                        double *const *records);
ICP_API int icp_host_chain_state(icp_host_chain *chain, double *theta_out, double *logp_out, int64_t *steps_done,
                                 int64_t *accepted);
/* per-method native calls the chain's adapters have made so far: out[0] icp_proposal_propose + icp_proposal_log_transition,
 * out[1] icp_evaluator_log_value, out[2..4] icp_chain_bind_stats (fused = 3; else zeros) */
ICP_API int icp_host_chain_native_calls(icp_host_chain *chain, int64_t *out /* [5] */);
/* the chain's whole proposal mixture: MixtureProposal.logTransitionProbability(from, to) = log-sum-exp over every leaf */
ICP_API int icp_host_chain_log_transition(icp_host_chain *chain, const double *theta_from, const double *theta_to, double *out);
/* MixedProposalDistributions.mixedRandomPoseProposal (MixedProposalDistributions.scala:29-39) by itself — host arithmetic only, no
 * context: its logTransitionProbability, and one propose() with the random numbers of (seed, step) as the chain would draw them
 * (mixture draw on lane 1, perturbation = sigma * normal(0)); leaf_out = 3..8 (Yaw, Pitch, Roll, X, Y, Z), name_out = generatedBy */
ICP_API int icp_host_pose_mixture_log_transition(int32_t n_params, const double *rot_sigma, const double *trans_sigma,
                                                 const double *theta_from, const double *theta_to, double *out);
ICP_API int icp_host_pose_mixture_propose(int32_t n_params, const double *rot_sigma, const double *trans_sigma, const double *theta,
                                          uint64_t seed, uint64_t step, double *theta_out, int32_t *leaf_out, char *name_out,
                                          int32_t name_len);
/* java.lang.Double.toString, as the proposal names interpolate their parameters ("RotationYaw-0.01", "RandomShape-1.0E-4") */
ICP_API int icp_host_scala_double(double x, char *out, int32_t out_len);
ICP_API void icp_host_chain_destroy(icp_host_chain *chain);
ICP_API const char *icp_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
