// icp_abi.hip — host side of libicp_proposal_amd.so: the C ABI of include/icp_proposal.h.
//
// Owns device memory, the per-context HIP stream, the state / posterior / likelihood caches that stand in
// for the reference's Memoize wrappers (NonRigidIcpProposal.scala:49, evaluators/EvaluationCaching.scala:32),
// and the order in which kernels are enqueued.  Every entry point enqueues its whole kernel sequence on the
// context stream and synchronises ONCE, when results are copied back.  No CPU fallback exists: without a HIP
// device icp_ctx_create fails.
//
// ONE translation unit, split by concern (round 5; it was a single 5,200-line file): the parts below are included in this order.
#include "abi_types.inl"  // common types: errors, device buffers, shared model / target data, state slots, icp_ctx
#include "abi_state.inl"  // icp_ctx: state slots (instances) and the search prefixes shared by proposals and evaluators
#include "abi_pools.inl"  // pools and caches: streams, pinned blocks, device buffers of destroyed objects; live-context and eigen-stream registries
#include "abi_posterior.inl"  // posterior memo entries, icp_proposal / icp_evaluator and their methods (NonRigidIcpProposal.scala:88-153, the evaluators)
#include "abi_hints.inl"  // search hints handed from the first evaluation of a (model, target) pair to the contexts and evaluators that follow
#include "abi_context.inl"  // C ABI: contexts — create / destroy / set_target / set_rotation, counters, profiling, geometry entry points
#include "abi_methods.inl"  // C ABI: per-method entry points — proposals, evaluators, deterministic fit, variability maps, metrics, icp_chain_eval_step
#include "abi_step.inl"  // the merged step (five launches): fronts, speculative decompositions, icp_chain_step / _prelaunch
#include "abi_wide.inl"  // the wide step's host side (kernels_wide.hip)
#include "abi_batched.inl"  // C ABI: icp_chain_step_batched_issue / _collect / _abandon
#include "abi_device_loop.inl"  // C ABI: icp_chains_run_on_device (the whole MH loop on the device) and the remaining queries
