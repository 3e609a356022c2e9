/*
 * NativeIcp.scala — the natives of libicp_jni.so (bindings/jni/icp_jni.c) over libicp_proposal_amd.so (include/icp_proposal.h).
 * Part of the reference-side binding a maintainer adds to unibas-gravis/icp-proposal (INTEGRATION.md); it is not compiled in this
 * repository's build image (no JVM, no Scalismo).  tests/test_bindings_cpu.py checks every declaration here against a
 * Java_api_gpu_NativeIcp_00024_* definition in icp_jni.c (name and argument count) and every icp_* call there against the header;
 * tests/test_gpu_jni.py runs the natives over a JNI test double.
 *
 * Handles are Longs (native pointers).  "flat" arrays hold B chains one after the other: theta [B*(10+r)], z [B*r], fwd / bwd [B*nProps].
 */
package api.gpu

object NativeIcp {
  System.loadLibrary("icp_jni")

  // ---------------------------------------------------------------- contexts
  /** StatisticalMeshModel + target TriangleMesh3D onto one MI355X.  Arrays exactly as Scalismo holds them: reference points and
    * cells, mean deformation (gp.meanVector − reference), UNSCALED basis (row-major 3N×r), variances; device = HIP ordinal, −1 = LOCAL_RANK. */
  @native def ctxCreate(n: Int, t: Int, r: Int, ref: Array[Double], mean: Array[Double], basis: Array[Double], variance: Array[Double],
                        tris: Array[Int], m: Int, tt: Int, targetPoints: Array[Double], targetTris: Array[Int], device: Int): Long
  /** icp_ctx_create_keyed: many contexts of ONE model (a context per chain thread / work item).  modelKey != 0 vouches that equal keys
    * mean equal model arrays: the contexts share the model's device data without the library hashing 137 MB of basis per context. */
  @native def ctxCreateKeyed(n: Int, t: Int, r: Int, ref: Array[Double], mean: Array[Double], basis: Array[Double], variance: Array[Double],
                             tris: Array[Int], m: Int, tt: Int, targetPoints: Array[Double], targetTris: Array[Int], device: Int,
                             modelKey: Long): Long
  @native def ctxDestroy(ctx: Long): Unit
  /** icp_ctx_set_target: the context goes on to another target (its proposals and evaluators destroyed first). */
  @native def ctxSetTarget(ctx: Long, m: Int, tt: Int, targetPoints: Array[Double], targetTris: Array[Int]): Unit
  @native def ctxRank(ctx: Long): Int
  /** Scalismo's own rotation matrix (row-major 3×3) for a triple of Euler angles: keeps Rotation(phi, theta, psi, center)'s convention
    * on the Scala side (ModelFittingParameters.scala:79-86).  rot == null withdraws the entry. */
  @native def setRotation(ctx: Long, angles: Array[Double], rot: Array[Double]): Unit
  /** (verified, mismatched): how many of the matrices registered so far agreed, to rounding, with the native side's own Rz·Ry·Rx.
    * mismatched == 0 after the first few steps of a chain says Scalismo's Rotation(phi, theta, psi, center) IS that convention, and the
    * whole chain loop — pose walks included — may then run on the device (chainsRunOnDevice). */
  @native def rotationConvention(ctx: Long): Array[Long]

  // ---------------------------------------------------------------- proposals and evaluators
  @native def proposalCreate(ctx: Long, step: Double, sigmaT: Double, sigmaN: Double, direction: Int, boundaryAware: Boolean,
                             nModelIds: Int, targetPts: Array[Double]): Long
  @native def proposalDestroy(prop: Long): Unit
  /** opt-in, NOT the reference's arithmetic: 0 = the KL basis posterior.sample() draws from, 1 = Cholesky root (same distribution). */
  @native def proposalSetSampler(prop: Long, sampler: Int): Unit
  @native def evaluatorCreate(ctx: Long, kind: Int, mode: Int, nModelIds: Int, targetPts: Array[Double], gaussMean: Double,
                              gaussSigma: Double, expRate: Double): Long
  @native def evaluatorDestroy(ev: Long): Unit

  // ---------------------------------------------------------------- the three plug-in methods
  @native def propose(prop: Long, theta: Array[Double], z: Array[Double], out: Array[Double]): Unit
  @native def logTransition(prop: Long, from: Array[Double], to: Array[Double]): Double
  @native def logValue(ev: Long, theta: Array[Double]): Double
  /** icp_chain_bind: ev and props (in the mixture's order) are ONE MetropolisHastings chain; from then on the first of a step's
    * per-method calls submits the whole step and the others find their values on the host.  Empty props unbinds. */
  @native def chainBind(ev: Long, props: Array[Long]): Unit
  /** (whole steps submitted by propose, by logValue, transition densities answered from a parked step) */
  @native def chainBindStats(ev: Long): Array[Long]

  // ---------------------------------------------------------------- whole steps
  /** icp_chain_step.  Returns the likelihood of the proposal; thetaProp is written when generator >= 0 and read when generator < 0;
    * fwd / bwd receive the transition log-densities of every proposal. */
  @native def chainStep(ev: Long, props: Array[Long], generator: Int, thetaCur: Array[Double], z: Array[Double], thetaProp: Array[Double],
                        fwd: Array[Double], bwd: Array[Double]): Double
  /** icp_chain_step_batched: B chains (a context each) in one submission; flat arrays; z may be null when no generator is >= 0.
    * Returns 0 or the first failing chain's status; per-chain codes in status. */
  @native def chainStepBatched(evs: Array[Long], nProps: Int, props: Array[Long], generator: Array[Int], thetaCur: Array[Double],
                               z: Array[Double], thetaProp: Array[Double], logValue: Array[Double], fwd: Array[Double], bwd: Array[Double],
                               status: Array[Int]): Int
  /** … in two halves (several groups of chains in flight): the ticket owns native copies of the inputs; launchCtx = 0: the first chain's. */
  @native def chainStepBatchedIssue(evs: Array[Long], nProps: Int, props: Array[Long], generator: Array[Int], thetaCur: Array[Double],
                                    z: Array[Double], thetaProp: Array[Double], launchCtx: Long): Long
  @native def chainStepBatchedCollect(ticket: Long, thetaProp: Array[Double], logValue: Array[Double], fwd: Array[Double], bwd: Array[Double],
                                      status: Array[Int]): Int
  @native def chainStepBatchedAbandon(ticket: Long): Unit
  /** icp_chains_run_on_device: nSteps steps of B chains without a host round trip per step.  mixture = GpuMixture.toArray (12 doubles);
    * theta (flat) and logValue in/out; records null or [B*nSteps*(4+10+r)]: rows [index, accepted, leaf id, log value, theta]. */
  @native def chainsRunOnDevice(evs: Array[Long], nProps: Int, props: Array[Long], mixture: Array[Double], seeds: Array[Long],
                                firstStep: Array[Long], theta: Array[Double], logValue: Array[Double], nSteps: Int, records: Array[Double],
                                accepted: Array[Long]): Unit

  // ---------------------------------------------------------------- queries
  /** 0 the five merged launches, 1 the wide step, 2 per-stage kernels. */
  @native def chainStepPath(ev: Long, props: Array[Long]): Int
  /** (merged, wide, per-stage, device loop) steps; ctx = 0: of the process. */
  @native def stepPaths(ctx: Long): Array[Long]
  /** (wait_timeouts, speculation_giveups, pipeline_fallbacks, step_redos, gate_timeouts): zeros in a healthy run. */
  @native def runtimeStats(ctx: Long): Array[Long]
  /** RegistrationComparison.evaluateReconstruction2GroundTruth[BoundaryAware]: (avg, hausdorff, boundary-aware avg, max, kept). */
  @native def meshMetrics(ctx: Long, theta: Array[Double]): Array[Double]
  @native def releaseCachedModels(): Unit
}
