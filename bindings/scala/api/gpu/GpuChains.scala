/*
 * GpuChains.scala — what sits between the adapters and the reference's chain code (api/sampling/SamplingRegistration.scala:45-93):
 *   bind          Scalismo's MetropolisHastings, unchanged, at one device submission per step           (icp_chain_bind)
 *   runOnDevice   the whole loop of B chains on the device, records in JSONAcceptRejectLogger's layout  (icp_chains_run_on_device)
 * Reference-side binding (INTEGRATION.md); not compiled in this repository.
 */
package api.gpu

import api.sampling.{ModelFittingParameters, PoseParameters, ScaleParameter, ShapeParameters}
import breeze.linalg.DenseVector
import scalismo.geometry.{EuclideanVector3D, Point3D}

/** The fields of icp_mh_mixture (include/icp_proposal.h): the ICP mixture of MixedProposalDistributions.mixedProposalICP (:48-68) with
  * the shape walk of mixedRandomShapeProposal (:41-46) and, wPose > 0, the six pose walks of mixedRandomPoseProposal (:29-39) in the outer
  * mixture of apps/femur/IcpProposalRegistration.scala:70-72 / apps/bfm/BfmFittingPartial.scala:70. */
case class GpuMixture(icpWeight: (Double, Double) = (0.5, 0.5), wIcp: Double = 0.9, wRw: Double = 0.1, rwSigma: Double = 0.1,
                      wPose: Double = 0.0, poseRotSigma: (Double, Double, Double) = (0.01, 0.01, 0.01),
                      poseTransSigma: (Double, Double, Double) = (0.1, 0.1, 0.1)) {
  def toArray: Array[Double] = Array(icpWeight._1, icpWeight._2, wIcp, wRw, rwSigma, wPose, poseRotSigma._1, poseRotSigma._2,
    poseRotSigma._3, poseTransSigma._1, poseTransSigma._2, poseTransSigma._3)
}

/** One chain's native objects: its context's likelihood evaluator and ICP proposals in the mixture's order. */
case class GpuChain(evaluator: GpuLikelihoodEvaluator, proposals: Seq[GpuNonRigidIcpProposal])

/** A step record of runOnDevice: what JSONAcceptRejectLogger writes per step (loggers/JSONAcceptRejectLogger.scala:35, :93-106). */
case class GpuStepRecord(index: Long, accepted: Boolean, leaf: Int, logValue: Double, theta: Array[Double])

object GpuChains {
  /** allParameters → ModelFittingParameters: [s | t(3) | rotation._1, _2, _3 | centre(3) | c(r)] (ModelFittingParameters.scala:28-36, :64) */
  def fromVector(v: Array[Double], generatedBy: String): ModelFittingParameters =
    ModelFittingParameters(ScaleParameter(v(0)), PoseParameters(EuclideanVector3D(v(1), v(2), v(3)), (v(4), v(5), v(6)), Point3D(v(7), v(8), v(9))),
      ShapeParameters(DenseVector(v.drop(10))), generatedBy)

  /** The drop-in path.  After this, `MetropolisHastings(generator, evaluator)` built from the adapters as
    * SamplingRegistration.scala:52-58 builds it runs UNCHANGED: propose() of a bound proposal (or logValue() of a state made by a
    * random-walk / pose proposal on the JVM) submits the whole step — instance, searches, every ICP proposal's posterior at the proposed
    * state, the likelihood, the transition densities both ways — and the five per-method calls behind it are answered on the host. */
  def bind(chain: GpuChain): Unit = NativeIcp.chainBind(chain.evaluator.handle, chain.proposals.map(_.handle).toArray)
  def unbind(chain: GpuChain): Unit = NativeIcp.chainBind(chain.evaluator.handle, Array.empty[Long])

  /** The replacement of `(0 until n).par.foreach { i => … runfitting … }` (apps/femur/RunMHRandomInitComparison.scala:66-87) and of the
    * chains of one target in apps/femur/StdIcpVsChainICPrandomInitComparisonAll.scala:113-160: numOfSamples steps of every chain (a
    * GpuContext each, one model) inside ONE native call.  Random numbers: the library's counter-based generator, stream (seed, step, lane)
    * — the reference's chains are not reproducible run to run either (unseeded proposal RNG, IcpProposalRegistration.scala:33).
    * initial(b) = chain b's start and its product log value (prior + likelihood); returns the final states and the records. */
  def runOnDevice(chains: Seq[GpuChain], mixture: GpuMixture, seeds: Seq[Long], initial: Seq[(ModelFittingParameters, Double)],
                  numOfSamples: Int, firstStep: Long = 0L, wantRecords: Boolean = true): (Seq[ModelFittingParameters], Seq[Seq[GpuStepRecord]]) = {
    val b = chains.size
    val nProps = chains.head.proposals.size
    val p = 10 + chains.head.evaluator.ctx.rank
    require(chains.forall(_.proposals.size == nProps) && seeds.size == b && initial.size == b)
    if (mixture.wPose > 0) require(chains.forall(_.evaluator.ctx.rotationConventionVerified), "Scalismo's rotation convention is not the library's")
    val theta = initial.flatMap(_._1.allParameters.toArray).toArray
    val logValue = initial.map(_._2).toArray
    val records = if (wantRecords) new Array[Double](b * numOfSamples * (4 + p)) else null
    val accepted = new Array[Long](b)
    NativeIcp.chainsRunOnDevice(chains.map(_.evaluator.handle).toArray, nProps, chains.flatMap(_.proposals.map(_.handle)).toArray,
      mixture.toArray, seeds.toArray, Array.fill(b)(firstStep), theta, logValue, numOfSamples, records, accepted)
    val states = (0 until b).map(i => fromVector(theta.slice(i * p, (i + 1) * p), "GpuChains.runOnDevice"))
    val recs = if (!wantRecords) Seq.fill(b)(Seq.empty[GpuStepRecord]) else (0 until b).map { i =>
      (0 until numOfSamples).map { s =>
        val o = (i * numOfSamples + s) * (4 + p)
        GpuStepRecord(records(o).toLong, records(o + 1) != 0.0, records(o + 2).toInt, records(o + 3), records.slice(o + 4, o + 4 + p))
      }
    }
    (states, recs)
  }
}
