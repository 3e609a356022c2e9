/*
 * GpuNonRigidIcpProposal.scala — drop-in for api.sampling.proposals.NonRigidIcpProposal (same constructor arguments,
 * NonRigidIcpProposal.scala:30-41, with the GpuContext in place of model and target) on the MI355X path.  Construction sites to switch:
 * MixedProposalDistributions.mixedProposalICP (api/sampling/MixedProposalDistributions.scala:48-68).  Reference-side binding
 * (INTEGRATION.md); not compiled in this repository.
 */
package api.gpu

import api.other.{IcpProjectionDirection, ModelSampling, TargetSampling}
import api.sampling.{ModelFittingParameters, ShapeParameters}
import breeze.linalg.DenseVector
import scalismo.sampling.{ProposalGenerator, TransitionProbability}

case class GpuNonRigidIcpProposal(ctx: GpuContext, stepLength: Double, tangentialNoise: Double, noiseAlongNormal: Double,
                                  numOfSamplePoints: Int, projectionDirection: IcpProjectionDirection = ModelSampling,
                                  boundaryAware: Boolean = true, generatedBy: String = "ShapeIcpProposal")(implicit rand: scalismo.utils.Random)
  extends ProposalGenerator[ModelFittingParameters] with TransitionProbability[ModelFittingParameters] with AutoCloseable {

  // the two decimations of NonRigidIcpProposal.scala:45-46 stay here; only their outcome crosses the boundary
  private val nModelIds = ctx.model.decimate(numOfSamplePoints).referenceMesh.pointSet.numberOfPoints
  private val targetPts = ctx.target.operations.decimate(numOfSamplePoints).pointSet.points.flatMap(_.toArray).toArray
  val handle: Long = NativeIcp.proposalCreate(ctx.handle, stepLength, tangentialNoise, noiseAlongNormal,
    if (projectionDirection == TargetSampling) 1 else 0, boundaryAware, nModelIds, targetPts)
  private var open = true
  ctx.adopt(this)

  override def propose(theta: ModelFittingParameters): ModelFittingParameters = {
    ctx.registerRotation(theta)
    val z = Array.fill(ctx.rank)(rand.scalaRandom.nextGaussian()) // what posterior.sample() draws (:55)
    val out = new Array[Double](10 + ctx.rank)
    NativeIcp.propose(handle, theta.allParameters.toArray, z, out)
    theta.copy(shapeParameters = ShapeParameters(DenseVector(out.drop(10))), generatedBy = generatedBy) // :64-67
  }

  override def logTransitionProbability(from: ModelFittingParameters, to: ModelFittingParameters): Double = {
    ctx.registerRotation(from)
    NativeIcp.logTransition(handle, from.allParameters.toArray, to.allParameters.toArray) // -inf unless only the shape differs (:72-74)
  }

  override def close(): Unit = if (open) { open = false; NativeIcp.proposalDestroy(handle) }
}
