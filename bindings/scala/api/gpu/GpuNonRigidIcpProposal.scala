/*
 * GpuNonRigidIcpProposal.scala — drop-in for api.sampling.proposals.NonRigidIcpProposal (same constructor arguments,
 * NonRigidIcpProposal.scala:30-41) on the MI355X path.  Construction sites to switch: MixedProposalDistributions.mixedProposalICP
 * (api/sampling/MixedProposalDistributions.scala:48-68).  Reference-side binding (INTEGRATION.md); not compiled in this repository.
 */
package api.gpu

import api.other.{IcpProjectionDirection, ModelSampling, TargetSampling}
import api.sampling.{ModelFittingParameters, ShapeParameters}
import breeze.linalg.DenseVector
import scalismo.mesh.TriangleMesh3D
import scalismo.sampling.{ProposalGenerator, TransitionProbability}
import scalismo.statisticalmodel.StatisticalMeshModel

case class GpuNonRigidIcpProposal(ctx: Long, model: StatisticalMeshModel, target: TriangleMesh3D, stepLength: Double,
                                  tangentialNoise: Double, noiseAlongNormal: Double, numOfSamplePoints: Int,
                                  projectionDirection: IcpProjectionDirection = ModelSampling, boundaryAware: Boolean = true,
                                  generatedBy: String = "ShapeIcpProposal")(implicit rand: scalismo.utils.Random)
  extends ProposalGenerator[ModelFittingParameters] with TransitionProbability[ModelFittingParameters] {

  // the two decimations of NonRigidIcpProposal.scala:45-46 stay here; only their outcome crosses the boundary
  private val nModelIds = model.decimate(numOfSamplePoints).referenceMesh.pointSet.numberOfPoints
  private val targetPts = target.operations.decimate(numOfSamplePoints).pointSet.points.flatMap(_.toArray).toArray
  private val handle = NativeIcp.proposalCreate(ctx, stepLength, tangentialNoise, noiseAlongNormal,
    if (projectionDirection == TargetSampling) 1 else 0, boundaryAware, nModelIds, targetPts)

  /** Scalismo's matrix for theta's Euler angles, so that the native side poses with Scalismo's convention (icp_ctx_set_rotation). */
  private def registerRotation(theta: ModelFittingParameters): Unit = {
    val e = theta.poseParameters.rotation
    val m = theta.poseTransform.rotation.rotationMatrix // scalismo.registration.RotationTransform[_3D]
    NativeIcp.setRotation(ctx, Array(e._1, e._2, e._3), Array(m(0, 0), m(0, 1), m(0, 2), m(1, 0), m(1, 1), m(1, 2), m(2, 0), m(2, 1), m(2, 2)))
  }

  override def propose(theta: ModelFittingParameters): ModelFittingParameters = {
    registerRotation(theta)
    val z = Array.fill(model.rank)(rand.scalaRandom.nextGaussian()) // what posterior.sample() draws (:55)
    val out = new Array[Double](10 + model.rank)
    NativeIcp.propose(handle, theta.allParameters.toArray, z, out)
    theta.copy(shapeParameters = ShapeParameters(DenseVector(out.drop(10))), generatedBy = generatedBy) // :64-67
  }

  override def logTransitionProbability(from: ModelFittingParameters, to: ModelFittingParameters): Double = {
    registerRotation(from)
    NativeIcp.logTransition(handle, from.allParameters.toArray, to.allParameters.toArray) // -inf unless only the shape differs (:72-74)
  }
}
