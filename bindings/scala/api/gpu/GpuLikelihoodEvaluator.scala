/*
 * GpuLikelihoodEvaluator.scala — drop-in for IndependentPointDistanceEvaluator / HausdorffDistanceEvaluator /
 * CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator (api/sampling/evaluators/) on the MI355X path.  Construction sites to
 * switch: api/sampling/ProductEvaluators.scala:38-94.  Reference-side binding (INTEGRATION.md); not compiled in this repository.
 */
package api.gpu

import api.sampling.ModelFittingParameters
import api.sampling.evaluators.{EvaluationMode, ModelToTargetEvaluation, SymmetricEvaluation, TargetToModelEvaluation}
import scalismo.mesh.TriangleMesh3D
import scalismo.sampling.DistributionEvaluator
import scalismo.statisticalmodel.StatisticalMeshModel

object GpuLikelihoodEvaluator {
  private def modeId(m: EvaluationMode): Int = m match {
    case ModelToTargetEvaluation => 0
    case TargetToModelEvaluation => 1
    case SymmetricEvaluation     => 2
  }

  /** IndependentPointDistanceEvaluator(model, target, Gaussian(mean, sigma), mode, numberOfPointsForComparison) (:27-31) */
  def independent(ctx: Long, model: StatisticalMeshModel, target: TriangleMesh3D, mean: Double, sigma: Double, mode: EvaluationMode,
                  numberOfPointsForComparison: Int): GpuLikelihoodEvaluator = {
    val nIds = model.decimate(numberOfPointsForComparison).referenceMesh.pointSet.numberOfPoints            // :34
    val tp = target.operations.decimate(numberOfPointsForComparison).pointSet.points.flatMap(_.toArray).toArray // :35
    GpuLikelihoodEvaluator(NativeIcp.evaluatorCreate(ctx, 0, modeId(mode), nIds, tp, mean, sigma, 1.0))
  }

  /** HausdorffDistanceEvaluator(model, target, Exponential(rate)) (HausdorffDistanceEvaluator.scala:25-28) */
  def hausdorff(ctx: Long, rate: Double): GpuLikelihoodEvaluator =
    GpuLikelihoodEvaluator(NativeIcp.evaluatorCreate(ctx, 1, 2, 0, null, 0.0, 1.0, rate))

  /** CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator (…BoundaryAwareEvaluator.scala:27-32) */
  def collective(ctx: Long, model: StatisticalMeshModel, target: TriangleMesh3D, avgMean: Double, avgSigma: Double, maxRate: Double,
                 mode: EvaluationMode, numberOfPointsForComparison: Int): GpuLikelihoodEvaluator = {
    val nIds = model.decimate(numberOfPointsForComparison).referenceMesh.pointSet.numberOfPoints
    val tp = target.operations.decimate(numberOfPointsForComparison).pointSet.points.flatMap(_.toArray).toArray
    GpuLikelihoodEvaluator(NativeIcp.evaluatorCreate(ctx, 2, modeId(mode), nIds, tp, avgMean, avgSigma, maxRate))
  }
}

case class GpuLikelihoodEvaluator(handle: Long) extends DistributionEvaluator[ModelFittingParameters] {
  // (the native side keeps the Memoize(3) of evaluators/EvaluationCaching.scala:32 itself)
  override def logValue(sample: ModelFittingParameters): Double = NativeIcp.logValue(handle, sample.allParameters.toArray)
}
