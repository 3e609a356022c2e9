/*
 * GpuLikelihoodEvaluator.scala — drop-in for IndependentPointDistanceEvaluator / HausdorffDistanceEvaluator /
 * CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator (api/sampling/evaluators/) on the MI355X path.  Construction sites to
 * switch: api/sampling/ProductEvaluators.scala:38-94.  Reference-side binding (INTEGRATION.md); not compiled in this repository.
 */
package api.gpu

import api.sampling.ModelFittingParameters
import api.sampling.evaluators.{EvaluationMode, ModelToTargetEvaluation, SymmetricEvaluation, TargetToModelEvaluation}
import scalismo.sampling.DistributionEvaluator

object GpuLikelihoodEvaluator {
  private def modeId(m: EvaluationMode): Int = m match {
    case ModelToTargetEvaluation => 0
    case TargetToModelEvaluation => 1
    case SymmetricEvaluation     => 2
  }

  /** IndependentPointDistanceEvaluator(model, target, Gaussian(mean, sigma), mode, numberOfPointsForComparison) (:27-31) */
  def independent(ctx: GpuContext, mean: Double, sigma: Double, mode: EvaluationMode, numberOfPointsForComparison: Int): GpuLikelihoodEvaluator = {
    val nIds = ctx.model.decimate(numberOfPointsForComparison).referenceMesh.pointSet.numberOfPoints                  // :34
    val tp = ctx.target.operations.decimate(numberOfPointsForComparison).pointSet.points.flatMap(_.toArray).toArray    // :35
    new GpuLikelihoodEvaluator(ctx, NativeIcp.evaluatorCreate(ctx.handle, 0, modeId(mode), nIds, tp, mean, sigma, 1.0))
  }

  /** HausdorffDistanceEvaluator(model, target, Exponential(rate)) (HausdorffDistanceEvaluator.scala:25-28) */
  def hausdorff(ctx: GpuContext, rate: Double): GpuLikelihoodEvaluator =
    new GpuLikelihoodEvaluator(ctx, NativeIcp.evaluatorCreate(ctx.handle, 1, 2, 0, null, 0.0, 1.0, rate))

  /** CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator (…BoundaryAwareEvaluator.scala:27-32) */
  def collective(ctx: GpuContext, avgMean: Double, avgSigma: Double, maxRate: Double, mode: EvaluationMode,
                 numberOfPointsForComparison: Int): GpuLikelihoodEvaluator = {
    val nIds = ctx.model.decimate(numberOfPointsForComparison).referenceMesh.pointSet.numberOfPoints
    val tp = ctx.target.operations.decimate(numberOfPointsForComparison).pointSet.points.flatMap(_.toArray).toArray
    new GpuLikelihoodEvaluator(ctx, NativeIcp.evaluatorCreate(ctx.handle, 2, modeId(mode), nIds, tp, avgMean, avgSigma, maxRate))
  }
}

class GpuLikelihoodEvaluator(val ctx: GpuContext, val handle: Long) extends DistributionEvaluator[ModelFittingParameters] with AutoCloseable {
  private var open = true
  ctx.adopt(this)

  // EVERY call goes to the native side, which keeps the Memoize(3) of evaluators/EvaluationCaching.scala:32 itself: a chain bound with
  // GpuChains.bind relies on seeing logValue(current) ahead of logValue(proposal) (MetropolisHastings.next's order)
  override def logValue(sample: ModelFittingParameters): Double = {
    ctx.registerRotation(sample)
    NativeIcp.logValue(handle, sample.allParameters.toArray)
  }

  override def close(): Unit = if (open) { open = false; NativeIcp.evaluatorDestroy(handle) }
}
