/*
 * GpuContext.scala — one statistical mesh model + one target mesh resident on one MI355X (icp_ctx): what the adapters of this package
 * are constructed over.  A context holds ONE chain's scratch: an experiment that runs chains from a parallel collection
 * (apps/femur/RunMHRandomInitComparison.scala:59-66) makes one per chain thread; they share the model's and the target's device data
 * (icp_ctx_create_keyed), a further context costs 1-3 ms.  A batch registration (one model, many targets:
 * apps/femur/StdIcpVsChainICPrandomInitComparisonAll.scala:106-163) keeps its contexts and hands them from target to target
 * (setTarget).  Reference-side binding (INTEGRATION.md); not compiled in this repository.
 */
package api.gpu

import java.util.concurrent.atomic.AtomicLong

import api.sampling.ModelFittingParameters
import scalismo.mesh.TriangleMesh3D
import scalismo.statisticalmodel.StatisticalMeshModel

object GpuContext {
  private val nextKey = new AtomicLong(1)
  private val keys = new java.util.IdentityHashMap[StatisticalMeshModel, java.lang.Long]()

  /** One key per StatisticalMeshModel OBJECT (models are immutable in Scalismo): equal keys mean equal arrays, as the library requires. */
  def modelKey(model: StatisticalMeshModel): Long = keys.synchronized {
    Option(keys.get(model)).map(_.longValue).getOrElse { val k = nextKey.getAndIncrement(); keys.put(model, k); k }
  }

  private def meshArrays(mesh: TriangleMesh3D): (Array[Double], Array[Int]) =
    (mesh.pointSet.points.flatMap(_.toArray).toArray,
      mesh.triangulation.triangles.flatMap(t => Seq(t.ptId1.id, t.ptId2.id, t.ptId3.id)).toArray)
}

class GpuContext(val model: StatisticalMeshModel, initialTarget: TriangleMesh3D, device: Int = -1) extends AutoCloseable {
  private var currentTarget = initialTarget
  private val closeables = scala.collection.mutable.ArrayBuffer[AutoCloseable]()

  val handle: Long = {
    val ref = model.referenceMesh
    val (refPts, refTris) = GpuContext.meshArrays(ref)
    val mean = (model.gp.meanVector.toArray, refPts).zipped.map(_ - _) // Statismo stores the mean SHAPE; the boundary takes the deformation
    val basis = model.gp.basisMatrix.t.toArray                          // Breeze is column-major: the transpose's storage is row-major 3N×r
    val (tp, tt) = GpuContext.meshArrays(initialTarget)
    NativeIcp.ctxCreateKeyed(ref.pointSet.numberOfPoints, ref.triangulation.triangles.size, model.rank, refPts, mean, basis,
      model.gp.variance.toArray, refTris, initialTarget.pointSet.numberOfPoints, initialTarget.triangulation.triangles.size, tp, tt,
      device, GpuContext.modelKey(model))
  }

  def target: TriangleMesh3D = currentTarget
  def rank: Int = model.rank

  /** Adapters register themselves: setTarget / close destroy them first (icp_ctx_set_target is refused while any lives). */
  private[gpu] def adopt(c: AutoCloseable): Unit = closeables += c

  /** icp_ctx_set_target: everything made for the old target is closed; the model's device data, streams and buffers stay. */
  def setTarget(mesh: TriangleMesh3D): Unit = {
    closeables.reverseIterator.foreach(_.close()); closeables.clear()
    val (tp, tt) = GpuContext.meshArrays(mesh)
    NativeIcp.ctxSetTarget(handle, mesh.pointSet.numberOfPoints, mesh.triangulation.triangles.size, tp, tt)
    currentTarget = mesh
  }

  /** Scalismo's matrix for theta's Euler angles, so that the native side poses with Scalismo's convention (icp_ctx_set_rotation). */
  def registerRotation(theta: ModelFittingParameters): Unit = {
    val e = theta.poseParameters.rotation
    val m = ModelFittingParameters.poseTransform(theta).rotation.rotationMatrix // ModelFittingParameters.scala:79-86
    NativeIcp.setRotation(handle, Array(e._1, e._2, e._3), Array(m(0, 0), m(0, 1), m(0, 2), m(1, 0), m(1, 1), m(1, 2), m(2, 0), m(2, 1), m(2, 2)))
  }

  /** Did every registered matrix agree with the library's Rz·Ry·Rx?  Then pose walks may run inside GpuChains.runOnDevice. */
  def rotationConventionVerified: Boolean = { val v = NativeIcp.rotationConvention(handle); v(0) > 0 && v(1) == 0 }

  def stepPaths: Array[Long] = NativeIcp.stepPaths(handle)
  def runtimeStats: Array[Long] = NativeIcp.runtimeStats(handle)

  override def close(): Unit = {
    closeables.reverseIterator.foreach(_.close()); closeables.clear()
    NativeIcp.ctxDestroy(handle)
  }
}
