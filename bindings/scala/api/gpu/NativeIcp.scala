/*
 * NativeIcp.scala — the natives of libicp_jni.so (bindings/jni/icp_jni.c) over libicp_proposal_amd.so (include/icp_proposal.h).
 * Part of the reference-side binding a maintainer adds to unibas-gravis/icp-proposal (INTEGRATION.md); it is not compiled in this
 * repository's build image (no JVM, no Scalismo), the C ABI underneath is what tests/ exercise.
 */
package api.gpu

object NativeIcp {
  System.loadLibrary("icp_jni")

  /** StatisticalMeshModel + target TriangleMesh3D onto one MI355X.  Arrays exactly as Scalismo holds them: reference points and
    * cells, mean deformation (gp.meanVector − reference), UNSCALED basis (row-major 3N×r), variances; device = HIP ordinal, −1 = LOCAL_RANK. */
  @native def ctxCreate(n: Int, t: Int, r: Int, ref: Array[Double], mean: Array[Double], basis: Array[Double], variance: Array[Double],
                        tris: Array[Int], m: Int, tt: Int, targetPoints: Array[Double], targetTris: Array[Int], device: Int): Long
  @native def ctxDestroy(ctx: Long): Unit
  /** Scalismo's own rotation matrix (row-major 3×3) for a triple of Euler angles: keeps Rotation(phi, theta, psi, center)'s convention
    * on the Scala side (ModelFittingParameters.scala:79-86).  rot == null withdraws the entry. */
  @native def setRotation(ctx: Long, angles: Array[Double], rot: Array[Double]): Unit
  /** (verified, mismatched): how many of the matrices registered so far agreed, to rounding, with the native side's own Rz·Ry·Rx.
    * mismatched == 0 after the first few steps of a chain says Scalismo's Rotation(phi, theta, psi, center) IS that convention, and the
    * whole chain loop — pose walks included — may then run on the device (icp_chains_run_on_device). */
  @native def rotationConvention(ctx: Long): Array[Long]

  @native def proposalCreate(ctx: Long, step: Double, sigmaT: Double, sigmaN: Double, direction: Int, boundaryAware: Boolean,
                             nModelIds: Int, targetPts: Array[Double]): Long
  @native def proposalDestroy(prop: Long): Unit
  @native def evaluatorCreate(ctx: Long, kind: Int, mode: Int, nModelIds: Int, targetPts: Array[Double], gaussMean: Double,
                              gaussSigma: Double, expRate: Double): Long
  @native def evaluatorDestroy(ev: Long): Unit

  @native def propose(prop: Long, theta: Array[Double], z: Array[Double], out: Array[Double]): Unit
  @native def logTransition(prop: Long, from: Array[Double], to: Array[Double]): Double
  @native def logValue(ev: Long, theta: Array[Double]): Double

  /** Optional accelerator (INTEGRATION.md §3): one MH step in one submission.  Returns the likelihood of the proposal; thetaProp is
    * written when generator >= 0 and read when generator < 0; fwd / bwd receive the transition log-densities of every proposal. */
  @native def chainStep(ev: Long, props: Array[Long], generator: Int, thetaCur: Array[Double], z: Array[Double], thetaProp: Array[Double],
                        fwd: Array[Double], bwd: Array[Double]): Double
}
