/*
 * icp_jni.c — JNI shim between the reference's Scala code and libicp_proposal_amd.so (include/icp_proposal.h).
 *
 * The reference (unibas-gravis/icp-proposal) is pure Scala on Scalismo and has no FFI of its own; this file and
 * the Scala sources under bindings/scala/api/gpu/ are what a maintainer adds to bind the MI355X path into the existing chain.
 * Natives of `object api.gpu.NativeIcp`.  Arrays are pinned with GetPrimitiveArrayCritical for the duration of one call — the
 * library copies what it keeps and never calls back into the JVM.  A non-zero icp_status becomes a RuntimeException carrying
 * icp_last_error(); -inf is a VALID return of logTransition (NonRigidIcpProposal.scala:72-74).
 *
 * Build (where a JDK is present):  cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *                                     icp_jni.c -L../../icp-proposal_amd -licp_proposal_amd -o libicp_jni.so
 * The build image of this repository has no JDK: the body is compiled only where <jni.h> exists (`make -C bindings/jni`
 * reports which case applies); the C ABI itself is exercised through ctypes by tests/.
 */
#if defined(__has_include)
#if __has_include(<jni.h>)
#define ICP_HAVE_JNI 1
#endif
#endif

#ifdef ICP_HAVE_JNI
#include <jni.h>
#include <stdint.h>
#include "icp_proposal.h"

static void throw_status(JNIEnv *env, int st) {
  if (st == ICP_OK) return;
  if (st == ICP_ERR_EMPTY) {
    /* no point survived the boundary filter: the reference's evaluator takes `.max` of an empty list there
     * (evaluators/CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator.scala:49-51, :61-63) and throws — so does the binding */
    jclass ecls = (*env)->FindClass(env, "java/lang/UnsupportedOperationException");
    if (ecls) (*env)->ThrowNew(env, ecls, "empty.max");
    return;
  }
  jclass cls = (*env)->FindClass(env, "java/lang/RuntimeException");
  if (cls) (*env)->ThrowNew(env, cls, icp_last_error());
}
/* Get*ArrayElements returns NULL when the JVM is out of memory (an OutOfMemoryError is then pending): never dereferenced */
static int oom(JNIEnv *env, const void *p) {
  (void)env;
  return p == 0;
}
#define PTR(T, h) ((T *)(intptr_t)(h))

JNIEXPORT jlong JNICALL Java_api_gpu_NativeIcp_00024_ctxCreate(JNIEnv *env, jobject self, jint n, jint t, jint r, jdoubleArray ref,
                                                               jdoubleArray mean, jdoubleArray basis, jdoubleArray variance, jintArray tris,
                                                               jint m, jint tt, jdoubleArray tpts, jintArray ttris, jint device) {
  (void)self;
  jdouble *a_ref = (*env)->GetDoubleArrayElements(env, ref, 0), *a_mean = mean ? (*env)->GetDoubleArrayElements(env, mean, 0) : 0;
  jdouble *a_basis = (*env)->GetDoubleArrayElements(env, basis, 0), *a_var = (*env)->GetDoubleArrayElements(env, variance, 0);
  jint *a_tris = (*env)->GetIntArrayElements(env, tris, 0), *a_ttris = (*env)->GetIntArrayElements(env, ttris, 0);
  jdouble *a_tpts = (*env)->GetDoubleArrayElements(env, tpts, 0);
  icp_ctx *ctx = 0;
  int st = ICP_OK;
  const int failed = oom(env, a_ref) || (mean && oom(env, a_mean)) || oom(env, a_basis) || oom(env, a_var) || oom(env, a_tris) ||
                     oom(env, a_ttris) || oom(env, a_tpts);
  if (!failed) {
    icp_model_desc md = {n, t, r, a_ref, a_mean, a_basis, a_var, (const int32_t *)a_tris};
    icp_mesh_desc td = {m, tt, a_tpts, (const int32_t *)a_ttris};
    st = icp_ctx_create(&md, &td, device, &ctx); /* copies everything to HBM; keeps no JVM pointer */
  }
  if (a_ref) (*env)->ReleaseDoubleArrayElements(env, ref, a_ref, JNI_ABORT);
  if (mean && a_mean) (*env)->ReleaseDoubleArrayElements(env, mean, a_mean, JNI_ABORT);
  if (a_basis) (*env)->ReleaseDoubleArrayElements(env, basis, a_basis, JNI_ABORT);
  if (a_var) (*env)->ReleaseDoubleArrayElements(env, variance, a_var, JNI_ABORT);
  if (a_tris) (*env)->ReleaseIntArrayElements(env, tris, a_tris, JNI_ABORT);
  if (a_tpts) (*env)->ReleaseDoubleArrayElements(env, tpts, a_tpts, JNI_ABORT);
  if (a_ttris) (*env)->ReleaseIntArrayElements(env, ttris, a_ttris, JNI_ABORT);
  if (failed) return 0; /* (the pending OutOfMemoryError surfaces in the caller) */
  throw_status(env, st);
  return (jlong)(intptr_t)ctx;
}

JNIEXPORT void JNICALL Java_api_gpu_NativeIcp_00024_ctxDestroy(JNIEnv *env, jobject self, jlong ctx) {
  (void)env; (void)self;
  icp_ctx_destroy(PTR(icp_ctx, ctx));
}

/* Scalismo's own Rotation(phi, theta, psi, centre) matrix for a theta's Euler angles (ModelFittingParameters.scala:79-86) */
JNIEXPORT void JNICALL Java_api_gpu_NativeIcp_00024_setRotation(JNIEnv *env, jobject self, jlong ctx, jdoubleArray angles, jdoubleArray rot) {
  (void)self;
  jdouble *a = (*env)->GetDoubleArrayElements(env, angles, 0), *r = rot ? (*env)->GetDoubleArrayElements(env, rot, 0) : 0;
  const int failed = oom(env, a) || (rot && oom(env, r));
  int st = failed ? ICP_OK : icp_ctx_set_rotation(PTR(icp_ctx, ctx), a, r);
  if (a) (*env)->ReleaseDoubleArrayElements(env, angles, a, JNI_ABORT);
  if (rot && r) (*env)->ReleaseDoubleArrayElements(env, rot, r, JNI_ABORT);
  if (failed) return;
  throw_status(env, st);
}

/* icp_ctx_rotation_convention: [verified, mismatched] — did every matrix registered so far agree with the library's Rz·Ry·Rx? */
JNIEXPORT jlongArray JNICALL Java_api_gpu_NativeIcp_00024_rotationConvention(JNIEnv *env, jobject self, jlong ctx) {
  (void)self;
  int64_t v[2] = {0, 0};
  int st = icp_ctx_rotation_convention(PTR(icp_ctx, ctx), &v[0], &v[1]);
  if (st != ICP_OK) { throw_status(env, st); return 0; }
  jlongArray out = (*env)->NewLongArray(env, 2);
  if (!out) return 0;
  const jlong jv[2] = {(jlong)v[0], (jlong)v[1]};
  (*env)->SetLongArrayRegion(env, out, 0, 2, jv);
  return out;
}

JNIEXPORT jlong JNICALL Java_api_gpu_NativeIcp_00024_proposalCreate(JNIEnv *env, jobject self, jlong ctx, jdouble step, jdouble sigma_t,
                                                                    jdouble sigma_n, jint direction, jboolean boundary_aware,
                                                                    jint n_model_ids, jdoubleArray target_pts) {
  (void)self;
  jsize nt = target_pts ? (*env)->GetArrayLength(env, target_pts) / 3 : 0;
  jdouble *tp = target_pts ? (*env)->GetDoubleArrayElements(env, target_pts, 0) : 0;
  if (target_pts && oom(env, tp)) return 0;
  icp_proposal_params prm = {step, sigma_t, sigma_n, direction, boundary_aware ? 1 : 0, n_model_ids, (int32_t)nt, tp};
  icp_proposal *p = 0;
  int st = icp_proposal_create(PTR(icp_ctx, ctx), &prm, &p);
  if (target_pts) (*env)->ReleaseDoubleArrayElements(env, target_pts, tp, JNI_ABORT);
  throw_status(env, st);
  return (jlong)(intptr_t)p;
}
JNIEXPORT void JNICALL Java_api_gpu_NativeIcp_00024_proposalDestroy(JNIEnv *env, jobject self, jlong p) {
  (void)env; (void)self;
  icp_proposal_destroy(PTR(icp_proposal, p));
}

JNIEXPORT jlong JNICALL Java_api_gpu_NativeIcp_00024_evaluatorCreate(JNIEnv *env, jobject self, jlong ctx, jint kind, jint mode, jint n_model_ids,
                                                                     jdoubleArray target_pts, jdouble gauss_mean, jdouble gauss_sigma,
                                                                     jdouble exp_rate) {
  (void)self;
  jsize nt = target_pts ? (*env)->GetArrayLength(env, target_pts) / 3 : 0;
  jdouble *tp = target_pts ? (*env)->GetDoubleArrayElements(env, target_pts, 0) : 0;
  icp_evaluator_params prm = {kind, mode, n_model_ids, (int32_t)nt, tp, gauss_mean, gauss_sigma, exp_rate};
  icp_evaluator *e = 0;
  int st = icp_evaluator_create(PTR(icp_ctx, ctx), &prm, &e);
  if (target_pts) (*env)->ReleaseDoubleArrayElements(env, target_pts, tp, JNI_ABORT);
  throw_status(env, st);
  return (jlong)(intptr_t)e;
}
JNIEXPORT void JNICALL Java_api_gpu_NativeIcp_00024_evaluatorDestroy(JNIEnv *env, jobject self, jlong e) {
  (void)env; (void)self;
  icp_evaluator_destroy(PTR(icp_evaluator, e));
}

JNIEXPORT void JNICALL Java_api_gpu_NativeIcp_00024_propose(JNIEnv *env, jobject self, jlong prop, jdoubleArray theta, jdoubleArray z,
                                                            jdoubleArray out) {
  (void)self;
  double *th = (*env)->GetPrimitiveArrayCritical(env, theta, 0), *zz = (*env)->GetPrimitiveArrayCritical(env, z, 0);
  double *o = (*env)->GetPrimitiveArrayCritical(env, out, 0);
  int st = icp_proposal_propose(PTR(icp_proposal, prop), th, zz, o, 0);
  (*env)->ReleasePrimitiveArrayCritical(env, out, o, 0);
  (*env)->ReleasePrimitiveArrayCritical(env, z, zz, JNI_ABORT);
  (*env)->ReleasePrimitiveArrayCritical(env, theta, th, JNI_ABORT);
  throw_status(env, st);
}

JNIEXPORT jdouble JNICALL Java_api_gpu_NativeIcp_00024_logTransition(JNIEnv *env, jobject self, jlong prop, jdoubleArray from, jdoubleArray to) {
  (void)self;
  double v = 0, *a = (*env)->GetPrimitiveArrayCritical(env, from, 0), *b = (*env)->GetPrimitiveArrayCritical(env, to, 0);
  int st = icp_proposal_log_transition(PTR(icp_proposal, prop), a, b, &v); /* -inf is a valid value */
  (*env)->ReleasePrimitiveArrayCritical(env, to, b, JNI_ABORT);
  (*env)->ReleasePrimitiveArrayCritical(env, from, a, JNI_ABORT);
  throw_status(env, st);
  return v;
}

JNIEXPORT jdouble JNICALL Java_api_gpu_NativeIcp_00024_logValue(JNIEnv *env, jobject self, jlong ev, jdoubleArray theta) {
  (void)self;
  double v = 0, *th = (*env)->GetPrimitiveArrayCritical(env, theta, 0);
  int st = icp_evaluator_log_value(PTR(icp_evaluator, ev), th, &v, 0);
  (*env)->ReleasePrimitiveArrayCritical(env, theta, th, JNI_ABORT);
  throw_status(env, st);
  return v;
}

/* optional accelerator: the whole MH step in one submission.  Returns the likelihood of the proposal. */
JNIEXPORT jdouble JNICALL Java_api_gpu_NativeIcp_00024_chainStep(JNIEnv *env, jobject self, jlong ev, jlongArray props, jint generator,
                                                                 jdoubleArray theta_cur, jdoubleArray z, jdoubleArray theta_prop,
                                                                 jdoubleArray fwd, jdoubleArray bwd) {
  (void)self;
  jsize n = (*env)->GetArrayLength(env, props);
  jlong *ph = (*env)->GetLongArrayElements(env, props, 0);
  icp_proposal *pp[8];
  for (jsize i = 0; i < n && i < 8; ++i) pp[i] = PTR(icp_proposal, ph[i]);
  (*env)->ReleaseLongArrayElements(env, props, ph, JNI_ABORT);
  double v = 0;
  double *cur = (*env)->GetPrimitiveArrayCritical(env, theta_cur, 0), *zz = z ? (*env)->GetPrimitiveArrayCritical(env, z, 0) : 0;
  double *prp = (*env)->GetPrimitiveArrayCritical(env, theta_prop, 0);
  double *f = (*env)->GetPrimitiveArrayCritical(env, fwd, 0), *b = (*env)->GetPrimitiveArrayCritical(env, bwd, 0);
  int st = icp_chain_step(PTR(icp_evaluator, ev), (int32_t)(n < 8 ? n : 8), pp, generator, cur, zz, prp, &v, f, b);
  (*env)->ReleasePrimitiveArrayCritical(env, bwd, b, 0);
  (*env)->ReleasePrimitiveArrayCritical(env, fwd, f, 0);
  (*env)->ReleasePrimitiveArrayCritical(env, theta_prop, prp, 0);
  if (z) (*env)->ReleasePrimitiveArrayCritical(env, z, zz, JNI_ABORT);
  (*env)->ReleasePrimitiveArrayCritical(env, theta_cur, cur, JNI_ABORT);
  throw_status(env, st);
  return v;
}
#else
/* no <jni.h> in this build environment: nothing to compile (see the header comment) */
typedef int icp_jni_not_built_here;
#endif
