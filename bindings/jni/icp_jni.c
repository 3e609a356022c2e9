/*
 * icp_jni.c — JNI shim between the reference's Scala code and libicp_proposal_amd.so (include/icp_proposal.h).
 *
 * The reference (unibas-gravis/icp-proposal) is pure Scala on Scalismo and has no FFI of its own; this file and the Scala sources under
 * bindings/scala/api/gpu/ are what a maintainer adds to bind the MI355X path into the existing chain.  Natives of
 * `object api.gpu.NativeIcp` (a Scala object compiles to the class NativeIcp$: hence `_00024` in the symbol names).
 *
 * Arrays: every per-step native copies its (10 + rank)-double arguments into a buffer on the C stack with Get<T>ArrayRegion and its
 * results back with Set<T>ArrayRegion.  No GetPrimitiveArrayCritical anywhere: a critical region may not block, and every call here
 * waits 0.1-3 ms for the GPU — with the reference's `.par` chains (apps/femur/RunMHRandomInitComparison.scala:66) every other JVM
 * thread's garbage collection would wait with it.  The bulk arrays of context creation use Get<T>ArrayElements (may copy, never
 * blocks the collector).  The library copies what it keeps and never calls back into the JVM.
 * Errors: a non-zero icp_status becomes a RuntimeException carrying icp_last_error(); ICP_ERR_EMPTY the reference's own
 * UnsupportedOperationException("empty.max"); -inf is a VALID return of logTransition (NonRigidIcpProposal.scala:72-74).
 *
 * Build (where a JDK is present):  cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *                                     icp_jni.c -L../../icp-proposal_amd -licp_proposal_amd -o libicp_jni.so
 * The build image of this repository has no JDK: the body is compiled only where <jni.h> exists (`make -C bindings/jni` reports which
 * case applies).  tests/ compile it against tests/support/jni_mock/jni.h — a test double that declares the JNI functions used here
 * with the signatures of the JNI specification and implements them over plain C arrays — so that every native is type-checked
 * against include/icp_proposal.h on the CPU (tests/test_bindings_cpu.py) and RUN on the GPU box (tests/test_gpu_jni.py).
 */
#if defined(__has_include)
#if __has_include(<jni.h>)
#define ICP_HAVE_JNI 1
#endif
#endif

#ifdef ICP_HAVE_JNI
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "icp_proposal.h"

#define PTR(T, h) ((T *)(intptr_t)(h))
#define NATIVE(ret, name) JNIEXPORT ret JNICALL Java_api_gpu_NativeIcp_00024_##name
enum { kStack = 288, kMaxProps = 8 }; /* 10 + rank doubles fit the stack buffer up to rank 278 (the library's fast paths end at 256) */

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
static void throw_oom(JNIEnv *env) {
  jclass cls = (*env)->FindClass(env, "java/lang/OutOfMemoryError");
  if (cls) (*env)->ThrowNew(env, cls, "icp_jni: native buffer");
}

/* ---- region copies: a Java array into a C buffer (the caller's stack buffer when it fits, else the heap) and back */
typedef struct { double *p; jsize n; int heap; } dvec;
static int dvec_in(JNIEnv *env, jdoubleArray a, double *stack, jsize stack_n, dvec *v) {
  v->p = 0; v->n = 0; v->heap = 0;
  if (!a) return 1;
  v->n = (*env)->GetArrayLength(env, a);
  if (v->n <= stack_n && stack) v->p = stack;
  else { v->p = (double *)malloc(sizeof(double) * (size_t)(v->n > 0 ? v->n : 1)); v->heap = 1; }
  if (!v->p) { throw_oom(env); return 0; }
  if (v->n > 0) (*env)->GetDoubleArrayRegion(env, a, 0, v->n, v->p);
  return 1;
}
static void dvec_out(JNIEnv *env, jdoubleArray a, const dvec *v) {
  if (a && v->p && v->n > 0) (*env)->SetDoubleArrayRegion(env, a, 0, v->n, v->p);
}
static void dvec_free(dvec *v) {
  if (v->heap && v->p) free(v->p);
  v->p = 0;
}
static int handles_in(JNIEnv *env, jlongArray a, void **out, jsize cap, jsize *n_out) {
  jlong tmp[64];
  jsize n = a ? (*env)->GetArrayLength(env, a) : 0;
  *n_out = n;
  if (n > cap || n > 64) return 0;
  if (n > 0) (*env)->GetLongArrayRegion(env, a, 0, n, tmp);
  for (jsize i = 0; i < n; ++i) out[i] = PTR(void, tmp[i]);
  return 1;
}
static jlongArray longs_out(JNIEnv *env, const int64_t *v, jsize n) {
  jlongArray out = (*env)->NewLongArray(env, n);
  if (!out) return 0;
  jlong jv[16];
  for (jsize i = 0; i < n && i < 16; ++i) jv[i] = (jlong)v[i];
  (*env)->SetLongArrayRegion(env, out, 0, n, jv);
  return out;
}

/* ================================================================== contexts */

static jlong ctx_create(JNIEnv *env, jint n, jint t, jint r, jdoubleArray ref, jdoubleArray mean, jdoubleArray basis, jdoubleArray variance,
                        jintArray tris, jint m, jint tt, jdoubleArray tpts, jintArray ttris, jint device, jlong model_key) {
  jdouble *a_ref = (*env)->GetDoubleArrayElements(env, ref, 0), *a_mean = mean ? (*env)->GetDoubleArrayElements(env, mean, 0) : 0;
  jdouble *a_basis = (*env)->GetDoubleArrayElements(env, basis, 0), *a_var = (*env)->GetDoubleArrayElements(env, variance, 0);
  jint *a_tris = (*env)->GetIntArrayElements(env, tris, 0), *a_ttris = (*env)->GetIntArrayElements(env, ttris, 0);
  jdouble *a_tpts = (*env)->GetDoubleArrayElements(env, tpts, 0);
  icp_ctx *ctx = 0;
  int st = ICP_OK;
  /* Get*ArrayElements returns NULL when the JVM is out of memory (an OutOfMemoryError is then pending): never dereferenced */
  const int failed = !a_ref || (mean && !a_mean) || !a_basis || !a_var || !a_tris || !a_ttris || !a_tpts;
  if (!failed) {
    icp_model_desc md = {n, t, r, a_ref, a_mean, a_basis, a_var, (const int32_t *)a_tris};
    icp_mesh_desc td = {m, tt, a_tpts, (const int32_t *)a_ttris};
    /* copies everything to HBM; keeps no JVM pointer */
    st = model_key ? icp_ctx_create_keyed(&md, &td, device, (uint64_t)model_key, &ctx) : icp_ctx_create(&md, &td, device, &ctx);
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

NATIVE(jlong, ctxCreate)(JNIEnv *env, jobject self, jint n, jint t, jint r, jdoubleArray ref, jdoubleArray mean, jdoubleArray basis,
                         jdoubleArray variance, jintArray tris, jint m, jint tt, jdoubleArray tpts, jintArray ttris, jint device) {
  (void)self;
  return ctx_create(env, n, t, r, ref, mean, basis, variance, tris, m, tt, tpts, ttris, device, 0);
}

/* icp_ctx_create_keyed: one context per chain thread of a `.par` experiment (RunMHRandomInitComparison.scala:59-66) or per work item of
 * a batch registration (StdIcpVsChainICPrandomInitComparisonAll.scala:106-163); modelKey != 0 identifies the StatisticalMeshModel */
NATIVE(jlong, ctxCreateKeyed)(JNIEnv *env, jobject self, jint n, jint t, jint r, jdoubleArray ref, jdoubleArray mean, jdoubleArray basis,
                              jdoubleArray variance, jintArray tris, jint m, jint tt, jdoubleArray tpts, jintArray ttris, jint device,
                              jlong model_key) {
  (void)self;
  return ctx_create(env, n, t, r, ref, mean, basis, variance, tris, m, tt, tpts, ttris, device, model_key);
}

NATIVE(void, ctxDestroy)(JNIEnv *env, jobject self, jlong ctx) {
  (void)env; (void)self;
  icp_ctx_destroy(PTR(icp_ctx, ctx));
}

/* icp_ctx_set_target: the context goes on to the next target of a batch registration (its proposals / evaluators destroyed first) */
NATIVE(void, ctxSetTarget)(JNIEnv *env, jobject self, jlong ctx, jint m, jint tt, jdoubleArray tpts, jintArray ttris) {
  (void)self;
  jdouble *a_tpts = (*env)->GetDoubleArrayElements(env, tpts, 0);
  jint *a_ttris = (*env)->GetIntArrayElements(env, ttris, 0);
  int st = ICP_OK;
  const int failed = !a_tpts || !a_ttris;
  if (!failed) {
    icp_mesh_desc td = {m, tt, a_tpts, (const int32_t *)a_ttris};
    st = icp_ctx_set_target(PTR(icp_ctx, ctx), &td);
  }
  if (a_tpts) (*env)->ReleaseDoubleArrayElements(env, tpts, a_tpts, JNI_ABORT);
  if (a_ttris) (*env)->ReleaseIntArrayElements(env, ttris, a_ttris, JNI_ABORT);
  if (!failed) throw_status(env, st);
}

NATIVE(jint, ctxRank)(JNIEnv *env, jobject self, jlong ctx) {
  (void)env; (void)self;
  return icp_ctx_rank(PTR(icp_ctx, ctx));
}

/* Scalismo's own Rotation(phi, theta, psi, centre) matrix for a theta's Euler angles (ModelFittingParameters.scala:79-86) */
NATIVE(void, setRotation)(JNIEnv *env, jobject self, jlong ctx, jdoubleArray angles, jdoubleArray rot) {
  (void)self;
  double a[3], r[9];
  if ((*env)->GetArrayLength(env, angles) != 3 || (rot && (*env)->GetArrayLength(env, rot) != 9)) { throw_status(env, ICP_ERR_INVALID_ARG); return; }
  (*env)->GetDoubleArrayRegion(env, angles, 0, 3, a);
  if (rot) (*env)->GetDoubleArrayRegion(env, rot, 0, 9, r);
  throw_status(env, icp_ctx_set_rotation(PTR(icp_ctx, ctx), a, rot ? r : 0));
}

/* icp_ctx_rotation_convention: [verified, mismatched] — did every matrix registered so far agree with the library's Rz·Ry·Rx? */
NATIVE(jlongArray, rotationConvention)(JNIEnv *env, jobject self, jlong ctx) {
  (void)self;
  int64_t v[2] = {0, 0};
  int st = icp_ctx_rotation_convention(PTR(icp_ctx, ctx), &v[0], &v[1]);
  if (st != ICP_OK) { throw_status(env, st); return 0; }
  return longs_out(env, v, 2);
}

/* ================================================================== proposals and evaluators */

NATIVE(jlong, proposalCreate)(JNIEnv *env, jobject self, jlong ctx, jdouble step, jdouble sigma_t, jdouble sigma_n, jint direction,
                              jboolean boundary_aware, jint n_model_ids, jdoubleArray target_pts) {
  (void)self;
  dvec tp;
  if (!dvec_in(env, target_pts, 0, 0, &tp)) return 0;
  icp_proposal_params prm = {step, sigma_t, sigma_n, direction, boundary_aware ? 1 : 0, n_model_ids, (int32_t)(tp.n / 3), tp.p};
  icp_proposal *p = 0;
  int st = icp_proposal_create(PTR(icp_ctx, ctx), &prm, &p);
  dvec_free(&tp);
  throw_status(env, st);
  return (jlong)(intptr_t)p;
}
NATIVE(void, proposalDestroy)(JNIEnv *env, jobject self, jlong p) {
  (void)env; (void)self;
  icp_proposal_destroy(PTR(icp_proposal, p));
}
/* opt-in, NOT the reference's arithmetic (icp_proposal_set_sampler): 0 = the KL basis of posterior.sample(), 1 = Cholesky root */
NATIVE(void, proposalSetSampler)(JNIEnv *env, jobject self, jlong p, jint sampler) {
  (void)self;
  throw_status(env, icp_proposal_set_sampler(PTR(icp_proposal, p), sampler));
}

NATIVE(jlong, evaluatorCreate)(JNIEnv *env, jobject self, jlong ctx, jint kind, jint mode, jint n_model_ids, jdoubleArray target_pts,
                               jdouble gauss_mean, jdouble gauss_sigma, jdouble exp_rate) {
  (void)self;
  dvec tp;
  if (!dvec_in(env, target_pts, 0, 0, &tp)) return 0;
  icp_evaluator_params prm = {kind, mode, n_model_ids, (int32_t)(tp.n / 3), tp.p, gauss_mean, gauss_sigma, exp_rate};
  icp_evaluator *e = 0;
  int st = icp_evaluator_create(PTR(icp_ctx, ctx), &prm, &e);
  dvec_free(&tp);
  throw_status(env, st);
  return (jlong)(intptr_t)e;
}
NATIVE(void, evaluatorDestroy)(JNIEnv *env, jobject self, jlong e) {
  (void)env; (void)self;
  icp_evaluator_destroy(PTR(icp_evaluator, e));
}

/* ================================================================== the three plug-in methods (SamplingRegistration.scala:52-58) */

NATIVE(void, propose)(JNIEnv *env, jobject self, jlong prop, jdoubleArray theta, jdoubleArray z, jdoubleArray out) {
  (void)self;
  double s_th[kStack], s_z[kStack], s_o[kStack];
  dvec th, zz, o;
  if (!dvec_in(env, theta, s_th, kStack, &th)) return;
  if (!dvec_in(env, z, s_z, kStack, &zz)) { dvec_free(&th); return; }
  o.n = th.n; o.heap = th.n > kStack; o.p = o.heap ? (double *)malloc(sizeof(double) * (size_t)th.n) : s_o;
  int st = ICP_ERR_INVALID_ARG;
  if (!o.p) throw_oom(env);
  else if ((*env)->GetArrayLength(env, out) == th.n) {
    st = icp_proposal_propose(PTR(icp_proposal, prop), th.p, zz.p, o.p, 0);
    if (st == ICP_OK) dvec_out(env, out, &o);
  }
  dvec_free(&o); dvec_free(&zz); dvec_free(&th);
  throw_status(env, st);
}

NATIVE(jdouble, logTransition)(JNIEnv *env, jobject self, jlong prop, jdoubleArray from, jdoubleArray to) {
  (void)self;
  double s_a[kStack], s_b[kStack], v = 0;
  dvec a, b;
  if (!dvec_in(env, from, s_a, kStack, &a)) return 0;
  if (!dvec_in(env, to, s_b, kStack, &b)) { dvec_free(&a); return 0; }
  int st = a.n == b.n ? icp_proposal_log_transition(PTR(icp_proposal, prop), a.p, b.p, &v) : ICP_ERR_INVALID_ARG; /* -inf is a valid value */
  dvec_free(&b); dvec_free(&a);
  throw_status(env, st);
  return v;
}

NATIVE(jdouble, logValue)(JNIEnv *env, jobject self, jlong ev, jdoubleArray theta) {
  (void)self;
  double s_th[kStack], v = 0;
  dvec th;
  if (!dvec_in(env, theta, s_th, kStack, &th)) return 0;
  int st = icp_evaluator_log_value(PTR(icp_evaluator, ev), th.p, &v, 0);
  dvec_free(&th);
  throw_status(env, st);
  return v;
}

/* icp_chain_bind: the evaluator and the ICP proposals of ONE MetropolisHastings chain (in the mixture's order).  After it the natives
 * above, called one by one by Scalismo's loop, cost one device submission per step instead of six. */
NATIVE(void, chainBind)(JNIEnv *env, jobject self, jlong ev, jlongArray props) {
  (void)self;
  void *pp[kMaxProps];
  jsize n = 0;
  if (!handles_in(env, props, pp, kMaxProps, &n)) { throw_status(env, ICP_ERR_INVALID_ARG); return; }
  throw_status(env, icp_chain_bind(PTR(icp_evaluator, ev), (int32_t)n, (icp_proposal *const *)pp));
}
NATIVE(jlongArray, chainBindStats)(JNIEnv *env, jobject self, jlong ev) {
  (void)self;
  int64_t v[3] = {0, 0, 0};
  int st = icp_chain_bind_stats(PTR(icp_evaluator, ev), v);
  if (st != ICP_OK) { throw_status(env, st); return 0; }
  return longs_out(env, v, 3);
}

/* ================================================================== whole steps */

/* one MH step in one submission (icp_chain_step).  Returns the likelihood of the proposal. */
NATIVE(jdouble, chainStep)(JNIEnv *env, jobject self, jlong ev, jlongArray props, jint generator, jdoubleArray theta_cur, jdoubleArray z,
                           jdoubleArray theta_prop, jdoubleArray fwd, jdoubleArray bwd) {
  (void)self;
  void *pp[kMaxProps];
  jsize n = 0;
  if (!handles_in(env, props, pp, kMaxProps, &n)) { throw_status(env, ICP_ERR_INVALID_ARG); return 0; }
  double s_cur[kStack], s_z[kStack], s_prop[kStack], f[kMaxProps], b[kMaxProps], v = 0;
  dvec cur, zz, prp;
  if (!dvec_in(env, theta_cur, s_cur, kStack, &cur)) return 0;
  if (!dvec_in(env, z, s_z, kStack, &zz)) { dvec_free(&cur); return 0; }
  if (!dvec_in(env, theta_prop, s_prop, kStack, &prp)) { dvec_free(&zz); dvec_free(&cur); return 0; }
  int st = ICP_ERR_INVALID_ARG;
  if (prp.n == cur.n && (*env)->GetArrayLength(env, fwd) >= n && (*env)->GetArrayLength(env, bwd) >= n) {
    st = icp_chain_step(PTR(icp_evaluator, ev), (int32_t)n, (icp_proposal *const *)pp, generator, cur.p, zz.p, prp.p, &v, f, b);
    if (st == ICP_OK || st == ICP_ERR_EMPTY) {
      dvec_out(env, theta_prop, &prp);
      if (n > 0) { (*env)->SetDoubleArrayRegion(env, fwd, 0, n, f); (*env)->SetDoubleArrayRegion(env, bwd, 0, n, b); }
    }
  }
  dvec_free(&prp); dvec_free(&zz); dvec_free(&cur);
  throw_status(env, st);
  return v;
}

/* ---- B chains per submission (icp_chain_step_batched and its two halves).  Flat arrays: thetaCur / thetaProp [B*(10+r)], z [B*r] (null
 * when no chain's generator is >= 0), logValue [B], fwd / bwd [B*nProps], status [B]; props [B*nProps] handles. */
typedef struct {
  int32_t B, n_props;
  jsize P, r;
  icp_evaluator **ev;
  icp_proposal **props;
  int32_t *gen, *status;
  double *cur, *z, *prop, *lv, *fwd, *bwd;
  const double **cur_p, **z_p;
  double **prop_p;
  icp_step_ticket *ticket;
} jni_batch;

static void batch_free(jni_batch *b) {
  if (!b) return;
  free(b->ev); free(b->props); free(b->gen); free(b->status); free(b->cur); free(b->z); free(b->prop); free(b->lv); free(b->fwd);
  free(b->bwd); free((void *)b->cur_p); free((void *)b->z_p); free(b->prop_p);
  free(b);
}

static jni_batch *batch_in(JNIEnv *env, jlongArray evs, jint n_props, jlongArray props, jintArray generator, jdoubleArray theta_cur,
                           jdoubleArray z, jdoubleArray theta_prop) {
  const jsize B = evs ? (*env)->GetArrayLength(env, evs) : 0;
  if (B < 1 || n_props < 0 || n_props > kMaxProps || !generator || (*env)->GetArrayLength(env, generator) != B || !theta_cur || !theta_prop ||
      (*env)->GetArrayLength(env, props) != B * n_props) { throw_status(env, ICP_ERR_INVALID_ARG); return 0; }
  const jsize total = (*env)->GetArrayLength(env, theta_cur);
  if (total % B != 0 || total / B <= 10 || (*env)->GetArrayLength(env, theta_prop) != total ||
      (z && (*env)->GetArrayLength(env, z) != B * (total / B - 10))) { throw_status(env, ICP_ERR_INVALID_ARG); return 0; }
  jni_batch *b = (jni_batch *)calloc(1, sizeof(jni_batch));
  if (!b) { throw_oom(env); return 0; }
  b->B = (int32_t)B; b->n_props = n_props; b->P = total / B; b->r = b->P - 10;
  const size_t nb = (size_t)B, np = nb * (size_t)(n_props > 0 ? n_props : 1);
  b->ev = (icp_evaluator **)calloc(nb, sizeof(void *)); b->props = (icp_proposal **)calloc(np, sizeof(void *));
  b->gen = (int32_t *)calloc(nb, sizeof(int32_t)); b->status = (int32_t *)calloc(nb, sizeof(int32_t));
  b->cur = (double *)calloc((size_t)total, sizeof(double)); b->prop = (double *)calloc((size_t)total, sizeof(double));
  b->z = (double *)calloc(nb * (size_t)b->r, sizeof(double));
  b->lv = (double *)calloc(nb, sizeof(double)); b->fwd = (double *)calloc(np, sizeof(double)); b->bwd = (double *)calloc(np, sizeof(double));
  b->cur_p = (const double **)calloc(nb, sizeof(void *)); b->z_p = (const double **)calloc(nb, sizeof(void *));
  b->prop_p = (double **)calloc(nb, sizeof(void *));
  jlong *hl = (jlong *)calloc(np > nb ? np : nb, sizeof(jlong));
  if (!b->ev || !b->props || !b->gen || !b->status || !b->cur || !b->prop || !b->z || !b->lv || !b->fwd || !b->bwd || !b->cur_p || !b->z_p ||
      !b->prop_p || !hl) { free(hl); batch_free(b); throw_oom(env); return 0; }
  (*env)->GetLongArrayRegion(env, evs, 0, B, hl);
  for (jsize i = 0; i < B; ++i) b->ev[i] = PTR(icp_evaluator, hl[i]);
  if (n_props > 0) {
    (*env)->GetLongArrayRegion(env, props, 0, B * n_props, hl);
    for (jsize i = 0; i < B * n_props; ++i) b->props[i] = PTR(icp_proposal, hl[i]);
  }
  free(hl);
  (*env)->GetIntArrayRegion(env, generator, 0, B, (jint *)b->gen);
  (*env)->GetDoubleArrayRegion(env, theta_cur, 0, total, b->cur);
  (*env)->GetDoubleArrayRegion(env, theta_prop, 0, total, b->prop);
  if (z) (*env)->GetDoubleArrayRegion(env, z, 0, B * b->r, b->z);
  for (jsize i = 0; i < B; ++i) {
    b->cur_p[i] = b->cur + (size_t)i * (size_t)b->P;
    b->prop_p[i] = b->prop + (size_t)i * (size_t)b->P;
    b->z_p[i] = z ? b->z + (size_t)i * (size_t)b->r : 0;
  }
  return b;
}

static int batch_out(JNIEnv *env, const jni_batch *b, jdoubleArray theta_prop, jdoubleArray log_value, jdoubleArray fwd, jdoubleArray bwd,
                     jintArray status) {
  if ((*env)->GetArrayLength(env, theta_prop) != b->B * b->P || (*env)->GetArrayLength(env, log_value) != b->B ||
      (*env)->GetArrayLength(env, fwd) != b->B * b->n_props || (*env)->GetArrayLength(env, bwd) != b->B * b->n_props ||
      (*env)->GetArrayLength(env, status) != b->B) return 0;
  (*env)->SetDoubleArrayRegion(env, theta_prop, 0, b->B * b->P, b->prop);
  (*env)->SetDoubleArrayRegion(env, log_value, 0, b->B, b->lv);
  if (b->n_props > 0) {
    (*env)->SetDoubleArrayRegion(env, fwd, 0, b->B * b->n_props, b->fwd);
    (*env)->SetDoubleArrayRegion(env, bwd, 0, b->B * b->n_props, b->bwd);
  }
  (*env)->SetIntArrayRegion(env, status, 0, b->B, (const jint *)b->status);
  return 1;
}

/* returns ICP_OK or the first failing chain's code; per-chain codes in status (ICP_ERR_EMPTY of a chain's evaluator is not thrown here:
 * the adapter raises it when that chain's logValue is asked for) */
NATIVE(jint, chainStepBatched)(JNIEnv *env, jobject self, jlongArray evs, jint n_props, jlongArray props, jintArray generator,
                               jdoubleArray theta_cur, jdoubleArray z, jdoubleArray theta_prop, jdoubleArray log_value, jdoubleArray fwd,
                               jdoubleArray bwd, jintArray status) {
  (void)self;
  jni_batch *b = batch_in(env, evs, n_props, props, generator, theta_cur, z, theta_prop);
  if (!b) return ICP_ERR_INVALID_ARG;
  int st = icp_chain_step_batched(b->B, b->ev, b->n_props, b->props, b->gen, b->cur_p, z ? b->z_p : 0, b->prop_p, b->lv, b->fwd, b->bwd, b->status);
  const int ok = batch_out(env, b, theta_prop, log_value, fwd, bwd, status);
  batch_free(b);
  if (!ok) { throw_status(env, ICP_ERR_INVALID_ARG); return ICP_ERR_INVALID_ARG; }
  if (st != ICP_OK && st != ICP_ERR_EMPTY) throw_status(env, st);
  return st;
}

/* the same in two halves: the ticket owns native copies of everything the library reads or writes until it is collected */
NATIVE(jlong, chainStepBatchedIssue)(JNIEnv *env, jobject self, jlongArray evs, jint n_props, jlongArray props, jintArray generator,
                                     jdoubleArray theta_cur, jdoubleArray z, jdoubleArray theta_prop, jlong launch_ctx) {
  (void)self;
  jni_batch *b = batch_in(env, evs, n_props, props, generator, theta_cur, z, theta_prop);
  if (!b) return 0;
  int st = icp_chain_step_batched_issue(b->B, b->ev, b->n_props, b->props, b->gen, b->cur_p, z ? b->z_p : 0, b->prop_p, b->lv, b->fwd, b->bwd,
                                        b->status, PTR(icp_ctx, launch_ctx), &b->ticket);
  if (st != ICP_OK) { batch_free(b); throw_status(env, st); return 0; }
  return (jlong)(intptr_t)b;
}
NATIVE(jint, chainStepBatchedCollect)(JNIEnv *env, jobject self, jlong ticket, jdoubleArray theta_prop, jdoubleArray log_value, jdoubleArray fwd,
                                      jdoubleArray bwd, jintArray status) {
  (void)self;
  jni_batch *b = PTR(jni_batch, ticket);
  if (!b) { throw_status(env, ICP_ERR_INVALID_ARG); return ICP_ERR_INVALID_ARG; }
  int st = icp_chain_step_batched_collect(b->ticket);
  const int ok = batch_out(env, b, theta_prop, log_value, fwd, bwd, status);
  batch_free(b);
  if (!ok) { throw_status(env, ICP_ERR_INVALID_ARG); return ICP_ERR_INVALID_ARG; }
  if (st != ICP_OK && st != ICP_ERR_EMPTY) throw_status(env, st);
  return st;
}
NATIVE(void, chainStepBatchedAbandon)(JNIEnv *env, jobject self, jlong ticket) {
  (void)self;
  jni_batch *b = PTR(jni_batch, ticket);
  if (!b) return;
  int st = icp_chain_step_batched_abandon(b->ticket);
  batch_free(b);
  throw_status(env, st);
}

/* ---- the whole Metropolis–Hastings loop of B chains on the device (icp_chains_run_on_device): the replacement of the
 * `(0 until 5).par.foreach { … runfitting … }` block of apps/femur/RunMHRandomInitComparison.scala:66-87 and of the chains of one target in
 * apps/bfm/BfmFittingPartial.scala:62-96.  mixture = [icpWeight0, icpWeight1, wIcp, wRw, rwSigma, wPose, rotYaw, rotPitch, rotRoll, transX,
 * transY, transZ] (12 doubles: the fields of icp_mh_mixture behind struct_size, which is filled in here from the header this file was
 * compiled against); theta [B*(10+r)] and logValue [B] in/out; records null or [B*nSteps*(4+10+r)]; accepted [B] out. */
NATIVE(void, chainsRunOnDevice)(JNIEnv *env, jobject self, jlongArray evs, jint n_props, jlongArray props, jdoubleArray mixture,
                                jlongArray seeds, jlongArray first_step, jdoubleArray theta, jdoubleArray log_value, jint n_steps,
                                jdoubleArray records, jlongArray accepted) {
  (void)self;
  const jsize B = evs ? (*env)->GetArrayLength(env, evs) : 0;
  if (B < 1 || n_props < 0 || n_props > kMaxProps || !mixture || (*env)->GetArrayLength(env, mixture) != 12 || !seeds || !first_step || !theta ||
      !log_value || !accepted || n_steps < 0 || (*env)->GetArrayLength(env, props) != B * n_props || (*env)->GetArrayLength(env, seeds) != B ||
      (*env)->GetArrayLength(env, first_step) != B || (*env)->GetArrayLength(env, log_value) != B || (*env)->GetArrayLength(env, accepted) != B) {
    throw_status(env, ICP_ERR_INVALID_ARG);
    return;
  }
  const jsize total = (*env)->GetArrayLength(env, theta), P = total / B;
  const jsize rec_len = (jsize)(4 + P), rec_total = B * n_steps * rec_len;
  if (total % B != 0 || P <= 10 || (records && (*env)->GetArrayLength(env, records) != rec_total)) { throw_status(env, ICP_ERR_INVALID_ARG); return; }
  double m[12];
  (*env)->GetDoubleArrayRegion(env, mixture, 0, 12, m);
  icp_mh_mixture mix;
  memset(&mix, 0, sizeof mix);
  mix.struct_size = sizeof(icp_mh_mixture);
  mix.icp_weight[0] = m[0]; mix.icp_weight[1] = m[1]; mix.w_icp = m[2]; mix.w_rw = m[3]; mix.rw_sigma = m[4]; mix.w_pose = m[5];
  for (int k = 0; k < 3; ++k) { mix.pose_rot_sigma[k] = m[6 + k]; mix.pose_trans_sigma[k] = m[9 + k]; }
  const size_t nb = (size_t)B, np = nb * (size_t)(n_props > 0 ? n_props : 1);
  icp_evaluator **ev = (icp_evaluator **)calloc(nb, sizeof(void *));
  icp_proposal **pp = (icp_proposal **)calloc(np, sizeof(void *));
  jlong *hl = (jlong *)calloc(np > nb ? np : nb, sizeof(jlong));
  uint64_t *sd = (uint64_t *)calloc(nb, sizeof(uint64_t));
  int64_t *fs = (int64_t *)calloc(nb, sizeof(int64_t)), *acc = (int64_t *)calloc(nb, sizeof(int64_t));
  double *th = (double *)calloc((size_t)total, sizeof(double)), *lv = (double *)calloc(nb, sizeof(double));
  double *rec = records ? (double *)calloc((size_t)(rec_total > 0 ? rec_total : 1), sizeof(double)) : 0;
  double **th_p = (double **)calloc(nb, sizeof(void *)), **rec_p = (double **)calloc(nb, sizeof(void *));
  int st = ICP_ERR_DEVICE;
  if (!ev || !pp || !hl || !sd || !fs || !acc || !th || !lv || (records && !rec) || !th_p || !rec_p) throw_oom(env);
  else {
    (*env)->GetLongArrayRegion(env, evs, 0, B, hl);
    for (jsize i = 0; i < B; ++i) ev[i] = PTR(icp_evaluator, hl[i]);
    if (n_props > 0) {
      (*env)->GetLongArrayRegion(env, props, 0, B * n_props, hl);
      for (jsize i = 0; i < B * n_props; ++i) pp[i] = PTR(icp_proposal, hl[i]);
    }
    (*env)->GetLongArrayRegion(env, seeds, 0, B, hl);
    for (jsize i = 0; i < B; ++i) sd[i] = (uint64_t)hl[i];
    (*env)->GetLongArrayRegion(env, first_step, 0, B, hl);
    for (jsize i = 0; i < B; ++i) fs[i] = (int64_t)hl[i];
    (*env)->GetDoubleArrayRegion(env, theta, 0, total, th);
    (*env)->GetDoubleArrayRegion(env, log_value, 0, B, lv);
    for (jsize i = 0; i < B; ++i) {
      th_p[i] = th + (size_t)i * (size_t)P;
      rec_p[i] = rec ? rec + (size_t)i * (size_t)n_steps * (size_t)rec_len : 0;
    }
    st = icp_chains_run_on_device((int32_t)B, ev, n_props, pp, &mix, sd, fs, th_p, lv, n_steps, rec ? rec_p : 0, acc);
    if (st == ICP_OK) {
      (*env)->SetDoubleArrayRegion(env, theta, 0, total, th);
      (*env)->SetDoubleArrayRegion(env, log_value, 0, B, lv);
      if (rec && rec_total > 0) (*env)->SetDoubleArrayRegion(env, records, 0, rec_total, rec);
      for (jsize i = 0; i < B; ++i) hl[i] = (jlong)acc[i];
      (*env)->SetLongArrayRegion(env, accepted, 0, B, hl);
    }
    throw_status(env, st);
  }
  free(ev); free(pp); free(hl); free(sd); free(fs); free(acc); free(th); free(lv); free(rec); free(th_p); free(rec_p);
}

/* ================================================================== queries */

/* icp_chain_step_path: 0 the five merged launches, 1 the wide step, 2 per-stage kernels (how a pool of chains should be grouped) */
NATIVE(jint, chainStepPath)(JNIEnv *env, jobject self, jlong ev, jlongArray props) {
  (void)self;
  void *pp[kMaxProps];
  jsize n = 0;
  if (!handles_in(env, props, pp, kMaxProps, &n)) { throw_status(env, ICP_ERR_INVALID_ARG); return -1; }
  return icp_chain_step_path(PTR(icp_evaluator, ev), (int32_t)n, (icp_proposal *const *)pp);
}
/* icp_ctx_step_paths: [merged, wide, per-stage, device loop]; ctx == 0: of the process */
NATIVE(jlongArray, stepPaths)(JNIEnv *env, jobject self, jlong ctx) {
  (void)self;
  int64_t v[4] = {0, 0, 0, 0};
  int st = icp_ctx_step_paths(PTR(icp_ctx, ctx), v);
  if (st != ICP_OK) { throw_status(env, st); return 0; }
  return longs_out(env, v, 4);
}
/* icp_ctx_runtime_stats: [wait_timeouts, speculation_giveups, pipeline_fallbacks, step_redos, gate_timeouts] — zeros in a healthy run */
NATIVE(jlongArray, runtimeStats)(JNIEnv *env, jobject self, jlong ctx) {
  (void)self;
  icp_runtime_stats s;
  memset(&s, 0, sizeof s);
  int st = icp_ctx_runtime_stats(PTR(icp_ctx, ctx), &s);
  if (st != ICP_OK) { throw_status(env, st); return 0; }
  const int64_t v[5] = {s.wait_timeouts, s.speculation_giveups, s.pipeline_fallbacks, s.step_redos, s.gate_timeouts};
  return longs_out(env, v, 5);
}
/* RegistrationComparison.evaluateReconstruction2GroundTruth[BoundaryAware] (api/other/RegistrationComparison.scala:24-49) of the mesh of
 * theta against the context's target: [avg, hausdorff, boundary-aware avg, boundary-aware max, vertices kept] */
NATIVE(jdoubleArray, meshMetrics)(JNIEnv *env, jobject self, jlong ctx, jdoubleArray theta) {
  (void)self;
  double s_th[kStack], out5[5] = {0, 0, 0, 0, 0};
  dvec th;
  if (!dvec_in(env, theta, s_th, kStack, &th)) return 0;
  int st = icp_mesh_metrics(PTR(icp_ctx, ctx), th.p, out5);
  dvec_free(&th);
  if (st != ICP_OK) { throw_status(env, st); return 0; }
  jdoubleArray out = (*env)->NewDoubleArray(env, 5);
  if (out) (*env)->SetDoubleArrayRegion(env, out, 0, 5, out5);
  return out;
}
NATIVE(void, releaseCachedModels)(JNIEnv *env, jobject self) {
  (void)env; (void)self;
  icp_release_cached_models();
}
#else
/* no <jni.h> in this build environment: nothing to compile (see the header comment) */
typedef int icp_jni_not_built_here;
#endif
