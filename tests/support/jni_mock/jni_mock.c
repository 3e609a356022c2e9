/*
 * jni_mock.c — TEST DOUBLE: the JNI functions bindings/jni/icp_jni.c calls, over plain C arrays (see jni.h beside this file).
 * Built together with the shim into tests/support/jni_mock/libicp_jni_mock.so; tests/test_gpu_jni.py drives the natives through it
 * with ctypes the way a JVM would: arrays are opaque handles, a thrown exception is recorded as (class name, message) and stays
 * pending until the test takes it.  Get<T>ArrayElements hands out a COPY (as a JVM may), so a shim that wrote through a pointer
 * released with JNI_ABORT would lose its writes here, too.
 */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct mock_object {
  int kind; /* 1 double[], 2 int[], 3 long[], 9 class */
  jsize n;
  void *data;
  char name[64];
};

static char g_exc_class[64], g_exc_msg[512];
static int g_exc_pending = 0;

static size_t elem_size(int kind) { return kind == 1 ? sizeof(jdouble) : kind == 2 ? sizeof(jint) : sizeof(jlong); }

static struct mock_object *new_array(int kind, jsize n) {
  struct mock_object *o = (struct mock_object *)calloc(1, sizeof *o);
  if (!o) return 0;
  o->kind = kind; o->n = n;
  o->data = calloc((size_t)(n > 0 ? n : 1), elem_size(kind));
  return o;
}

static jclass m_FindClass(JNIEnv *env, const char *name) {
  (void)env;
  struct mock_object *o = (struct mock_object *)calloc(1, sizeof *o);
  o->kind = 9;
  snprintf(o->name, sizeof o->name, "%s", name);
  return o; /* (leaked: a handful per failing call in a test process) */
}
static jint m_ThrowNew(JNIEnv *env, jclass c, const char *msg) {
  (void)env;
  snprintf(g_exc_class, sizeof g_exc_class, "%s", c ? c->name : "?");
  snprintf(g_exc_msg, sizeof g_exc_msg, "%s", msg ? msg : "");
  g_exc_pending = 1;
  return 0;
}
static jsize m_GetArrayLength(JNIEnv *env, jarray a) { (void)env; return a->n; }
static jdoubleArray m_NewDoubleArray(JNIEnv *env, jsize n) { (void)env; return new_array(1, n); }
static jlongArray m_NewLongArray(JNIEnv *env, jsize n) { (void)env; return new_array(3, n); }
static void *elements_copy(jarray a) {
  void *p = malloc((size_t)(a->n > 0 ? a->n : 1) * elem_size(a->kind));
  if (p) memcpy(p, a->data, (size_t)a->n * elem_size(a->kind));
  return p;
}
static jdouble *m_GetDoubleArrayElements(JNIEnv *env, jdoubleArray a, jboolean *is_copy) {
  (void)env;
  if (is_copy) *is_copy = JNI_TRUE;
  return (jdouble *)elements_copy(a);
}
static jint *m_GetIntArrayElements(JNIEnv *env, jintArray a, jboolean *is_copy) {
  (void)env;
  if (is_copy) *is_copy = JNI_TRUE;
  return (jint *)elements_copy(a);
}
static void release_copy(jarray a, void *p, jint mode) {
  if (mode != JNI_ABORT) memcpy(a->data, p, (size_t)a->n * elem_size(a->kind));
  free(p);
}
static void m_ReleaseDoubleArrayElements(JNIEnv *env, jdoubleArray a, jdouble *p, jint mode) { (void)env; release_copy(a, p, mode); }
static void m_ReleaseIntArrayElements(JNIEnv *env, jintArray a, jint *p, jint mode) { (void)env; release_copy(a, p, mode); }

#define REGION(NAME, T, KIND)                                                                            \
  static void m_Get##NAME##ArrayRegion(JNIEnv *env, jarray a, jsize s, jsize len, T *buf) {              \
    (void)env;                                                                                           \
    if (a->kind != KIND || s < 0 || len < 0 || s + len > a->n) { m_ThrowNew(env, 0, "ArrayIndexOutOfBounds (mock)"); return; } \
    memcpy(buf, (T *)a->data + s, sizeof(T) * (size_t)len);                                              \
  }                                                                                                      \
  static void m_Set##NAME##ArrayRegion(JNIEnv *env, jarray a, jsize s, jsize len, const T *buf) {        \
    (void)env;                                                                                           \
    if (a->kind != KIND || s < 0 || len < 0 || s + len > a->n) { m_ThrowNew(env, 0, "ArrayIndexOutOfBounds (mock)"); return; } \
    memcpy((T *)a->data + s, buf, sizeof(T) * (size_t)len);                                              \
  }
REGION(Double, jdouble, 1)
REGION(Int, jint, 2)
REGION(Long, jlong, 3)

static const struct JNINativeInterface_ g_table = {
    m_FindClass, m_ThrowNew, m_GetArrayLength, m_NewDoubleArray, m_NewLongArray, m_GetDoubleArrayElements, m_GetIntArrayElements,
    m_ReleaseDoubleArrayElements, m_ReleaseIntArrayElements, m_GetDoubleArrayRegion, m_SetDoubleArrayRegion, m_GetIntArrayRegion,
    m_SetIntArrayRegion, m_GetLongArrayRegion, m_SetLongArrayRegion};
static JNIEnv g_env = &g_table;

/* ---- what the test (ctypes) uses */
JNIEXPORT JNIEnv *mock_env(void) { return &g_env; }
JNIEXPORT jarray mock_new_array(int kind, jsize n, const void *init) {
  struct mock_object *o = new_array(kind, n);
  if (o && init && n > 0) memcpy(o->data, init, (size_t)n * elem_size(kind));
  return o;
}
JNIEXPORT void *mock_array_data(jarray a) { return a ? a->data : 0; }
JNIEXPORT jsize mock_array_length(jarray a) { return a ? a->n : -1; }
JNIEXPORT void mock_free_array(jarray a) {
  if (a) { free(a->data); free(a); }
}
/* 1 and the pending exception's class / message, cleared; 0 if none */
JNIEXPORT int mock_take_exception(char *cls, int cls_len, char *msg, int msg_len) {
  if (!g_exc_pending) return 0;
  snprintf(cls, (size_t)cls_len, "%s", g_exc_class);
  snprintf(msg, (size_t)msg_len, "%s", g_exc_msg);
  g_exc_pending = 0;
  return 1;
}
