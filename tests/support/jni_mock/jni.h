/*
 * jni.h — TEST DOUBLE, not a JDK header.  The build image has no JDK; this file declares the part of the Java Native Interface that
 * bindings/jni/icp_jni.c uses — the primitive types, the array handle types and, as members of JNINativeInterface_, the functions it
 * calls, each with the signature the JNI specification (chapter 4, "JNI Functions") gives it — so that the shim is type-checked against
 * include/icp_proposal.h on the CPU and can be RUN over plain C arrays on the GPU box (jni_mock.c implements the table).
 * The order of the table's members is NOT the JVM's: a library built against this header works with jni_mock.c only, never in a JVM.
 */
#ifndef ICP_TEST_JNI_MOCK_H
#define ICP_TEST_JNI_MOCK_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef uint8_t jboolean;
typedef double jdouble;
typedef jint jsize;

struct mock_object;
typedef struct mock_object *jobject;
typedef jobject jclass;
typedef jobject jthrowable;
typedef jobject jarray;
typedef jarray jdoubleArray;
typedef jarray jintArray;
typedef jarray jlongArray;

#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNI_ABORT 2
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv *env, const char *name);
  jint (*ThrowNew)(JNIEnv *env, jclass clazz, const char *message);
  jsize (*GetArrayLength)(JNIEnv *env, jarray array);
  jdoubleArray (*NewDoubleArray)(JNIEnv *env, jsize length);
  jlongArray (*NewLongArray)(JNIEnv *env, jsize length);
  jdouble *(*GetDoubleArrayElements)(JNIEnv *env, jdoubleArray array, jboolean *isCopy);
  jint *(*GetIntArrayElements)(JNIEnv *env, jintArray array, jboolean *isCopy);
  void (*ReleaseDoubleArrayElements)(JNIEnv *env, jdoubleArray array, jdouble *elems, jint mode);
  void (*ReleaseIntArrayElements)(JNIEnv *env, jintArray array, jint *elems, jint mode);
  void (*GetDoubleArrayRegion)(JNIEnv *env, jdoubleArray array, jsize start, jsize len, jdouble *buf);
  void (*SetDoubleArrayRegion)(JNIEnv *env, jdoubleArray array, jsize start, jsize len, const jdouble *buf);
  void (*GetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, jint *buf);
  void (*SetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, const jint *buf);
  void (*GetLongArrayRegion)(JNIEnv *env, jlongArray array, jsize start, jsize len, jlong *buf);
  void (*SetLongArrayRegion)(JNIEnv *env, jlongArray array, jsize start, jsize len, const jlong *buf);
};
#endif
