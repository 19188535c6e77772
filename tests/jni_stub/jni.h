/* Minimal stand-in for <jni.h>, used ONLY by tests/test_host_logic.py to syntax-check jni/mhap_jni.c in an image without a JDK.
 * It declares the JNI types and the JNIEnv function-table members the shim uses, with the JNI specification's signatures;
 * nothing here is linked or run.  A real build uses the JDK's header. */
#ifndef MHAP_TEST_JNI_STUB_H
#define MHAP_TEST_JNI_STUB_H
#include <stdint.h>
typedef int32_t jint; typedef int64_t jlong; typedef int8_t jbyte; typedef uint8_t jboolean; typedef double jdouble; typedef jint jsize;
struct _jobject; typedef struct _jobject* jobject; typedef jobject jclass; typedef jobject jstring; typedef jobject jarray;
typedef jarray jbyteArray; typedef jarray jintArray; typedef jarray jlongArray; typedef jobject jthrowable;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv*, const char*);
  jint (*ThrowNew)(JNIEnv*, jclass, const char*);
  const char* (*GetStringUTFChars)(JNIEnv*, jstring, jboolean*);
  void (*ReleaseStringUTFChars)(JNIEnv*, jstring, const char*);
  void* (*GetPrimitiveArrayCritical)(JNIEnv*, jarray, jboolean*);
  void (*ReleasePrimitiveArrayCritical)(JNIEnv*, jarray, void*, jint);
  jbyteArray (*NewByteArray)(JNIEnv*, jsize);
  void (*SetByteArrayRegion)(JNIEnv*, jbyteArray, jsize, jsize, const jbyte*);
  jsize (*GetArrayLength)(JNIEnv*, jarray);
  void (*GetByteArrayRegion)(JNIEnv*, jbyteArray, jsize, jsize, jbyte*);
  void (*GetIntArrayRegion)(JNIEnv*, jintArray, jsize, jsize, jint*);
  void (*GetLongArrayRegion)(JNIEnv*, jlongArray, jsize, jsize, jlong*);
  jlongArray (*NewLongArray)(JNIEnv*, jsize);
  void (*SetLongArrayRegion)(JNIEnv*, jlongArray, jsize, jsize, const jlong*);
  jboolean (*ExceptionCheck)(JNIEnv*);
};
#endif
