/* mhap_jni.c — JNI shim between MHAP's Java host and libmhaphip.so (include/mhap_hip.h).
 *
 * Binds the native methods of edu.umd.marbl.mhap.impl.HipMinHashSearch (jni/HipMinHashSearch.java), the GPU-backed
 * replacement of MinHashSearch behind the reference's only operator seam, AbstractMatchSearch
 * (src/main/java/edu/umd/marbl/mhap/impl/AbstractMatchSearch.java:47,67-117,121-199,203-285).
 *
 * Not compiled in this repository's image (no JDK: no jni.h).  Build where a JDK is present:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../include mhap_jni.c \
 *       -L../mhap_amd/lib -lmhaphip -Wl,-rpath,'$ORIGIN' -o libmhapjni.so
 * tests/test_host_logic.py checks that every native method of the Java class has its Java_... function here, that the
 * argument counts agree and that every mhap_* function called exists in the header; it also compiles this file against a
 * minimal stand-in jni.h for syntax.
 *
 * Conventions: no JNI global references are held across calls; arrays are pinned with Get/ReleasePrimitiveArrayCritical
 * only around the one library call that reads them; records come back as one byte[] of packed mhap_record (64 bytes each,
 * little-endian, the layout of include/mhap_hip.h) that the Java side decodes with a ByteBuffer; every library error
 * becomes an unchecked MhapRuntimeException carrying mhap_last_error().
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mhap_hip.h"

#define MHAP_EXC "edu/umd/marbl/mhap/impl/MhapRuntimeException"

static mhap_handle* H(jlong h) { return (mhap_handle*)(intptr_t)h; }

static void throw_mhap(JNIEnv* env, const char* msg) {
  jclass c = (*env)->FindClass(env, MHAP_EXC);
  if (c) (*env)->ThrowNew(env, c, msg ? msg : "libmhaphip error");
}

static int check(JNIEnv* env, mhap_handle* h, int rc) {
  if (rc == MHAP_OK) return 0;
  throw_mhap(env, h ? mhap_last_error(h) : "libmhaphip error");
  return 1;
}

/* growing buffer the record sink appends to (the library calls the sink from the calling thread, one batch at a time) */
typedef struct { mhap_record* p; int64_t n, cap; int oom; } rec_buf;

static int collect_sink(const mhap_record* recs, int64_t n, void* user) {
  rec_buf* b = (rec_buf*)user;
  if (b->n + n > b->cap) {
    int64_t ncap = b->cap ? b->cap : 65536;
    while (ncap < b->n + n) ncap *= 2;
    mhap_record* np = (mhap_record*)realloc(b->p, (size_t)ncap * sizeof(mhap_record));
    if (!np) { b->oom = 1; return 1; }
    b->p = np; b->cap = ncap;
  }
  memcpy(b->p + b->n, recs, (size_t)n * sizeof(mhap_record));
  b->n += n;
  return 0;
}

static jbyteArray records_to_java(JNIEnv* env, rec_buf* b) {
  jbyteArray out = NULL;
  if (b->oom) throw_mhap(env, "out of memory while collecting overlap records");
  else {
    const jsize bytes = (jsize)(b->n * (int64_t)sizeof(mhap_record));
    out = (*env)->NewByteArray(env, bytes);
    if (out && bytes > 0) (*env)->SetByteArrayRegion(env, out, 0, bytes, (const jbyte*)b->p);
  }
  free(b->p);
  return out;
}

/* long nativeCreate(int kmerSize, int numHashes, int orderedKmerSize, int orderedSketchSize, int numMinMatches,
 *                   int minStoreLength, int minOlapLength, double acceptScore, double maxShift, double repeatWeight, int device)
 * <- new MinHashSearch(...) impl/MinHashSearch.java:63-98 (flag defaults: main/MhapMain.java:67-125) */
JNIEXPORT jlong JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeCreate(
    JNIEnv* env, jclass cls, jint kmerSize, jint numHashes, jint orderedKmerSize, jint orderedSketchSize, jint numMinMatches,
    jint minStoreLength, jint minOlapLength, jdouble acceptScore, jdouble maxShift, jdouble repeatWeight, jint device) {
  mhap_params p;
  mhap_handle* h = NULL;
  char err[512];
  (void)cls;
  mhap_default_params(&p);
  p.kmer_size = kmerSize; p.num_hashes = numHashes; p.ordered_kmer_size = orderedKmerSize; p.ordered_sketch_size = orderedSketchSize;
  p.num_min_matches = numMinMatches; p.min_store_length = minStoreLength; p.min_olap_length = minOlapLength; p.device = device;
  p.threshold = acceptScore; p.max_shift = maxShift; p.repeat_weight = repeatWeight;
  err[0] = 0;
  if (mhap_create(&p, &h, err, sizeof err) != MHAP_OK) { throw_mhap(env, err); return 0; }
  return (jlong)(intptr_t)h;
}

/* void nativeDestroy(long handle) */
JNIEXPORT void JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeDestroy(JNIEnv* env, jclass cls, jlong handle) {
  (void)env; (void)cls;
  mhap_destroy(H(handle));
}

/* void nativeSetFilterFile(long handle, String path, double filterCutoff, double offset, int removeUnique, boolean noTf,
 *                          double range, boolean doReverseCompliment)
 * <- new FrequencyCounts(reader, filterCutoff, offset, removeUnique, noTf, numThreads, range, doRC) sketch/FrequencyCounts.java:63-229 */
JNIEXPORT void JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeSetFilterFile(
    JNIEnv* env, jclass cls, jlong handle, jstring path, jdouble filterCutoff, jdouble offset, jint removeUnique, jboolean noTf,
    jdouble range, jboolean doReverseCompliment) {
  const char* cpath = (*env)->GetStringUTFChars(env, path, NULL);
  int rc;
  (void)cls;
  if (!cpath) return;
  rc = mhap_set_filter_file(H(handle), cpath, filterCutoff, offset, removeUnique, noTf ? 1 : 0, range, doReverseCompliment ? 1 : 0, NULL, 0);
  (*env)->ReleaseStringUTFChars(env, path, cpath);
  if (rc == MHAP_E_IO) throw_mhap(env, "Could not parse k-mer filter file.");
  else check(env, H(handle), rc);
}

/* void nativeAddReads(long handle, byte[] bases, long[] offsets, int[] lengths, long[] ids, int n)
 * <- AbstractMatchSearch.addData + MinHashSearch.addSequence for a batch of reads (both strands are sketched and indexed on
 *    the GPU): impl/AbstractMatchSearch.java:67-117, impl/MinHashSearch.java:100-147, impl/SequenceSketchStreamer.java:123-156.
 *    bases: the upper-cased sequences back to back (FastaData.java:194), offsets/lengths per read, ids = SequenceId.getHeaderId(). */
JNIEXPORT void JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeAddReads(
    JNIEnv* env, jclass cls, jlong handle, jbyteArray bases, jlongArray offsets, jintArray lengths, jlongArray ids, jint n) {
  jbyte* b = (jbyte*)(*env)->GetPrimitiveArrayCritical(env, bases, NULL);
  jlong* o = (jlong*)(*env)->GetPrimitiveArrayCritical(env, offsets, NULL);
  jint* l = (jint*)(*env)->GetPrimitiveArrayCritical(env, lengths, NULL);
  jlong* i = (jlong*)(*env)->GetPrimitiveArrayCritical(env, ids, NULL);
  int rc = MHAP_E_NOMEM;
  (void)cls;
  if (b && o && l && i) rc = mhap_index_add_reads(H(handle), (const char*)b, (const int64_t*)o, (const int32_t*)l, (const int64_t*)i, (int64_t)n);
  if (i) (*env)->ReleasePrimitiveArrayCritical(env, ids, i, JNI_ABORT);
  if (l) (*env)->ReleasePrimitiveArrayCritical(env, lengths, l, JNI_ABORT);
  if (o) (*env)->ReleasePrimitiveArrayCritical(env, offsets, o, JNI_ABORT);
  if (b) (*env)->ReleasePrimitiveArrayCritical(env, bases, b, JNI_ABORT);
  check(env, H(handle), rc);
}

/* byte[] nativeFindMatchesSelf(long handle)
 * <- AbstractMatchSearch.findMatches() (every stored forward sequence against the index, toSelf = true):
 *    impl/AbstractMatchSearch.java:121-199, impl/MinHashSearch.java:150-251.  Returns packed mhap_record[]. */
JNIEXPORT jbyteArray JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeFindMatchesSelf(JNIEnv* env, jclass cls, jlong handle) {
  rec_buf buf = {NULL, 0, 0, 0};
  const int rc = mhap_find_matches_self(H(handle), 0, -1, collect_sink, &buf);
  (void)cls;
  if (rc != MHAP_OK && !buf.oom) { free(buf.p); check(env, H(handle), rc); return NULL; }
  return records_to_java(env, &buf);
}

/* byte[] nativeFindMatchesReads(long handle, byte[] bases, long[] offsets, int[] lengths, long[] ids, int n)
 * <- AbstractMatchSearch.findMatches(SequenceSketchStreamer) for a batch of query READS (forward strand only, toSelf = false):
 *    impl/AbstractMatchSearch.java:203-285.  The queries are sketched on the GPU. */
JNIEXPORT jbyteArray JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeFindMatchesReads(
    JNIEnv* env, jclass cls, jlong handle, jbyteArray bases, jlongArray offsets, jintArray lengths, jlongArray ids, jint n) {
  rec_buf buf = {NULL, 0, 0, 0};
  jbyte* b = (jbyte*)(*env)->GetPrimitiveArrayCritical(env, bases, NULL);
  jlong* o = (jlong*)(*env)->GetPrimitiveArrayCritical(env, offsets, NULL);
  jint* l = (jint*)(*env)->GetPrimitiveArrayCritical(env, lengths, NULL);
  jlong* i = (jlong*)(*env)->GetPrimitiveArrayCritical(env, ids, NULL);
  int rc = MHAP_E_NOMEM;
  (void)cls;
  if (b && o && l && i)
    rc = mhap_find_matches_reads(H(handle), (const char*)b, (const int64_t*)o, (const int32_t*)l, (const int64_t*)i, (int64_t)n, collect_sink, &buf);
  if (i) (*env)->ReleasePrimitiveArrayCritical(env, ids, i, JNI_ABORT);
  if (l) (*env)->ReleasePrimitiveArrayCritical(env, lengths, l, JNI_ABORT);
  if (o) (*env)->ReleasePrimitiveArrayCritical(env, offsets, o, JNI_ABORT);
  if (b) (*env)->ReleasePrimitiveArrayCritical(env, bases, b, JNI_ABORT);
  if (rc != MHAP_OK && !buf.oom) { free(buf.p); check(env, H(handle), rc); return NULL; }
  return records_to_java(env, &buf);
}

/* byte[] nativeFindMatchesSketches(long handle, long[] ids, int[] seqLength, int[] minHashes, int[] ordered, int[] orderedSize,
 *                                  int[] orderedSeqLength, int m)
 * <- the same driver when the streamer hands out SequenceSketch objects Java computed or read from a .dat file
 *    (impl/SequenceSketchStreamer.java:158-172,278-320): minHashes = m rows of MinHashSketch.getMinHashArray()
 *    (sketch/MinHashSketch.java:232), ordered = m rows of --ordered-sketch-size (hash, pos) pairs
 *    (sketch/BottomOverlapSketch.java:568-576), zero padded. */
JNIEXPORT jbyteArray JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeFindMatchesSketches(
    JNIEnv* env, jclass cls, jlong handle, jlongArray ids, jintArray seqLength, jintArray minHashes, jintArray ordered,
    jintArray orderedSize, jintArray orderedSeqLength, jint m) {
  rec_buf buf = {NULL, 0, 0, 0};
  jlong* i = (jlong*)(*env)->GetPrimitiveArrayCritical(env, ids, NULL);
  jint* sl = (jint*)(*env)->GetPrimitiveArrayCritical(env, seqLength, NULL);
  jint* mh = (jint*)(*env)->GetPrimitiveArrayCritical(env, minHashes, NULL);
  jint* od = (jint*)(*env)->GetPrimitiveArrayCritical(env, ordered, NULL);
  jint* os = (jint*)(*env)->GetPrimitiveArrayCritical(env, orderedSize, NULL);
  jint* ol = (jint*)(*env)->GetPrimitiveArrayCritical(env, orderedSeqLength, NULL);
  int rc = MHAP_E_NOMEM;
  (void)cls;
  if (i && sl && mh && od && os && ol)
    rc = mhap_find_matches_sketches(H(handle), (const int64_t*)i, (const int32_t*)sl, (const int32_t*)mh, (const int32_t*)od, (const int32_t*)os,
                                    (const int32_t*)ol, (int64_t)m, collect_sink, &buf);
  if (ol) (*env)->ReleasePrimitiveArrayCritical(env, orderedSeqLength, ol, JNI_ABORT);
  if (os) (*env)->ReleasePrimitiveArrayCritical(env, orderedSize, os, JNI_ABORT);
  if (od) (*env)->ReleasePrimitiveArrayCritical(env, ordered, od, JNI_ABORT);
  if (mh) (*env)->ReleasePrimitiveArrayCritical(env, minHashes, mh, JNI_ABORT);
  if (sl) (*env)->ReleasePrimitiveArrayCritical(env, seqLength, sl, JNI_ABORT);
  if (i) (*env)->ReleasePrimitiveArrayCritical(env, ids, i, JNI_ABORT);
  if (rc != MHAP_OK && !buf.oom) { free(buf.p); check(env, H(handle), rc); return NULL; }
  return records_to_java(env, &buf);
}

/* long[] nativeStats(long handle) -> {strandsIndexed, queriesSearched, candidatesCompared, matchesFound, tableElements}
 * <- size(), getNumberSequencesSearched(), getNumberSequencesFullyCompared(), getMatchesProcessed(), getNumberElementsProcessed()
 *    (impl/MinHashSearch.java:253-300, main/MhapMain.java:572-590) */
JNIEXPORT jlongArray JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeStats(JNIEnv* env, jclass cls, jlong handle) {
  mhap_stats st;
  jlong v[5];
  jlongArray out;
  (void)cls;
  if (check(env, H(handle), mhap_get_stats(H(handle), &st))) return NULL;
  v[0] = st.strands_indexed; v[1] = st.queries_searched; v[2] = st.candidates_compared; v[3] = st.matches_found; v[4] = st.table_elements;
  out = (*env)->NewLongArray(env, 5);
  if (out) (*env)->SetLongArrayRegion(env, out, 0, 5, v);
  return out;
}
