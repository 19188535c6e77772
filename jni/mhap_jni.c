/* mhap_jni.c — JNI shim between MHAP's Java host and libmhaphip.so (include/mhap_hip.h).
 *
 * Binds the native methods of edu.umd.marbl.mhap.impl.HipMinHashSearch (jni/HipMinHashSearch.java), the GPU-backed
 * replacement of MinHashSearch behind the reference's only operator seam, AbstractMatchSearch
 * (src/main/java/edu/umd/marbl/mhap/impl/AbstractMatchSearch.java:47,67-117,121-199,203-285).
 *
 * The engine behind a Java object is a GROUP of ranks (mhap_group_*): one rank per GPU the host names, a group of one for a
 * single GPU.  The reads of every addData batch are dealt round-robin over the ranks, every rank sketches and indexes its share,
 * and a search gathers the forward query sketches of all ranks over xGMI inside the library — the Java host never sees the
 * sharding (J/impl/AbstractMatchSearch.java has one index and a thread pool).
 *
 * Not compiled in this repository's image (no JDK: no jni.h).  Build where a JDK is present:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../include mhap_jni.c \
 *       -L../mhap_amd/lib -lmhaphip -Wl,-rpath,'$ORIGIN' -o libmhapjni.so
 * tests/test_host_logic.py checks that every native method of the Java class has its Java_... function here, that the
 * argument counts agree and that every mhap_* function called exists in the header; it also compiles this file against a
 * minimal stand-in jni.h for syntax.
 *
 * Conventions
 *  - no JNI global references are held across calls;
 *  - input arrays are COPIED out with Get<Type>ArrayRegion before the library is called (no GetPrimitiveArrayCritical region
 *    is held across a multi-second GPU call: that would stall the JVM's collector);
 *  - overlap records never cross as one array: a search parks them natively (64-byte mhap_record, little-endian, the layout of
 *    include/mhap_hip.h) and returns their COUNT; the Java side then takes them out in bounded chunks (nativeTakeRecords,
 *    at most 1 << 20 records = 64 MB per byte[]) and hands each chunk to AbstractMatchSearch.outputResults — no 2 GB array limit,
 *    no silent truncation, the Java heap holds one chunk at a time;
 *  - the self search STREAMS: HipMinHashSearch.findMatches() runs nativeFindMatchesSelf on a worker thread and takes records on the
 *    calling thread while the GPUs are still searching (nativeTakeRecords blocks until a chunk is there or the search is over), so
 *    outputResults runs during the search like the reference's pool does every 20 000 matches (AbstractMatchSearch.java:55,158) and
 *    at most PARK_CAP records (256 MB) wait natively: the library's sink blocks until Java has taken some.  A C5-scale search
 *    (10^8 records) no longer sits in native memory until it ends;
 *  - every library error becomes an unchecked MhapRuntimeException carrying mhap_last_error() / mhap_group_last_error().
 */
#include <jni.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mhap_hip.h"

#define MHAP_EXC "edu/umd/marbl/mhap/impl/MhapRuntimeException"
#define MAX_TAKE (1 << 20)
#define PARK_CAP (4 << 20)   /* records parked natively before the sink waits for Java (streaming searches only) */

/* parked records: a list of blocks the sink appends to (the library calls the sink one batch at a time, also with several ranks) */
typedef struct rec_block { struct rec_block* next; int64_t n, taken; mhap_record recs[1]; } rec_block;
typedef struct {
  mhap_group* g; rec_block *head, *tail; int64_t parked, total; int oom; int32_t H, S;
  pthread_mutex_t mu; pthread_cond_t data, space;   /* parked records: the library's sink thread appends, a Java thread takes */
  int streaming, done, abandoned;                   /* streaming: bounded parking; done: no search is running; abandoned: the taker gave up */
} engine;

static engine* E(jlong h) { return (engine*)(intptr_t)h; }

static void throw_mhap(JNIEnv* env, const char* msg) {
  jclass c = (*env)->FindClass(env, MHAP_EXC);
  if (c) (*env)->ThrowNew(env, c, msg ? msg : "libmhaphip error");
}

static int check(JNIEnv* env, engine* e, int rc) {
  if (rc == MHAP_OK) return 0;
  throw_mhap(env, e && e->g ? mhap_group_last_error(e->g) : "libmhaphip error");
  return 1;
}

static void drop_records(engine* e) {
  pthread_mutex_lock(&e->mu);
  while (e->head) { rec_block* b = e->head; e->head = b->next; free(b); }
  e->tail = NULL; e->parked = 0; e->oom = 0;
  pthread_cond_broadcast(&e->space);
  pthread_mutex_unlock(&e->mu);
}

static int park_sink(const mhap_record* recs, int64_t n, void* user) {
  engine* e = (engine*)user;
  rec_block* b;
  if (n <= 0) return 0;
  b = (rec_block*)malloc(sizeof(rec_block) + (size_t)(n - 1) * sizeof(mhap_record));
  if (!b) { e->oom = 1; return 1; }
  b->next = NULL; b->n = n; b->taken = 0;
  memcpy(b->recs, recs, (size_t)n * sizeof(mhap_record));
  pthread_mutex_lock(&e->mu);
  while (e->streaming && e->parked > PARK_CAP && !e->abandoned) pthread_cond_wait(&e->space, &e->mu);   /* Java is behind: wait for a take */
  if (e->abandoned) { pthread_mutex_unlock(&e->mu); free(b); return 1; }                                  /* nobody will take them: abort the search */
  if (e->tail) e->tail->next = b; else e->head = b;
  e->tail = b;
  e->parked += n; e->total += n;
  pthread_cond_broadcast(&e->data);
  pthread_mutex_unlock(&e->mu);
  return 0;
}

static void search_begins(engine* e) { pthread_mutex_lock(&e->mu); e->done = 0; e->total = 0; pthread_mutex_unlock(&e->mu); }
static void search_ends(engine* e) { pthread_mutex_lock(&e->mu); e->done = 1; pthread_cond_broadcast(&e->data); pthread_mutex_unlock(&e->mu); }

/* a search finished with code rc: its records are parked; returns their count or throws */
static jlong finish_search(JNIEnv* env, engine* e, int rc) {
  search_ends(e);
  if (e->oom) { drop_records(e); throw_mhap(env, "out of memory while collecting overlap records"); return -1; }
  if (rc != MHAP_OK) { drop_records(e); check(env, e, rc); return -1; }
  return (jlong)e->total;     /* records this search delivered (some may have been taken already) */
}

/* copies of the four arrays that describe a batch of reads */
typedef struct { char* bases; int64_t* offsets; int32_t* lengths; int64_t* ids; } read_batch;
static void free_batch(read_batch* b) { free(b->bases); free(b->offsets); free(b->lengths); free(b->ids); }
static int copy_batch(JNIEnv* env, read_batch* b, jbyteArray bases, jlongArray offsets, jintArray lengths, jlongArray ids, jint n) {
  const jsize nb = (*env)->GetArrayLength(env, bases);
  memset(b, 0, sizeof *b);
  if (n < 0 || (*env)->GetArrayLength(env, offsets) < n || (*env)->GetArrayLength(env, lengths) < n || (*env)->GetArrayLength(env, ids) < n) {
    throw_mhap(env, "read batch arrays are shorter than the read count");
    return 1;
  }
  b->bases = (char*)malloc((size_t)nb + 1);
  b->offsets = (int64_t*)malloc(((size_t)n + 1) * sizeof(int64_t));
  b->lengths = (int32_t*)malloc(((size_t)n + 1) * sizeof(int32_t));
  b->ids = (int64_t*)malloc(((size_t)n + 1) * sizeof(int64_t));
  if (!b->bases || !b->offsets || !b->lengths || !b->ids) { free_batch(b); throw_mhap(env, "out of memory while copying a read batch"); return 1; }
  (*env)->GetByteArrayRegion(env, bases, 0, nb, (jbyte*)b->bases);
  if (!(*env)->ExceptionCheck(env)) (*env)->GetLongArrayRegion(env, offsets, 0, n, (jlong*)b->offsets);
  if (!(*env)->ExceptionCheck(env)) (*env)->GetIntArrayRegion(env, lengths, 0, n, (jint*)b->lengths);
  if (!(*env)->ExceptionCheck(env)) (*env)->GetLongArrayRegion(env, ids, 0, n, (jlong*)b->ids);
  if ((*env)->ExceptionCheck(env)) { free_batch(b); return 1; }   /* ArrayIndexOutOfBoundsException is pending */
  {
    jint i;   /* every read must lie inside the bases array: the library trusts offsets and lengths */
    for (i = 0; i < n; i++)
      if (b->lengths[i] < 0 || b->offsets[i] < 0 || b->offsets[i] > (int64_t)nb || (int64_t)b->lengths[i] > (int64_t)nb - b->offsets[i]) {
        free_batch(b);
        throw_mhap(env, "a read of the batch lies outside its bases array");
        return 1;
      }
  }
  return 0;
}

/* long nativeCreate(int kmerSize, int numHashes, int orderedKmerSize, int orderedSketchSize, int numMinMatches,
 *                   int minStoreLength, int minOlapLength, double acceptScore, double maxShift, double repeatWeight, int[] devices)
 * <- new MinHashSearch(...) impl/MinHashSearch.java:63-98 (flag defaults: main/MhapMain.java:67-125); devices: one HIP device
 *    ordinal per rank (one entry = one GPU) */
JNIEXPORT jlong JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeCreate(
    JNIEnv* env, jclass cls, jint kmerSize, jint numHashes, jint orderedKmerSize, jint orderedSketchSize, jint numMinMatches,
    jint minStoreLength, jint minOlapLength, jdouble acceptScore, jdouble maxShift, jdouble repeatWeight, jintArray devices) {
  mhap_params p;
  engine* e;
  char err[512];
  int32_t devs[64];
  const jsize nd = (*env)->GetArrayLength(env, devices);
  (void)cls;
  if (nd < 1 || nd > 64) { throw_mhap(env, "between 1 and 64 devices, please"); return 0; }
  (*env)->GetIntArrayRegion(env, devices, 0, nd, (jint*)devs);
  mhap_default_params(&p);
  p.kmer_size = kmerSize; p.num_hashes = numHashes; p.ordered_kmer_size = orderedKmerSize; p.ordered_sketch_size = orderedSketchSize;
  p.num_min_matches = numMinMatches; p.min_store_length = minStoreLength; p.min_olap_length = minOlapLength; p.device = devs[0];
  p.threshold = acceptScore; p.max_shift = maxShift; p.repeat_weight = repeatWeight;
  e = (engine*)calloc(1, sizeof(engine));
  if (!e) { throw_mhap(env, "out of memory"); return 0; }
  err[0] = 0;
  if (mhap_group_create(&p, devs, (int32_t)nd, &e->g, err, sizeof err) != MHAP_OK) { free(e); throw_mhap(env, err); return 0; }
  e->H = numHashes; e->S = orderedSketchSize; e->done = 1;
  pthread_mutex_init(&e->mu, NULL); pthread_cond_init(&e->data, NULL); pthread_cond_init(&e->space, NULL);
  return (jlong)(intptr_t)e;
}

/* void nativeDestroy(long engine) */
JNIEXPORT void JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeDestroy(JNIEnv* env, jclass cls, jlong handle) {
  engine* e = E(handle);
  (void)env; (void)cls;
  if (!e) return;
  drop_records(e);
  mhap_group_destroy(e->g);
  pthread_cond_destroy(&e->data); pthread_cond_destroy(&e->space); pthread_mutex_destroy(&e->mu);
  free(e);
}

/* void nativeSetFilterFile(long engine, String path, double filterCutoff, double offset, int removeUnique, boolean noTf,
 *                          double range, boolean doReverseCompliment)
 * <- new FrequencyCounts(reader, filterCutoff, offset, removeUnique, noTf, numThreads, range, doRC) sketch/FrequencyCounts.java:63-229;
 *    every rank applies the filter to the reads it sketches */
JNIEXPORT void JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeSetFilterFile(
    JNIEnv* env, jclass cls, jlong handle, jstring path, jdouble filterCutoff, jdouble offset, jint removeUnique, jboolean noTf,
    jdouble range, jboolean doReverseCompliment) {
  engine* e = E(handle);
  const char* cpath = (*env)->GetStringUTFChars(env, path, NULL);
  int rc = MHAP_OK, r;
  mhap_handle* bad = NULL;
  (void)cls;
  if (!cpath) return;
  for (r = 0; r < mhap_group_size(e->g) && rc == MHAP_OK; r++) {
    bad = mhap_group_rank(e->g, r);
    rc = mhap_set_filter_file(bad, cpath, filterCutoff, offset, removeUnique, noTf ? 1 : 0, range, doReverseCompliment ? 1 : 0, NULL, 0);
  }
  (*env)->ReleaseStringUTFChars(env, path, cpath);
  if (rc == MHAP_E_IO) throw_mhap(env, "Could not parse k-mer filter file.");
  else if (rc != MHAP_OK) throw_mhap(env, mhap_last_error(bad));
}

/* void nativeAddReads(long engine, byte[] bases, long[] offsets, int[] lengths, long[] ids, int n)
 * <- AbstractMatchSearch.addData + MinHashSearch.addSequence for a batch of reads (both strands are sketched and indexed on
 *    the GPUs): impl/AbstractMatchSearch.java:67-117, impl/MinHashSearch.java:100-147, impl/SequenceSketchStreamer.java:123-156.
 *    bases: the upper-cased sequences back to back (FastaData.java:194), offsets/lengths per read, ids = SequenceId.getHeaderId(). */
JNIEXPORT void JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeAddReads(
    JNIEnv* env, jclass cls, jlong handle, jbyteArray bases, jlongArray offsets, jintArray lengths, jlongArray ids, jint n) {
  engine* e = E(handle);
  read_batch b;
  (void)cls;
  if (copy_batch(env, &b, bases, offsets, lengths, ids, n)) return;
  check(env, e, mhap_group_add_reads(e->g, b.bases, b.offsets, b.lengths, b.ids, (int64_t)n));
  free_batch(&b);
}

/* long nativeFindMatchesSelf(long engine) -> number of records parked (take them with nativeTakeRecords)
 * <- AbstractMatchSearch.findMatches() (every stored forward sequence against the index, toSelf = true):
 *    impl/AbstractMatchSearch.java:121-199, impl/MinHashSearch.java:150-251 */
JNIEXPORT jlong JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeFindMatchesSelf(JNIEnv* env, jclass cls, jlong handle) {
  engine* e = E(handle);
  (void)cls;
  search_begins(e);
  return finish_search(env, e, mhap_group_find_matches_self(e->g, park_sink, e));
}

/* long nativeFindMatchesReads(long engine, byte[] bases, long[] offsets, int[] lengths, long[] ids, int n) -> records parked
 * <- AbstractMatchSearch.findMatches(SequenceSketchStreamer) for a batch of query READS (forward strand only, toSelf = false):
 *    impl/AbstractMatchSearch.java:203-285.  The queries are sketched on the GPUs (dealt over the ranks like the stored reads). */
JNIEXPORT jlong JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeFindMatchesReads(
    JNIEnv* env, jclass cls, jlong handle, jbyteArray bases, jlongArray offsets, jintArray lengths, jlongArray ids, jint n) {
  engine* e = E(handle);
  read_batch b;
  int rc;
  (void)cls;
  if (copy_batch(env, &b, bases, offsets, lengths, ids, n)) return -1;
  search_begins(e);
  rc = mhap_group_find_matches_reads(e->g, b.bases, b.offsets, b.lengths, b.ids, (int64_t)n, park_sink, e);
  free_batch(&b);
  return finish_search(env, e, rc);
}

/* long nativeFindMatchesSketches(long engine, long[] ids, int[] seqLength, int[] minHashes, int[] ordered, int[] orderedSize,
 *                                int[] orderedSeqLength, int m) -> records parked
 * <- the same driver when the streamer hands out SequenceSketch objects Java computed or read from a .dat file
 *    (impl/SequenceSketchStreamer.java:158-172,278-320): minHashes = m rows of MinHashSketch.getMinHashArray()
 *    (sketch/MinHashSketch.java:232), ordered = m rows of --ordered-sketch-size (hash, pos) pairs
 *    (sketch/BottomOverlapSketch.java:568-576), zero padded.  toSelf = false has no id rule, so every rank simply searches the
 *    sketches against its own shard: the union over the ranks is the whole index's answer. */
JNIEXPORT jlong JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeFindMatchesSketches(
    JNIEnv* env, jclass cls, jlong handle, jlongArray ids, jintArray seqLength, jintArray minHashes, jintArray ordered,
    jintArray orderedSize, jintArray orderedSeqLength, jint m) {
  engine* e = E(handle);
  const jsize nmh = (*env)->GetArrayLength(env, minHashes), nod = (*env)->GetArrayLength(env, ordered);
  int64_t *i; int32_t *sl, *mh, *od, *os, *ol;
  int rc = MHAP_E_NOMEM, r;
  mhap_handle* bad = NULL;
  (void)cls;
  /* the library reads m rows of --num-hashes ints and m rows of --ordered-sketch-size (hash, pos) pairs: short arrays are an error
   * here, not an over-read there */
  if (m < 0 || (*env)->GetArrayLength(env, ids) < m || (*env)->GetArrayLength(env, seqLength) < m || (*env)->GetArrayLength(env, orderedSize) < m ||
      (*env)->GetArrayLength(env, orderedSeqLength) < m || (int64_t)nmh < (int64_t)m * e->H || (int64_t)nod < (int64_t)m * e->S * 2) {
    throw_mhap(env, "query sketch arrays are shorter than the sketch count asks for");
    return -1;
  }
  i = (int64_t*)malloc(((size_t)m + 1) * 8);
  sl = (int32_t*)malloc(((size_t)m + 1) * 4);
  mh = (int32_t*)malloc(((size_t)nmh + 1) * 4);
  od = (int32_t*)malloc(((size_t)nod + 1) * 4);
  os = (int32_t*)malloc(((size_t)m + 1) * 4);
  ol = (int32_t*)malloc(((size_t)m + 1) * 4);
  if (i && sl && mh && od && os && ol) {
    (*env)->GetLongArrayRegion(env, ids, 0, m, (jlong*)i);
    if (!(*env)->ExceptionCheck(env)) (*env)->GetIntArrayRegion(env, seqLength, 0, m, (jint*)sl);
    if (!(*env)->ExceptionCheck(env)) (*env)->GetIntArrayRegion(env, minHashes, 0, nmh, (jint*)mh);
    if (!(*env)->ExceptionCheck(env)) (*env)->GetIntArrayRegion(env, ordered, 0, nod, (jint*)od);
    if (!(*env)->ExceptionCheck(env)) (*env)->GetIntArrayRegion(env, orderedSize, 0, m, (jint*)os);
    if (!(*env)->ExceptionCheck(env)) (*env)->GetIntArrayRegion(env, orderedSeqLength, 0, m, (jint*)ol);
    if ((*env)->ExceptionCheck(env)) { free(i); free(sl); free(mh); free(od); free(os); free(ol); return -1; }
    search_begins(e);
    rc = MHAP_OK;
    for (r = 0; r < mhap_group_size(e->g) && rc == MHAP_OK && !e->oom; r++) {
      bad = mhap_group_rank(e->g, r);
      rc = mhap_find_matches_sketches(bad, i, sl, mh, od, os, ol, (int64_t)m, park_sink, e);
    }
  }
  free(i); free(sl); free(mh); free(od); free(os); free(ol);
  if (rc != MHAP_OK && !e->oom) { search_ends(e); drop_records(e); throw_mhap(env, bad ? mhap_last_error(bad) : "out of memory while copying query sketches"); return -1; }
  return finish_search(env, e, MHAP_OK);
}

/* byte[] nativeTakeRecords(long engine, int maxRecords) -> the next chunk of parked records (packed mhap_record[], at most
 * min(maxRecords, 1 << 20) of them); BLOCKS while a search is running and nothing is parked; null once the search is over and
 * nothing is left */
JNIEXPORT jbyteArray JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeTakeRecords(JNIEnv* env, jclass cls, jlong handle, jint maxRecords) {
  engine* e = E(handle);
  int64_t want = maxRecords < 1 ? 1 : (maxRecords > MAX_TAKE ? MAX_TAKE : maxRecords), got = 0;
  jbyteArray out;
  rec_block *first = NULL, *last = NULL, *b;
  (void)cls;
  /* unlink up to `want` records under the lock, copy them into the Java array outside it */
  pthread_mutex_lock(&e->mu);
  while (e->parked <= 0 && !e->done) pthread_cond_wait(&e->data, &e->mu);
  if (e->parked <= 0) { pthread_mutex_unlock(&e->mu); return NULL; }
  if (want > e->parked) want = e->parked;
  while (got < want && e->head) {
    int64_t k;
    b = e->head;
    k = b->n - b->taken;
    if (k <= want - got) {                         /* the whole rest of this block */
      e->head = b->next; if (!e->head) e->tail = NULL;
      b->next = NULL;
      if (last) last->next = b; else first = b;
      last = b; got += k;
    } else {                                        /* part of it: copied here, the block stays */
      rec_block* part = (rec_block*)malloc(sizeof(rec_block) + (size_t)(want - got - 1) * sizeof(mhap_record));
      if (!part) break;
      part->next = NULL; part->n = want - got; part->taken = 0;
      memcpy(part->recs, b->recs + b->taken, (size_t)(want - got) * sizeof(mhap_record));
      b->taken += want - got;
      if (last) last->next = part; else first = part;
      last = part; got = want;
    }
  }
  e->parked -= got;
  pthread_cond_broadcast(&e->space);
  pthread_mutex_unlock(&e->mu);
  if (got == 0) {   /* records are parked but the split block could not be allocated: raise, a zero-length array would make Java's take() loop spin */
    jclass oom = (*env)->FindClass(env, "java/lang/OutOfMemoryError");
    if (oom) (*env)->ThrowNew(env, oom, "mhap_jni: cannot allocate a record block");
    return NULL;
  }
  out = (*env)->NewByteArray(env, (jsize)(got * (int64_t)sizeof(mhap_record)));   /* <= 64 MB: fits a jsize */
  got = 0;
  while (first) {
    b = first; first = b->next;
    if (out) (*env)->SetByteArrayRegion(env, out, (jsize)(got * (int64_t)sizeof(mhap_record)), (jsize)((b->n - b->taken) * (int64_t)sizeof(mhap_record)),
                                        (const jbyte*)(b->recs + b->taken));
    got += b->n - b->taken;
    free(b);
  }
  return out;                                                                      /* NULL: OutOfMemoryError is pending */
}

/* void nativeSetStreaming(long engine, boolean on) -> bounded parking (the sink waits for nativeTakeRecords) for the searches that follow;
 * turning it off also releases a sink that is waiting */
JNIEXPORT void JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeSetStreaming(JNIEnv* env, jclass cls, jlong handle, jboolean on) {
  engine* e = E(handle);
  (void)env; (void)cls;
  pthread_mutex_lock(&e->mu);
  e->streaming = on ? 1 : 0; e->abandoned = 0;
  if (on) e->done = 0;      /* a search is about to start on another thread: a take that comes first must wait for it, not see "over" */
  pthread_cond_broadcast(&e->space);
  pthread_mutex_unlock(&e->mu);
}

/* void nativeAbandon(long engine) -> the taking thread gives up (an exception on the Java side): a sink that waits returns "abort", the
 * running search ends with an error instead of waiting for ever */
JNIEXPORT void JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeAbandon(JNIEnv* env, jclass cls, jlong handle) {
  engine* e = E(handle);
  (void)env; (void)cls;
  pthread_mutex_lock(&e->mu);
  e->abandoned = 1;
  pthread_cond_broadcast(&e->space);
  pthread_mutex_unlock(&e->mu);
}

/* long[] nativeStats(long engine) -> {strandsIndexed, queriesSearched, candidatesCompared, matchesFound, tableElements}, summed over the ranks
 * <- size(), getNumberSequencesSearched(), getNumberSequencesFullyCompared(), getMatchesProcessed(), getNumberElementsProcessed()
 *    (impl/MinHashSearch.java:253-300, main/MhapMain.java:572-590) */
JNIEXPORT jlongArray JNICALL Java_edu_umd_marbl_mhap_impl_HipMinHashSearch_nativeStats(JNIEnv* env, jclass cls, jlong handle) {
  engine* e = E(handle);
  mhap_stats st;
  jlong v[5];
  jlongArray out;
  (void)cls;
  if (check(env, e, mhap_group_get_stats(e->g, &st))) return NULL;
  v[0] = st.strands_indexed; v[1] = st.queries_searched; v[2] = st.candidates_compared; v[3] = st.matches_found; v[4] = st.table_elements;
  out = (*env)->NewLongArray(env, 5);
  if (out) (*env)->SetLongArrayRegion(env, out, 0, 5, v);
  return out;
}
