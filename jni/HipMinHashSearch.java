/*
 * HipMinHashSearch — MHAP's match search backed by libmhaphip.so (MI355X), behind the reference's own operator seam.
 *
 * Drop this file into src/main/java/edu/umd/marbl/mhap/impl/ of marbl/MHAP 2.1.3 (it uses package-private access to
 * MatchResult's constructor, exactly like MinHashSearch does) and build jni/mhap_jni.c into libmhapjni.so.  It extends
 * AbstractMatchSearch (impl/AbstractMatchSearch.java:47) and replaces MinHashSearch (impl/MinHashSearch.java): the three
 * DRIVERS are overridden — adding the data, findMatches() and findMatches(SequenceSketchStreamer) — because the GPU works
 * on batches; the per-read abstract methods (addSequence, findMatches(SequenceSketch, boolean)) are implemented on top of
 * the batch calls for completeness.  Records are still printed by AbstractMatchSearch.outputResults, i.e. by Java's own
 * String.format in MatchResult.toString (impl/MatchResult.java:98-113).
 *
 * Wiring in MhapMain (main/MhapMain.java:453-541): replace
 *     SequenceSketchStreamer seqStreamer = getSequenceHashStreamer(this.inFile, seqNumberProcessed);
 *     MinHashSearch hashSearch = getMatchSearch(seqStreamer);
 *     seqNumberProcessed += seqStreamer.getNumberProcessed()/2;
 * by
 *     HipMinHashSearch hashSearch = new HipMinHashSearch(new FastaData(this.inFile, seqNumberProcessed), this.kmerSize, this.numHashes,
 *         this.orderedKmerSize, this.orderedSketchSize, this.numMinMatches, this.numThreads, false, this.minStoreLength, this.minOlapLength,
 *         this.maxShift, this.acceptScore, this.repeatWeight, new int[] {0});     // one HIP device ordinal per GPU: {0,1,...,7} shards the index over a node
 *     if (this.filterFile != null) hashSearch.setFilterFile(this.filterFile, this.filterThreshold, offset, this.supressNoise, this.noTf,
 *         this.repeatIdfScale, this.doReverseCompliment);      // before the reads are added
 *     hashSearch.addData();
 *     seqNumberProcessed += hashSearch.size()/2;
 * and keep hashSearch.findMatches() / hashSearch.findMatches(streamer) as they are.
 *
 * Not compiled in this repository (no JDK in the image); tests/test_host_logic.py checks it against jni/mhap_jni.c.
 */
package edu.umd.marbl.mhap.impl;

import java.io.IOException;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.util.ArrayList;
import java.util.HashMap;
import java.util.List;
import java.util.Map;

import edu.umd.marbl.mhap.sketch.BottomOverlapSketch;
import edu.umd.marbl.mhap.utils.ReadBuffer;

public final class HipMinHashSearch extends AbstractMatchSearch
{
	static
	{
		System.loadLibrary("mhapjni");
	}

	/** reads handed to the GPUs per native call (bases are concatenated into one byte[], which the native side copies) */
	private final static int READS_PER_BATCH = 65536;
	private final static long BASES_PER_BATCH = 1L << 28;
	private final static int RECORD_BYTES = 64;
	/** overlap records taken out of the native side per call: the Java heap holds one such chunk (<= 64 MB) at a time */
	private final static int RECORDS_PER_TAKE = 1 << 20;

	private final long handle;
	private final FastaData data;
	private final int numHashes;
	private final int orderedSketchSize;
	private final boolean storeResults;
	private final Map<Long, String> fullIds;
	private final List<SequenceId> storedForwardIds;
	private long matchesProcessed;
	private long sequencesSearched;
	private boolean closed;

	private static native long nativeCreate(int kmerSize, int numHashes, int orderedKmerSize, int orderedSketchSize, int numMinMatches,
			int minStoreLength, int minOlapLength, double acceptScore, double maxShift, double repeatWeight, int[] devices);

	private static native void nativeDestroy(long handle);

	private static native void nativeSetFilterFile(long handle, String path, double filterCutoff, double offset, int removeUnique,
			boolean noTf, double range, boolean doReverseCompliment);

	private static native void nativeAddReads(long handle, byte[] bases, long[] offsets, int[] lengths, long[] ids, int n);

	/** the three searches park their records natively and return how many there are; nativeTakeRecords hands them out in chunks */
	private static native long nativeFindMatchesSelf(long handle);

	private static native long nativeFindMatchesReads(long handle, byte[] bases, long[] offsets, int[] lengths, long[] ids, int n);

	private static native long nativeFindMatchesSketches(long handle, long[] ids, int[] seqLength, int[] minHashes, int[] ordered,
			int[] orderedSize, int[] orderedSeqLength, int m);

	private static native byte[] nativeTakeRecords(long handle, int maxRecords);

	private static native void nativeSetStreaming(long handle, boolean on);

	private static native void nativeAbandon(long handle);

	private static native long[] nativeStats(long handle);

	public HipMinHashSearch(FastaData data, int kmerSize, int numHashes, int orderedKmerSize, int orderedSketchSize, int numMinMatches,
			int numThreads, boolean storeResults, int minStoreLength, int minOlapLength, double maxShift, double acceptScore,
			double repeatWeight, int[] devices)
	{
		super(numThreads, storeResults);
		this.data = data;
		this.numHashes = numHashes;
		this.orderedSketchSize = orderedSketchSize;
		this.storeResults = storeResults;
		this.fullIds = new HashMap<Long, String>();
		this.storedForwardIds = new ArrayList<SequenceId>();
		this.handle = nativeCreate(kmerSize, numHashes, orderedKmerSize, orderedSketchSize, numMinMatches, minStoreLength, minOlapLength,
				acceptScore, maxShift, repeatWeight, devices);
	}

	/** new FrequencyCounts(...) for the GPU path (sketch/FrequencyCounts.java:63-229); call before addData(). */
	public void setFilterFile(String path, double filterCutoff, double offset, int removeUnique, boolean noTf, double range,
			boolean doReverseCompliment)
	{
		nativeSetFilterFile(this.handle, path, filterCutoff, offset, removeUnique, noTf, range, doReverseCompliment);
	}

	/** One batch of reads as the native side wants them. */
	private final static class ReadBatch
	{
		final ArrayList<Sequence> reads = new ArrayList<Sequence>();
		long bases = 0;

		boolean full()
		{
			return this.reads.size() >= READS_PER_BATCH || this.bases >= BASES_PER_BATCH;
		}

		void add(Sequence seq)
		{
			this.reads.add(seq);
			this.bases += seq.length();
		}

		byte[] baseBytes()
		{
			byte[] out = new byte[(int) this.bases];
			int at = 0;
			for (Sequence seq : this.reads)
			{
				String s = seq.getSquenceString();
				// FASTA sequences are ASCII (FastaData.java:166,194): one byte per Java char
				for (int i = 0; i < s.length(); i++)
					out[at++] = (byte) s.charAt(i);
			}
			return out;
		}

		long[] offsets()
		{
			long[] out = new long[this.reads.size()];
			long at = 0;
			for (int i = 0; i < out.length; i++)
			{
				out[i] = at;
				at += this.reads.get(i).length();
			}
			return out;
		}

		int[] lengths()
		{
			int[] out = new int[this.reads.size()];
			for (int i = 0; i < out.length; i++)
				out[i] = this.reads.get(i).length();
			return out;
		}

		long[] ids()
		{
			long[] out = new long[this.reads.size()];
			for (int i = 0; i < out.length; i++)
				out[i] = this.reads.get(i).getId().getHeaderId();
			return out;
		}
	}

	private void rememberIds(ReadBatch batch, boolean stored)
	{
		for (Sequence seq : batch.reads)
		{
			SequenceId id = seq.getId();
			if (SequenceId.STORE_FULL_ID)
				this.fullIds.put(id.getHeaderId(), id.getHeader());
			if (stored)
				this.storedForwardIds.add(id);
		}
	}

	/**
	 * The driver AbstractMatchSearch.addData (impl/AbstractMatchSearch.java:67-117) + SequenceSketchStreamer's fwd/rc
	 * sketching (impl/SequenceSketchStreamer.java:123-156): every read of the FASTA file goes to the GPU, which sketches both
	 * strands and stores them.  Reads below --min-olap-length are passed too: they consume an id and are skipped natively.
	 */
	public void addData() throws IOException
	{
		ReadBatch batch = new ReadBatch();
		Sequence seq = this.data.dequeue();
		while (seq != null)
		{
			batch.add(seq);
			if (batch.full())
			{
				flushAdd(batch);
				batch = new ReadBatch();
			}
			seq = this.data.dequeue();
		}
		flushAdd(batch);
		System.err.println("Stored " + size() + " sequences in the index.");
	}

	private void flushAdd(ReadBatch batch)
	{
		if (batch.reads.isEmpty())
			return;
		rememberIds(batch, true);
		nativeAddReads(this.handle, batch.baseBytes(), batch.offsets(), batch.lengths(), batch.ids(), batch.reads.size());
	}

	/** per-read seam kept for completeness: a single stored sketch cannot be injected, reads are added through addData(). */
	@Override
	protected boolean addSequence(SequenceSketch seqHashes)
	{
		throw new MhapRuntimeException("HipMinHashSearch stores reads in batches (addData), not sketch by sketch.");
	}

	private SequenceId idOf(long headerId, boolean isFwd)
	{
		String str = this.fullIds.get(headerId);
		return new SequenceId(headerId, isFwd, str);
	}

	/** packed mhap_record[] (include/mhap_hip.h: little-endian, 64 bytes each) -> MatchResult list */
	private ArrayList<MatchResult> decode(byte[] packed)
	{
		ArrayList<MatchResult> out = new ArrayList<MatchResult>(packed.length / RECORD_BYTES);
		ByteBuffer bb = ByteBuffer.wrap(packed).order(ByteOrder.LITTLE_ENDIAN);
		for (int at = 0; at + RECORD_BYTES <= packed.length; at += RECORD_BYTES)
		{
			long fromId = bb.getLong(at), toId = bb.getLong(at + 8);
			double score = bb.getDouble(at + 16), raw = bb.getDouble(at + 24);
			int a1 = bb.getInt(at + 32), a2 = bb.getInt(at + 36), alen = bb.getInt(at + 40);
			int b1 = bb.getInt(at + 44), b2 = bb.getInt(at + 48), blen = bb.getInt(at + 52);
			boolean toRc = bb.getInt(at + 56) != 0;
			// the library already flipped b for a reverse-complement match (MatchResult.java:56-57); MatchResult's constructor flips
			// again, so hand it the unflipped interval
			int ob1 = toRc ? blen - b2 - 1 : b1, ob2 = toRc ? blen - b1 - 1 : b2;
			out.add(new MatchResult(idOf(fromId, true), idOf(toId, !toRc), new OverlapInfo(score, raw, a1, a2, ob1, ob2), alen, blen));
		}
		return out;
	}

	/**
	 * Takes the records a native search parked out in chunks of at most RECORDS_PER_TAKE and hands every chunk on the way the
	 * reference's drivers do (impl/AbstractMatchSearch.java:155-170,316-338): printed through outputResults in pieces of
	 * NUM_ELEMENTS_PER_OUTPUT, or collected when storeResults is set.
	 */
	private void deliver(long parked, ArrayList<MatchResult> combined)
	{
		long taken = take(combined);
		if (taken != parked)
			throw new MhapRuntimeException("Overlap records lost between the library and Java: " + taken + " of " + parked + ".");
	}

	/** takes chunks until the library has none left (during a streaming search: until the search is over); returns the records taken */
	private long take(ArrayList<MatchResult> combined)
	{
		long taken = 0;
		byte[] chunk = nativeTakeRecords(this.handle, RECORDS_PER_TAKE);
		while (chunk != null)
		{
			ArrayList<MatchResult> matches = decode(chunk);
			taken += matches.size();
			this.matchesProcessed += matches.size();
			if (this.storeResults)
				combined.addAll(matches);
			else
				for (int from = 0; from < matches.size(); from += NUM_ELEMENTS_PER_OUTPUT)
					outputResults(matches.subList(from, Math.min(matches.size(), from + NUM_ELEMENTS_PER_OUTPUT)));
			chunk = nativeTakeRecords(this.handle, RECORDS_PER_TAKE);
		}
		return taken;
	}

	/** The self driver (impl/AbstractMatchSearch.java:121-199): every stored forward sequence against the index, toSelf = true. */
	@Override
	public ArrayList<MatchResult> findMatches()
	{
		// The search runs on a worker thread; this thread takes records while the GPUs are still searching and prints them through
		// outputResults, as the reference's pool does every NUM_ELEMENTS_PER_OUTPUT matches (impl/AbstractMatchSearch.java:55,158).
		// The library's sink waits when more than 4 M records are parked natively, so a human-scale search never holds its whole
		// output in memory — native or Java (unless storeResults asks for exactly that).
		final ArrayList<MatchResult> combined = new ArrayList<MatchResult>();
		final long[] delivered = new long[1];
		final Throwable[] failure = new Throwable[1];
		final long h = this.handle;
		nativeSetStreaming(h, true);
		Thread searcher = new Thread(new Runnable()
		{
			@Override
			public void run()
			{
				try
				{
					delivered[0] = nativeFindMatchesSelf(h);
				}
				catch (Throwable t)
				{
					failure[0] = t;
				}
			}
		}, "mhap-hip-search");
		searcher.start();
		long taken = 0;
		try
		{
			taken = take(combined);            // returns when the search is over and every parked record has been taken
		}
		catch (RuntimeException | Error e)
		{
			nativeAbandon(h);                  // the sink must not wait for a taker that is gone
			throw e;
		}
		finally
		{
			boolean interrupted = false;
			while (searcher.isAlive())
			{
				try
				{
					searcher.join();
				}
				catch (InterruptedException ie)
				{
					interrupted = true;
				}
			}
			nativeSetStreaming(h, false);
			if (interrupted)
				Thread.currentThread().interrupt();
		}
		if (failure[0] != null)
			throw new MhapRuntimeException(failure[0]);
		if (taken != delivered[0])
			throw new MhapRuntimeException("Overlap records lost between the library and Java: " + taken + " of " + delivered[0] + ".");
		this.sequencesSearched = nativeStats(this.handle)[1];
		flushOutput();
		return combined;
	}

	/**
	 * The stream driver (impl/AbstractMatchSearch.java:203-285): forward query sketches against the index, toSelf = false.  The
	 * streamer hands out SequenceSketch objects (sketched by Java from a FASTA file, or read from a .dat file); they are
	 * searched on the GPU in batches.  To have the queries sketched on the GPU as well, use findMatches(FastaData).
	 */
	@Override
	public ArrayList<MatchResult> findMatches(final SequenceSketchStreamer data) throws IOException
	{
		ArrayList<MatchResult> combined = new ArrayList<MatchResult>();
		ReadBuffer buf = new ReadBuffer();
		ArrayList<SequenceSketch> batch = new ArrayList<SequenceSketch>();
		SequenceSketch sketch = data.dequeue(true, buf);
		while (sketch != null)
		{
			batch.add(sketch);
			if (batch.size() >= 8192)
			{
				deliver(searchSketches(batch), combined);
				batch.clear();
			}
			sketch = data.dequeue(true, buf);
		}
		deliver(searchSketches(batch), combined);
		flushOutput();
		return combined;
	}

	/** searches a batch of query sketches; returns the number of records parked natively */
	private long searchSketches(List<SequenceSketch> batch)
	{
		int m = batch.size();
		if (m == 0)
			return 0;
		int S = this.orderedSketchSize;
		long[] ids = new long[m];
		int[] seqLength = new int[m], orderedSize = new int[m], orderedSeqLength = new int[m];
		int[] minHashes = new int[m * this.numHashes];
		int[] ordered = new int[m * S * 2];
		for (int i = 0; i < m; i++)
		{
			SequenceSketch sk = batch.get(i);
			SequenceId id = sk.getSequenceId();
			ids[i] = id.getHeaderId();
			if (SequenceId.STORE_FULL_ID)
				this.fullIds.put(id.getHeaderId(), id.getHeader());
			seqLength[i] = sk.getSequenceLength();
			int[] mh = sk.getMinHashes().getMinHashArray();
			if (mh.length != this.numHashes)
				throw new MhapRuntimeException("Number of hashes does not match. Stored size " + this.numHashes + ", input size " + mh.length + ".");
			System.arraycopy(mh, 0, minHashes, i * this.numHashes, this.numHashes);
			// (hash, pos) pairs are only reachable through the sketch's wire form: int seqLength, int kmerSize, int size, then the pairs
			// (sketch/BottomOverlapSketch.java:561-585), big-endian
			BottomOverlapSketch os = sk.getOrderedHashes();
			ByteBuffer wire = ByteBuffer.wrap(os.getAsByteArray());
			orderedSeqLength[i] = wire.getInt();
			wire.getInt();
			int size = wire.getInt();
			if (size > S)
				throw new MhapRuntimeException("Ordered sketch larger than --ordered-sketch-size.");
			orderedSize[i] = size;
			for (int j = 0; j < 2 * size; j++)
				ordered[i * S * 2 + j] = wire.getInt();
		}
		this.sequencesSearched += m;
		return nativeFindMatchesSketches(this.handle, ids, seqLength, minHashes, ordered, orderedSize, orderedSeqLength, m);
	}

	/** Query READS straight from a FASTA file: sketched (forward strand only) and searched on the GPU. */
	public ArrayList<MatchResult> findMatches(FastaData queries) throws IOException
	{
		ArrayList<MatchResult> combined = new ArrayList<MatchResult>();
		ReadBatch batch = new ReadBatch();
		Sequence seq = queries.dequeue();
		while (seq != null)
		{
			batch.add(seq);
			if (batch.full())
			{
				deliver(searchReads(batch), combined);
				batch = new ReadBatch();
			}
			seq = queries.dequeue();
		}
		deliver(searchReads(batch), combined);
		flushOutput();
		return combined;
	}

	/** sketches and searches a batch of query reads on the GPUs; returns the number of records parked natively */
	private long searchReads(ReadBatch batch)
	{
		if (batch.reads.isEmpty())
			return 0;
		rememberIds(batch, false);
		long before = nativeStats(this.handle)[1];
		long parked = nativeFindMatchesReads(this.handle, batch.baseBytes(), batch.offsets(), batch.lengths(), batch.ids(), batch.reads.size());
		this.sequencesSearched += nativeStats(this.handle)[1] - before;
		return parked;
	}

	/** per-query seam (impl/AbstractMatchSearch.java:201): one sketch against the index. */
	@Override
	protected List<MatchResult> findMatches(SequenceSketch hashes, boolean toSelf)
	{
		if (toSelf)
			throw new MhapRuntimeException("Self matches are computed for the whole index at once: call findMatches().");
		ArrayList<SequenceSketch> one = new ArrayList<SequenceSketch>();
		one.add(hashes);
		long parked = searchSketches(one);
		ArrayList<MatchResult> out = new ArrayList<MatchResult>();
		byte[] chunk = nativeTakeRecords(this.handle, RECORDS_PER_TAKE);
		while (chunk != null)
		{
			out.addAll(decode(chunk));
			chunk = nativeTakeRecords(this.handle, RECORDS_PER_TAKE);
		}
		if (out.size() != parked)
			throw new MhapRuntimeException("Overlap records lost between the library and Java.");
		return out;
	}

	@Override
	public long getMatchesProcessed()
	{
		return this.matchesProcessed;
	}

	@Override
	public long getNumberSequencesSearched()
	{
		return this.sequencesSearched;
	}

	public long getNumberSequencesFullyCompared()
	{
		return nativeStats(this.handle)[2];
	}

	public long getNumberElementsProcessed()
	{
		return nativeStats(this.handle)[4];
	}

	@Override
	public List<SequenceId> getStoredForwardSequenceIds()
	{
		return this.storedForwardIds;
	}

	@Override
	public SequenceSketch getStoredSequenceHash(SequenceId id)
	{
		throw new MhapRuntimeException("Stored sketches live in GPU memory; export them with mhap-hip -p (.dat files).");
	}

	/** number of stored sketches, forward and reverse (MinHashSearch.size(), impl/MinHashSearch.java:298-301) */
	@Override
	public int size()
	{
		return (int) nativeStats(this.handle)[0];
	}

	public synchronized void close()
	{
		if (!this.closed)
		{
			nativeDestroy(this.handle);
			this.closed = true;
		}
	}

	@Override
	protected void finalize() throws Throwable
	{
		close();
		super.finalize();
	}
}
