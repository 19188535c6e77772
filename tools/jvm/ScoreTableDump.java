// ScoreTableDump — the three places where MHAP's records depend on JDK / Guava arithmetic that this repository restates without a
// JVM (SURVEY.md Appendix D hazards 3, 4 and the Bloom sizing), dumped bit-exactly so that the first box with a JVM can diff them
// against the native side (tools/jvm/native_dump.py) in one command: tests/golden/verify_against_jar.sh.
//   javac -cp $MHAP_JAR -d $T tools/jvm/ScoreTableDump.java && java -cp $MHAP_JAR:$T ScoreTableDump score 12 1536 > score.jvm
//   ... fmt6 > fmt6.jvm ; ... bloom > bloom.jvm
// score: for every (inter, k), 0 <= inter <= k <= S: jaccardToIdentity(inter / k) exactly as
//        BottomOverlapSketch.java:391-395 computes it (Math.log, Math.exp), as the hex of Double.doubleToLongBits.
// fmt6 : String.format("%.6f") (MatchResult.java:100, Locale.US) of 20 000 doubles from a fixed LCG, incl. exact 7th-digit ties.
// bloom: Guava BloomFilter.create(funnel(putLong), n, 1e-5): bit size, number of hash functions, and mightContain of 64 probes
//        after putting 1000 values (FrequencyCounts.java:137,192,272-278).
import java.lang.reflect.Field;
import java.util.Locale;

import com.google.common.hash.BloomFilter;

public final class ScoreTableDump
{
	static long lcg(long x)
	{
		return x * 6364136223846793005L + 1442695040888963407L;
	}

	public static void main(String[] args) throws Exception
	{
		Locale.setDefault(Locale.US);
		String what = args.length > 0 ? args[0] : "score";
		StringBuilder sb = new StringBuilder();
		if (what.equals("score"))
		{
			int k2 = Integer.parseInt(args[1]), S = Integer.parseInt(args[2]);
			for (int k = 0; k <= S; k++)
				for (int inter = 0; inter <= k; inter++)
				{
					double score = (k == 0) ? 0.0 : (double) inter / (double) k;
					double d = -1.0 / (double) k2 * Math.log(2.0 * score / (1.0 + score));
					sb.append(Long.toHexString(Double.doubleToLongBits(Math.exp(-d)))).append('\n');
					if (sb.length() > (1 << 20))
					{
						System.out.print(sb);
						sb.setLength(0);
					}
				}
		}
		else if (what.equals("fmt6"))
		{
			long x = 0x4D484150L;
			for (int i = 0; i < 20000; i++)
			{
				x = lcg(x);
				double v;
				switch (i % 4)
				{
				case 0:
					v = (double) (x >>> 11) / 9007199254740992.0;                 // [0, 1)
					break;
				case 1:
					v = ((x >>> 40) % 2000000L) / 1.0e6 + 5.0e-7;                   // n.nnnnnn5: a HALF_UP tie of the shortest representation
					break;
				case 2:
					v = (double) ((x >>> 44) % 3000L);                              // rawScore: a count
					break;
				default:
					v = 1.0 - (double) (x >>> 11) / 9007199254740992.0 * 0.3;     // 1 - score for identities around 0.78..1
				}
				sb.append(Long.toHexString(Double.doubleToLongBits(v))).append(' ').append(String.format("%.6f", v)).append('\n');
			}
		}
		else
		{
			long[] sizes = {1L, 2L, 10L, 1000L, 65536L, 1000000L, 123456789L, 4000000000L};
			Field fBits = BloomFilter.class.getDeclaredField("bits");
			Field fK = BloomFilter.class.getDeclaredField("numHashFunctions");
			fBits.setAccessible(true);
			fK.setAccessible(true);
			for (long n : sizes)
			{
				if (n > 200000000L)
					continue;   // (the bit array itself is allocated: keep the probe small)
				BloomFilter<Long> bf = BloomFilter.create((value, sink) -> sink.putLong(value), n, 1.0e-5);
				Object bits = fBits.get(bf);
				java.lang.reflect.Method bitSize = bits.getClass().getDeclaredMethod("bitSize");
				bitSize.setAccessible(true);
				sb.append("size ").append(n).append(' ').append(bitSize.invoke(bits)).append(' ').append(fK.getInt(bf)).append('\n');
			}
			BloomFilter<Long> bf = BloomFilter.create((value, sink) -> sink.putLong(value), 1000L, 1.0e-5);
			long x = 7L;
			for (int i = 0; i < 1000; i++)
			{
				x = lcg(x);
				bf.put(x);
			}
			long y = 7L;
			for (int i = 0; i < 64; i++)
			{
				y = lcg(y);
				long probe = (i % 2 == 0) ? y : y ^ 0x5555555555555555L;       // every other probe was put
				sb.append("probe ").append(probe).append(' ').append(bf.mightContain(probe) ? 1 : 0).append('\n');
			}
		}
		System.out.print(sb);
	}
}
