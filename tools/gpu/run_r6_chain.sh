# Round 6: the chain form of the lean loop (dev_modular_wave.h: wave_decode_channel_chain) against the build without it —
# pixels of every fixture bit for bit, the phases of a lone flight, then the headline A/B.
ulimit -c 0
OUT=${OUT:-gpurun_out/chain}; mkdir -p $OUT
A=${A:-tools/gpu/ab/libjxlamd_A.so}
JXLAMD_LIB=$A timeout 900 python tools/gpu/decode_digest.py > $OUT/digest_A.txt 2>$OUT/digest_A.err
timeout 900 python tools/gpu/decode_digest.py > $OUT/digest_B.txt 2>$OUT/digest_B.err
echo "[chain] digests: $(wc -l < $OUT/digest_A.txt) files, $(diff $OUT/digest_A.txt $OUT/digest_B.txt | grep -c '^<') differ, $(grep -c ERROR $OUT/digest_B.txt) refused"
diff $OUT/digest_A.txt $OUT/digest_B.txt | head -20
JXLAMD_LIB=$A timeout 600 python tools/prof_flight.py 64 > $OUT/flight_A.txt 2>&1
timeout 600 python tools/prof_flight.py 64 > $OUT/flight_B.txt 2>&1
echo "[chain] flight A"; tail -12 $OUT/flight_A.txt
echo "[chain] flight B"; tail -12 $OUT/flight_B.txt
VARIANTS="A:LIB=$A B:" REPS=2 OUT=$OUT/ab bash tools/gpu/ab.sh
