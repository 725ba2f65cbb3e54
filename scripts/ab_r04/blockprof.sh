for lib in scripts/ab/libthesia_amd_bprof.so scripts/ab/libthesia_amd_bprof1.so; do
echo "== $lib"
THESIA_AMD_LIB=$lib python3 scripts/block_prof.py --nfft 32768
THESIA_AMD_LIB=$lib python3 scripts/block_prof.py --nfft 32768 --win 19200 --hop 4800
THESIA_AMD_LIB=$lib python3 scripts/block_prof.py --nfft 16384
THESIA_AMD_LIB=$lib python3 scripts/block_prof.py --nfft 8192
done
