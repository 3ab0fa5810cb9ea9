#!/bin/bash
# Host code under the sanitizers, no GPU: the file reader / fan-in / stand-alone CLI against the test stand-ins
# (tests/host_stub) with ASan + UBSan, and the ordered walk (readsb_amd/csrc/resolve.cpp) replayed from a dumped chunk
# (MGPU_DUMP_DIR=… during any GPU run; default gpurun_out/dump) with TSan and ASan + UBSan; the library's GPU-free self-checks
# (tools/selftest_check.cpp: parallel walk, the sharded walk's protocol, the block-wise double sum) with both as well.
#   usage: tools/sanitize_host.sh [dump_dir]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
H=$R/readsb_amd/host; T=$R/tests/host_stub; O=$R/oracle
D=${1:-$R/gpurun_out/dump}
W=$(mktemp -d)
cd "$W"
SAN="-g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer"
gcc -std=gnu11 $SAN -Wall -Wextra -o reader_check $T/reader_check.c $H/demod_gpu.c -lpthread -lm
gcc -std=gnu11 $SAN -Wall -Wextra -o fanin_check $T/fanin_check.c $H/sdr_gpu_fanin.c $H/demod_gpu.c -lpthread -lm
gcc -std=gnu11 $SAN -ffp-contract=off -Wall -o cli_standin $H/readsb_gpu_ifile.c $H/demod_gpu.c $T/modes_gpu_standin.c $O/modes_oracle.c $O/modes_oracle_fields.c -lpthread -lm
head -c 2371641 /dev/urandom > in.iq
./reader_check in.iq UC8 3 out.iq
cmp <(head -c 2371640 in.iq) out.iq
./fanin_check pref --ifile in.iq --iformat SC16 --ifile in.iq --gpu-chunk-buffers 2
./cli_standin --ifile in.iq --iformat UC8 --fix --raw --mlat --gpu-chunk-buffers 4 --modeac > /dev/null
if [ -f "$D/walk_recs.bin" ]; then
  g++ -O1 -g -std=c++17 -fsanitize=thread -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -pthread -o walk_tsan $R/tools/walk_replay.cpp $R/readsb_amd/csrc/resolve.cpp
  ./walk_tsan "$D" | tail -2
  g++ -O1 $SAN -std=c++17 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -pthread -o walk_asan $R/tools/walk_replay.cpp $R/readsb_amd/csrc/resolve.cpp
  ./walk_asan "$D" | tail -2
else
  echo "no dumped chunk under $D: walk replay skipped"
fi
C=$R/readsb_amd/csrc
CXXS="-std=c++17 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -pthread $R/tools/selftest_check.cpp $C/selftest.cpp $C/resolve.cpp $C/seqsum.cpp"
g++ -O1 $SAN $CXXS -o selftest_asan
./selftest_asan | tail -3
g++ -O1 -g -fsanitize=thread $CXXS -o selftest_tsan          # minutes: the walk's spinning workers under TSan
./selftest_tsan 2> tsan.log | tail -3
if grep -q "WARNING: ThreadSanitizer" tsan.log; then head -40 tsan.log; exit 1; fi
echo "sanitizers: clean"
