# Host code of libza_amd.so under AddressSanitizer + UBSan (device code is compiled as usual and never runs here):
#   bash tools/asan_host.sh            builds /tmp/za_amd_asan/libza_amd.so, runs the CPU tests that call the library
#                                      and the parser fuzzer against it
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${ZA_ASAN_DIR:-/tmp/za_amd_asan}
mkdir -p $OUT
for s in msm ntt heval formats keyload prover synth setup pairing outer; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result -fno-gpu-rdc \
    -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -c $ROOT/za_amd/csrc/$s.hip -o $OUT/$s.o &
done
wait
mkdir -p $OUT/synth && mv $OUT/synth.o $OUT/synth/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $OUT/libza_amd.so $OUT/*.o -Wl,--version-script=$ROOT/za_amd/csrc/libza_amd.map
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $OUT/libza_amd_synth.so $OUT/synth/synth.o -L$OUT -lza_amd -Wl,-rpath,'$ORIGIN'
ASAN=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export ZA_AMD_LIB=$OUT/libza_amd.so LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
cd $ROOT
python -m pytest tests/test_outer.py tests/test_abi.py -x -q -s 2>&1 | tail -5
python tools/fuzz_host_parsers.py ${1:-120}
