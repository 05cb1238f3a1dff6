# A/B library with extra defines for csrc/dgrad16.hip:
#   tools/build_alt_dg.sh same -DDG_PROBE=1   -> ctc_asr_amd/csrc/_obj/alt_dg_same.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python -m ctc_asr_amd.build >/dev/null 2>&1
obj=ctc_asr_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize "$@" \
    -Rpass-analysis=kernel-resource-usage -c ctc_asr_amd/csrc/dgrad16.hip \
    -o $obj/alt_dg_$name.o 2> $obj/alt_dg_$name.remarks
others=$(ls $obj/*.o | grep -v "alt_\|/dgrad16.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $obj/alt_dg_$name.so $obj/alt_dg_$name.o $others
echo $obj/alt_dg_$name.so
