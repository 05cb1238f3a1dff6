# A/B library with extra defines for ONE source file of csrc/:
#   tools/build_alt_file.sh wgrad16 w32 -DWG_W=32   -> ctc_asr_amd/csrc/_obj/alt_wgrad16_w32.so
# (use with CTCASR_LIB=...)
set -e
cd "$(dirname "$0")/.."
file=$1; name=$2; shift; shift
python -m ctc_asr_amd.build >/dev/null 2>&1
obj=ctc_asr_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude "$@" \
    -Rpass-analysis=kernel-resource-usage -c ctc_asr_amd/csrc/$file.hip \
    -o $obj/alt_${file}_$name.o 2> $obj/alt_${file}_$name.remarks
others=$(ls $obj/*.o | grep -v "alt_\|/$file.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $obj/alt_${file}_$name.so $obj/alt_${file}_$name.o $others
echo $obj/alt_${file}_$name.so
