# Build variants of the own split-GEMM kernel (csrc/split_gemm.hip) with different -D tuning
# macros into ctc_asr_amd/csrc/_obj/sg_<name>.so, for tools/split_gemm_variants.py:
#   tools/split_gemm_variants.sh q24v4u9 -DSG_QUIET=24 -DSG_VALU=4 -DSG_UNIT=9
set -e
cd "$(dirname "$0")/.."
name=$1; shift
obj=ctc_asr_amd/csrc/_obj
mkdir -p $obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" \
    ctc_asr_amd/csrc/split_gemm.hip -o $obj/sg_$name.so
echo $obj/sg_$name.so
