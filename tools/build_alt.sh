# A/B library with extra defines for the recurrence kernels:
#   tools/build_alt.sh lb8 -DPRNN_CHAIN_LB=8   -> ctc_asr_amd/csrc/_obj/alt_lb8.so
# (use with CTCASR_LIB=... tools/rnn_microbench.py / tools/ab_rnn.sh)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python -m ctc_asr_amd.build >/dev/null
obj=ctc_asr_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" \
    -Rpass-analysis=kernel-resource-usage -c ctc_asr_amd/csrc/rnn_persistent.hip \
    -o $obj/alt_$name.o 2> $obj/alt_$name.remarks
others=$(ls $obj/*.o | grep -v "alt_\|rnn_persistent.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $obj/alt_$name.so $obj/alt_$name.o $others
echo $obj/alt_$name.so
