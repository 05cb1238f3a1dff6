for T in 10 60 500; do echo "T=$T"; CTCASR_F16=1 python tools/rnn_microbench.py $T 32 1024 | grep -E "fwd:|bwd:"; done
echo fp32; for T in 10 60; do python tools/rnn_microbench.py $T 32 1024 | grep -E "fwd:|bwd:"; done
