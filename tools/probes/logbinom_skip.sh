# which parts of the log-binomial kernel (four-byte-gather form) have to be there for bs_rank1_bias on a second stream to disturb it
# (diagnostics build; BS_LOGBINOM_SKIP: 1 = no dot products, 2 = no GELU, 4 = no softmax phase)
for sk in ${SKIPS:-0 1 2 4 3 7}; do echo "##### BS_LOGBINOM_SKIP=$sk"; BS_LOGBINOM_SKIP=$sk bash tools/probes/gather_beside_stream_kinds.sh ".qkv.r1::20" 2>&1 | grep "beside\|alone"; done
