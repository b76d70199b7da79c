for a in ${ABLS:-0 1 2 4 8}; do echo "##### BS_RANK1_ABLATE=$a"; BS_RANK1_ABLATE=$a bash tools/probes/gather_beside_stream_kinds.sh ".qkv.r1::20" 2>&1 | grep "beside"; done
