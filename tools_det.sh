#!/bin/bash
# dev tool: is the headless driver bit-reproducible run to run?
set -e
R=/root/repo; W=/tmp/det; rm -rf $W; mkdir -p $W/weights; cd $W
printf "M 4\nLk 1\nLl 1\nS 2\nrmax 1\n" > New_Layer_Param.txt
run() { tag=$1; shift; for i in 1 2 3; do $R/autoencoder-fft_amd/aefft_headless --size 32 --seed 5 --dump $tag$i.f32 "$@" > $tag$i.log 2>&1; done; md5sum $tag*.f32 | awk '{print $1}' | sort | uniq -c | awk -v t=$tag '{printf "%s: %s x %s\n", t, $1, $2}'; }
run fwd --frames 3 --script "g.."
run one --frames 4 --script "g1.."
run sp --frames 4 --script "f1.."
run addl --frames 4 --script "gn.."
run two --frames 8 --script "g1..1..."
run lr --frames 16 --script "g55555555555.1.."
