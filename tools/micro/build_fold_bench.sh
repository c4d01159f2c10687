#!/bin/bash
# Compiles tools/micro/fold_bench.hip into root-simple-mcmc_amd/build/micro (git-ignored, but it travels to the GPU box) and prints the register /
# scratch statistics of the fold_ring_kernel instantiations from the kept listing.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/root-simple-mcmc_amd/build/micro
mkdir -p $OUT
cd $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I$ROOT/include -I$ROOT/root-simple-mcmc_amd/csrc -I$ROOT/tools/micro \
    $ROOT/tools/micro/fold_bench.hip -o fold_bench -save-temps=obj
S=fold_bench-hip-amdgcn-amd-amdhsa-gfx950.s
for k in ILi5ELb0E ILi4ELb0E ILi5ELb1E; do
    awk "/^_ZN5smcmcL16fold_ring_kernel${k}/,/\.end_amdhsa_kernel/" $S > /tmp/fr_$k.s
    echo $k lines $(wc -l < /tmp/fr_$k.s) mfma $(grep -c v_mfma /tmp/fr_$k.s) scratch $(grep -c "scratch_" /tmp/fr_$k.s) \
        accvgpr $(grep -c "v_accvgpr" /tmp/fr_$k.s) $(grep -E "next_free_vgpr|accum_offset|private_segment_fixed" /tmp/fr_$k.s | tr -s '\t\n' ' ')
done
