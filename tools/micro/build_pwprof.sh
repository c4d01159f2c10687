# The profiling build of the per-chain wave kernel (PW_PROFILE: s_memtime stamps per section instead of the record):
# root-simple-mcmc_amd/build/prof/libsmcmc_amd_prof.so = the plain library's objects with perchain_wave.o replaced.
# Read by tools/micro/pwprof.py on the GPU box.  usage: bash tools/micro/build_pwprof.sh   (after the plain build)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); B=$R/root-simple-mcmc_amd/build; P=$B/prof
mkdir -p $P
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function \
    -I$R/include -I$R/root-simple-mcmc_amd/csrc -DPW_PROFILE -c $R/root-simple-mcmc_amd/csrc/smcmc_perchain_wave_inst.hip -o $P/perchain_wave.o
OBJS=$(ls $B/*.o | grep -v -e '_frozen_definition\.o$' -e '_user\.o$' -e '/user_large' -e '/perchain_wave\.o$' -e '/inst_dp[0-9]*_l3\.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $OBJS $P/perchain_wave.o -o $P/libsmcmc_amd_prof.so
echo built $P/libsmcmc_amd_prof.so
