# The profiling build of the reference-order large-dimension kernel (PANEL_PROFILE: cycle stamps per section, printed by
# workgroup 0): root-simple-mcmc_amd/build/prof/libsmcmc_amd_panelprof.so = the plain library's objects with
# panel_w4.o / panel_w8.o replaced.  usage: bash tools/micro/build_panelprof.sh   (after the plain build), then on the
# GPU box: SMCMC_AMD_LIBRARY=... see tools/micro/panelprof.py
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); B=$R/root-simple-mcmc_amd/build; P=$B/prof
mkdir -p $P
for W in 4 8; do
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function \
    -I$R/include -I$R/root-simple-mcmc_amd/csrc -DPANEL_PROFILE -DSMCMC_PANEL_W=$W -c $R/root-simple-mcmc_amd/csrc/smcmc_panel_inst.hip -o $P/panel_w$W.o &
done; wait
OBJS=$(ls $B/*.o | grep -v -e '_frozen_definition\.o$' -e '_user\.o$' -e '/user_large' -e '/panel_w[48]\.o$' -e '/inst_dp[0-9]*_l3\.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $OBJS $P/panel_w4.o $P/panel_w8.o -o $P/libsmcmc_amd_panelprof.so
echo built $P/libsmcmc_amd_panelprof.so
