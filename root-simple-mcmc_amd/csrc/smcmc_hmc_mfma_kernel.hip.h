// smcmc_hmc_mfma_kernel.hip.h -- TSimpleHMC::Step() for the quadratic-form likelihood with the
// gradient contraction on the FP64 matrix pipe.
//
// Same chain as hmc_step_kernel (smcmc_hmc_kernel.hip.h; reference TSimpleHMC.H:279-401,
// 554-570, 582-651) except for ONE thing, which the engine calls the fused order
// (SMCMC_P_EXACT_ARITHMETIC = 0 of the HMC engine, mirrored by oracle/hmc_oracle.c): the
// gradient g_i = sum_j Error(i,j) q_j (TDummyLogLikelihood.H:34-42) accumulates j in the
// reference's ascending order but with one fused multiply-add per term, which is what a chain
// of v_mfma_f64_16x16x4_f64 does (tests/test_gpu_parity.py pins that order on the hardware).
// Everything else -- momentum refresh, leapfrog updates, kinetic energy and potential summed in
// dimension order, Hamiltonian test -- keeps the reference's operations.
//
// Layout: a workgroup of 8 wavefronts advances 32 chains (two 16-chain tiles); the work of a
// trajectory is the 21 products  G[512 x 32] = Error[512 x 512] . Q[512 x 32].  A matrix
// instruction produces D[row = component][col = chain]; lane l holds column l & 15 and rows
// (l >> 4) + 4 r.  Positions, momenta and gradients stay in registers in exactly that layout
// (wavefront w owns the component tiles w, w + 8, ...), so the leapfrog updates are register to
// register.  Only Q has to be seen by everybody: it is published to LDS as q[component][chain]
// (128 KB) before every gradient, from where the B operand of k-quad kq is the row quad
// 4 kq .. 4 kq + 3.  Error comes from L2 in operand order (Eop[tile][kq][lane], one coalesced
// 512-byte read per instruction, laid out by the host).  Sums the reference runs over all
// components in index order (kinetic energy, potential) are formed term by term in registers,
// published through the same LDS array and added up in order by one lane per chain.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smcmc.h"
#include "smcmc_detmath.h"
#include "smcmc_hmc_kernel.hip.h"
#include "smcmc_panel_mfma_kernel.hip.h"

namespace smcmc {

constexpr int kMfCT = 32;                            // chains per workgroup
constexpr int kMfW = 8;                              // wavefronts per workgroup
constexpr int kMfTIMax = 4;                          // 16-component tiles per wavefront: dim <= 512
constexpr int kMfDimMax = 16 * kMfW * kMfTIMax;

// Eop[(tile * nkqp + kq) * 64 + lane] = Error(16 tile + (lane & 15), 4 kq + (lane >> 4)), zero padded
// (k-quads per tile rounded up to the prefetch depth of pm_contract, zero padded)
inline size_t hmc_mfma_eop_doubles(int dim) { return panel_mfma_uop_doubles(dim); }

// Ex[(tile * dim + j) * 16 + 4 rq + r] = Error(16 tile + 4 r + rq, j): column j of a tile's sixteen rows in the
// order the matrix layout holds them (lane quarter rq, register r), one 32-byte read per lane
inline size_t hmc_exact_ex_doubles(int dim) { return (size_t)((dim + 15) / 16) * dim * 16; }

// TI = 16-component tiles a wavefront owns: dim <= 128 TI.
// FUSED = false: the reference's order, g[i] -= Error(i,j) * q[j] as an un-fused multiply and subtract, j
// ascending (TDummyLogLikelihood.H:34-42), on the vector pipe in the same layout (p.Eperm = Ex).
template <int kMfTI, bool FUSED = true>
__global__ void __launch_bounds__(kMfW* kWave, 1) hmc_mfma_kernel(const HmcParams p) {
    __shared__ double qs[16 * kMfW * kMfTI * kMfCT];   // [component][chain]: the published vector
    // one 16-component tile per wavefront x 32 chains: ordered sums that must leave qs (the positions) alone; its
    // first bytes double as the verdict of the Hamiltonian test (with kMfTI = 4 the two arrays fill the 160 KB)
    __shared__ double rs[16 * kMfW * kMfCT];
    int* const verdict = (int*)rs;

    const int lane = threadIdx.x & (kWave - 1);
    const int w = threadIdx.x / kWave;
    const int c = lane & 15, rq = lane >> 4;
    const int base = blockIdx.x * kMfCT;
    const int D = p.dim;
    // the row stride is re-read through an opaque copy every step: otherwise the 96 element addresses of
    // q, pm and qn are hoisted out of the step loop as 64-bit values and spilled
    size_t NP = (size_t)p.npad;
    const int ntiles = (D + 15) / 16, nkqp = panel_mfma_nkq_padded(D);
    const uint32_t ew = smcmc_accept_word((uint32_t)D);

    // this lane's chains (one per chain tile) and, for the summing lanes of wavefront 0, `mychain`
    const bool summer = (w == 0) && (lane < kMfCT);
    const int mychain = base + lane;

    typedef double f64x4v __attribute__((ext_vector_type(4)));
    f64x4v pn[kMfTI][2], gr[kMfTI][2];   // momenta and gradients of the owned elements; positions live in qs

    // element (t, ct, r): component 16 (t W + w) + 4 r + rq of chain base + 16 ct + c
    auto comp = [&](int t, int r) { return 16 * (t * kMfW + w) + 4 * r + rq; };
    auto owns = [&](int t) { return t * kMfW + w < ntiles; };
    auto slot = [&](int t, int ct, int r) { return comp(t, r) * kMfCT + 16 * ct + c; };

    // v[t][ct][r] -> qs[component][chain]; rows past D (inside the owned tiles) carry zeros
    auto publish = [&](auto&& value) {
        __syncthreads();   // readers of the previous contents are done
#pragma unroll
        for (int t = 0; t < kMfTI; ++t) {
            if (!owns(t)) continue;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) qs[slot(t, ct, r)] = (comp(t, r) < D) ? value(t, ct, r) : 0.0;
        }
        __syncthreads();
    };
    // sum of the published column of every chain, components in ascending order (one lane per chain)
    // (eight LDS reads are issued ahead of the eight serially dependent additions that consume them, the next eight in flight
    // meanwhile: read one at a time, every addition waited out an LDS round trip while seven wavefronts sat at the barrier)
    auto ordered_sum = [&]() {
        double s = 0.0;
        if (summer) {
            constexpr int kAhead = 8;
            int i0 = 0;
            double v[kAhead], vn[kAhead];
            if (D >= kAhead) {
#pragma unroll
                for (int u = 0; u < kAhead; ++u) v[u] = qs[u * kMfCT + lane];
            }
            for (; i0 + kAhead <= D; i0 += kAhead) {
                const int nx = (i0 + 2 * kAhead <= D) ? i0 + kAhead : i0;   // (or this chunk again: a harmless re-read)
#pragma unroll
                for (int u = 0; u < kAhead; ++u) vn[u] = qs[(nx + u) * kMfCT + lane];
#pragma unroll
                for (int u = 0; u < kAhead; ++u) s += v[u];
#pragma unroll
                for (int u = 0; u < kAhead; ++u) v[u] = vn[u];
            }
            for (; i0 < D; ++i0) s += qs[i0 * kMfCT + lane];
        }
        return s;
    };

    // sum over the components, ascending, of value(t, ct, r) for every chain, 128 components at a time through rs
    auto ordered_sum_tiles = [&](auto&& value) {
        double sacc = 0.0;
#pragma unroll
        for (int t = 0; t < kMfTI; ++t) {
            __syncthreads();
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    rs[(w * 16 + 4 * r + rq) * kMfCT + 16 * ct + c] = (owns(t) && comp(t, r) < D) ? value(t, ct, r) : 0.0;
            __syncthreads();
            if (summer)
                for (int k = 0; k < 16 * kMfW; ++k)
                    if (16 * kMfW * t + k < D) sacc += rs[k * kMfCT + lane];
        }
        __syncthreads();
        return sacc;
    };

    // gr = Error q for the owned components, q = the positions in qs (PotentialGradient,
    // TSimpleHMC.H:467-492, for the quadratic form of TDummyLogLikelihood.H:34-42)
    auto gradient = [&]() {
        __syncthreads();   // every owner has written its positions
#pragma unroll
        for (int t = 0; t < kMfTI; ++t)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) gr[t][ct] = f64x4v{0.0, 0.0, 0.0, 0.0};
        if constexpr (FUSED) {
            const double* qs_lane = qs + rq * kMfCT + c;
#pragma unroll
            for (int t = 0; t < kMfTI; ++t) {
                if (!owns(t)) continue;
                pm_contract(p.Eperm + lane + (size_t)(t * kMfW + w) * nkqp * 64, qs_lane, nkqp, gr[t][0], gr[t][1]);
            }
        } else {
            const double* ex = p.Eperm + 4 * rq;
            const double* qv = qs + c;
            f64x4v en[kMfTI];   // column j + 1 of Error is fetched while column j is consumed
#pragma unroll
            for (int t = 0; t < kMfTI; ++t)
                en[t] = owns(t) ? *(const f64x4v*)(ex + ((size_t)(t * kMfW + w) * D) * 16) : f64x4v{0.0, 0.0, 0.0, 0.0};
            for (int j = 0; j < D; ++j) {
                const double q0 = qv[j * kMfCT], q1 = qv[j * kMfCT + 16];
                f64x4v e[kMfTI];
#pragma unroll
                for (int t = 0; t < kMfTI; ++t) {
                    e[t] = en[t];
                    if (owns(t) && j + 1 < D) en[t] = *(const f64x4v*)(ex + ((size_t)(t * kMfW + w) * D + j + 1) * 16);
                }
#pragma unroll
                for (int t = 0; t < kMfTI; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gr[t][0][r] -= e[t][r] * q0;
                        gr[t][1][r] -= e[t][r] * q1;
                    }
            }
#pragma unroll
            for (int t = 0; t < kMfTI; ++t) {   // TSimpleHMC.H:486: the potential's gradient is -grad(log L)
                gr[t][0] = -gr[t][0];
                gr[t][1] = -gr[t][1];
            }
        }
        __syncthreads();   // the positions may be overwritten again
    };

    // potential of the positions in qs from their gradient (the engine's association, see
    // hmc_step_kernel): U = sum_i 0.5 q_i (Error q)_i in dimension order.  The positions are parked
    // in the proposal buffer qn (HBM) first, because the sum goes through qs.
    auto potential = [&]() {
#pragma unroll
        for (int t = 0; t < kMfTI; ++t)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = comp(t, r);
                    if (owns(t) && i < D) {
                        const double qv = qs[slot(t, ct, r)];
                        p.qn[(size_t)i * NP + base + 16 * ct + c] = qv;
                        gr[t][ct][r] = 0.5 * qv * gr[t][ct][r];
                    }
                }
        publish([&](int t, int ct, int r) { return gr[t][ct][r]; });
        return ordered_sum();
    };
    auto kinetic = [&]() {   // KineticEnergy (:535-542): ke += p*p/2.0
        publish([&](int t, int ct, int r) { return pn[t][ct][r] * pn[t][ct][r] / 2.0; });
        return ordered_sum();
    };

    // rows past the last owned tile (the contraction runs to a multiple of the prefetch depth) are never
    // published: they must hold zeros, not whatever the LDS held before
    for (int k = threadIdx.x; k < 16 * kMfW * kMfTI * kMfCT; k += kMfW * kWave) qs[k] = 0.0;
    __syncthreads();

    // per-chain state lives in the summing lanes
    double pot_acc = 0.0, pot_prop = 0.0, acceptance = 0.0;
    int naccept = 0, last_accept = 0, trials = 0;
    if (summer) {
        pot_acc = -p.lane_f64[SMCMC_LANE_LOGL * NP + mychain];
        pot_prop = -p.lane_f64[SMCMC_LANE_LOGL_PROPOSED * NP + mychain];
        acceptance = p.lane_f64[SMCMC_LANE_ACCEPTANCE * NP + mychain];
        naccept = p.lane_i32[SMCMC_LANE_NACCEPT * NP + mychain];
        last_accept = p.lane_i32[SMCMC_LANE_LAST_ACCEPT * NP + mychain];
        trials = p.lane_i32[SMCMC_LANE_TRIALS * NP + mychain];
    }

    // positions of the accepted point into qs (LeapFrog: qNew = position, :586)
    auto load_q = [&]() {
        publish([&](int t, int ct, int r) { return p.q[(size_t)comp(t, r) * NP + base + 16 * ct + c]; });
    };

    if (p.init_only) {
        // Start (:210-269): SetPosition's Potential(start) for every chain
        load_q();
        gradient();
        const double u0 = potential();
        if (summer && mychain < p.nchains) {
            p.lane_f64[SMCMC_LANE_LOGL * NP + mychain] = -u0;
            p.lane_f64[SMCMC_LANE_LOGL_PROPOSED * NP + mychain] = -u0;
        }
        return;
    }

    // adaptive mode: the chains' own step length and leapfrog count (lanes); the summing lanes carry their chain's
    // tuning state through the step (one step per launch)
    double mean_eps = 0.0, reversal = 0.0;
    int lfrog = 0, contributes = 1;
    if (summer && p.adaptive) {
        mean_eps = p.lane_f64[kHmcLaneMeanEpsilon * NP + mychain];
        reversal = p.lane_f64[kHmcLaneReversalLen * NP + mychain];
        lfrog = p.lane_i32[kHmcLaneLeapfrog * NP + mychain];
    }

    for (int s = 0; s < p.nsteps; ++s) {
        const uint64_t step = (uint64_t)(p.step0 + (uint32_t)s + 1u);   // ++fStepCount, :286
        ++trials;
        asm volatile("" : "+s"(NP));

        // ---- ProposeMomentum (:554-570) ----
        const double mix = __builtin_sqrt(1.0 - p.alpha * p.alpha);
        double eps[2];
        int Lc[2];   // leapfrog count of this lane's two chains
        int automatic = 0;   // some chain of the workgroup keeps an automatic leapfrog count (fLeapFrogSteps > 0)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const uint32_t gid = p.chain_offset + (uint32_t)(base + 16 * ct + c);
            double abs_eps = p.abs_eps;
            Lc[ct] = p.leapfrog;
            if (p.adaptive) {
                abs_eps = __builtin_fabs(p.lane_f64[kHmcLaneMeanEpsilon * NP + base + 16 * ct + c]);
                const int l = p.lane_i32[kHmcLaneLeapfrog * NP + base + 16 * ct + c];
                Lc[ct] = (base + 16 * ct + c < p.nchains) ? (l < 0 ? -l : l) : 0;
                automatic |= (l > 0) ? 1 : 0;
            }
#pragma unroll
            for (int t = 0; t < kMfTI; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = comp(t, r);
                    double v = 0.0;
                    if (owns(t) && i < D) {
                        const double m = p.pm[(size_t)i * NP + base + 16 * ct + c];
                        if (p.alpha >= 1.0) {
                            v = m / p.alpha;
                        } else {
                            const uint32_t pr = (uint32_t)i >> 1;
                            smcmc_u32x4 blk = smcmc_draw_block(p.seed, gid, step, pr >> 1, SMCMC_STREAM_HMC);
                            double n0, n1;
                            const uint32_t w0 = (pr & 1u) ? blk.v[2] : blk.v[0], w1 = (pr & 1u) ? blk.v[3] : blk.v[1];
                            smcmc_normal_pair(w0, w1, &n0, &n1);
                            v = p.alpha * m + mix * ((i & 1) ? n1 : n0);
                        }
                    }
                    pn[t][ct][r] = v;
                    if (p.adaptive && owns(t) && i < D) p.pn[(size_t)i * NP + base + 16 * ct + c] = v;   // LeapFrog: momentum = pNew (:587)
                    __builtin_amdgcn_sched_barrier(0);   // one draw at a time: interleaved they exhaust the registers
                }
            // ---- epsilon (:297-298) ----
            const smcmc_u32x4 eblk = smcmc_draw_block(p.seed, gid, step, ew >> 2, SMCMC_STREAM_HMC);
            const double lo = 0.9 * abs_eps, hi = 1.1 * abs_eps;
            eps[ct] = lo + (hi - lo) * smcmc_u01(smcmc_select_word(eblk, ew & 3u));
        }
        const double ke0 = kinetic();                                   // :292
        load_q();

        // ---- LeapFrog (:582-651).  Chains of the workgroup may differ in their leapfrog count: the workgroup runs to
        // the largest one, a chain takes its last (half) kick at its own count and then stands still. ----
        const int Lmax = wave_max_i32(Lc[0] > Lc[1] ? Lc[0] : Lc[1]);
        // the reversal test (:633-638) only ever reaches a chain through hmc_retune_after_leapfrog, which listens when the
        // chain's count is automatic: with every count fixed (SetLeapFrog) its 500-term ordered sums are skipped.
        // Every wavefront of the workgroup holds the same 32 chains, so a wavefront-wide test is workgroup-uniform.
        const bool reversal_wanted = p.adaptive && wave_max_i32(automatic) != 0;
        const int Lmine = lfrog < 0 ? -lfrog : lfrog;                   // summing lanes: their chain's count
        int status = 1;                                                 // leapStatus of the summing lane's chain
        // iteration ls of a chain with L steps: ls < L - 1 the body of :623-639, ls == L - 1 the last position step
        // and half kick of :641-648, afterwards nothing; ls = -1 is the first half kick of :618-620
        auto kick = [&](int ls) {
            const bool live0 = ls < Lc[0], live1 = ls < Lc[1];
            const bool half0 = (ls < 0) || (ls == Lc[0] - 1), half1 = (ls < 0) || (ls == Lc[1] - 1);
#pragma unroll
            for (int t = 0; t < kMfTI; ++t)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool live = ct ? live1 : live0, half = ct ? half1 : half0;
                        const double m = pn[t][ct][r];
                        const double k = half ? m - eps[ct] * gr[t][ct][r] / 2.0 : m - eps[ct] * gr[t][ct][r];
                        pn[t][ct][r] = live ? k : m;
                    }
        };
        auto drift = [&](int ls) {                                      // qNew[i] += eps*pNew[i]
#pragma unroll
            for (int t = 0; t < kMfTI; ++t) {
                if (!owns(t)) continue;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    if (!(ls < Lc[ct])) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (comp(t, r) < D) qs[slot(t, ct, r)] = qs[slot(t, ct, r)] + eps[ct] * pn[t][ct][r];
                    }
                }
            }
        };
        if (Lmax < 1) {
            // the one-step shortcut (:598-611): qNew += eps*(momentum + pNew)/2 with pNew == momentum
#pragma unroll
            for (int t = 0; t < kMfTI; ++t) {
                if (!owns(t)) continue;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double m = pn[t][ct][r];
                        if (comp(t, r) < D) qs[slot(t, ct, r)] = qs[slot(t, ct, r)] + eps[ct] * (m + m) / 2.0;
                    }
            }
            gradient();                                                 // for the potential below
        } else {
            gradient();                                                 // :615
            kick(-1);                                                   // :618-620
            for (int ls = 0; ls < Lmax; ++ls) {
                drift(ls);
                gradient();
                kick(ls);
                if (reversal_wanted && ls < Lmax - 1) {
                    // has the direction reversed (:633-638)?  inner += pNew[j]*momentum[j], dimension order; the
                    // starting momentum was parked in p.pn
                    const double inner = ordered_sum_tiles([&](int t, int ct, int r) {
                        return pn[t][ct][r] * p.pn[(size_t)comp(t, r) * NP + base + 16 * ct + c];
                    });
                    if (summer && ls < Lmine - 1 && !(inner >= 0.0)) status = 2;
                }
            }
        }
        if (summer && p.adaptive) {
            const double my_eps = (lane < 16) ? eps[0] : eps[1];        // chain base + lane is this lane's chain tile lane >> 4
            hmc_retune_after_leapfrog(status, my_eps, mean_eps, lfrog, reversal);   // :302-323
        }

        // ---- proposed potential and kinetic energy (:326-327), dimension order ----
        const double u1 = potential();
        const double ke1 = kinetic();

        // ---- Hamiltonian test (:333-387), one lane per chain decides ----
        if (summer) {
            const uint32_t gid = p.chain_offset + (uint32_t)mychain;
            const smcmc_u32x4 ablk = smcmc_draw_block(p.seed, gid, step, (ew + 1u) >> 2, SMCMC_STREAM_HMC);
            const double uacc = smcmc_u01(smcmc_select_word(ablk, (ew + 1u) & 3u));
            pot_prop = u1;
            const double delta = (pot_prop + ke1) - (pot_acc + ke0);
            const double trial = -smcmc_log_pos(uacc);
            const bool reject = (delta > trial) || !__builtin_isfinite(delta) || !(mychain < p.nchains);
            // UpdateCovariance runs on a finite proposal (okLeap is never zero); otherwise the step length shrinks (:336-344)
            contributes = __builtin_isfinite(pot_prop) ? 1 : 0;
            if (p.adaptive && !contributes && mean_eps > 0) mean_eps = 0.3 * mean_eps;
            verdict[lane] = reject ? 0 : 1;
            if (!reject) {
                pot_acc = pot_prop;
                acceptance = (acceptance * 4999.0 + 1.0) / 5000.0;      // :386
                ++naccept;
            } else {
                acceptance = (acceptance * 4999.0) / 5000.0;            // :367
            }
            last_accept = reject ? 0 : 1;
        }
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const bool take = verdict[16 * ct + c] != 0;
            const int chain = base + 16 * ct + c;
#pragma unroll
            for (int t = 0; t < kMfTI; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = comp(t, r);
                    if (owns(t) && i < D && chain < p.nchains) {
                        if (p.adaptive) p.qprev[(size_t)i * NP + chain] = p.q[(size_t)i * NP + chain];   // what :338 folds
                        if (take) {                                     // :380-383
                            p.q[(size_t)i * NP + chain] = p.qn[(size_t)i * NP + chain];
                            p.pm[(size_t)i * NP + chain] = pn[t][ct][r];
                        } else {                                        // :364-366
                            p.pm[(size_t)i * NP + chain] = -p.pm[(size_t)i * NP + chain];
                        }
                    }
                }
        }
        __syncthreads();
    }

    if (summer && mychain < p.nchains) {
        p.lane_f64[SMCMC_LANE_LOGL * NP + mychain] = -pot_acc;
        p.lane_f64[SMCMC_LANE_LOGL_PROPOSED * NP + mychain] = -pot_prop;
        p.lane_f64[SMCMC_LANE_ACCEPTANCE * NP + mychain] = acceptance;
        p.lane_i32[SMCMC_LANE_NACCEPT * NP + mychain] = naccept;
        p.lane_i32[SMCMC_LANE_LAST_ACCEPT * NP + mychain] = last_accept;
        p.lane_i32[SMCMC_LANE_TRIALS * NP + mychain] = trials;
        if (p.adaptive) {
            p.lane_f64[kHmcLaneMeanEpsilon * NP + mychain] = mean_eps;
            p.lane_f64[kHmcLaneReversalLen * NP + mychain] = reversal;
            p.lane_i32[kHmcLaneLeapfrog * NP + mychain] = lfrog;
            p.lane_i32[kHmcLaneContributes * NP + mychain] = contributes;
        }
    }
}

hipError_t launch_hmc_mfma(const HmcParams& p, hipStream_t stream);
hipError_t launch_hmc_matrix_exact(const HmcParams& p, hipStream_t stream);

}  // namespace smcmc
