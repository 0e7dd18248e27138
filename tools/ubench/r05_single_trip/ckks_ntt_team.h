// ckks_ntt_team.h — EXPERIMENT (round 5, lf_tune LF_TUNE_NTT_TEAMS): the two passes of an exact logN-16 forward transform in
// ONE launch, the intermediate of a limb handed from its column phase to its tile phase through the L2 of one XCD.
//
// A limb of 2^16 words is 512 KiB — three times the LDS of a CU — so the transform is a column phase (the 4 leading stages as
// one radix-16 register step per column) and a tile phase (12 stages per 4096-word tile); as two launches the intermediate
// makes a round trip through HBM (2 x 16 N bytes per limb instead of 16 N).  Here a TEAM of 16 persistent workgroups — placed on
// one XCD: blocks b, b + 8, b + 16, .. share an XCD, and the placement is VERIFIED at run time from HW_REG_XCC_ID — owns a limb
// at a time: each member runs one column unit (256 columns x 16 words), the team meets at a counter in that XCD's L2, each
// member runs one tile.  The eight teams of an XCD work on the same limb row of different polynomials (twiddle rows shared in
// L2), every XCD takes an eighth of the batch.  In flight between the phases: at most 16 x 32 KiB per team in a 4 MiB L2.
//
// Visibility inside a team that sits on one XCD: column-phase stores are complete (s_waitcnt vmcnt(0): acknowledged by L2)
// before the member's arrival — an L2 atomic — and the tile phase reads with nontemporal loads, which bypass the CU's L1 and
// are served by that same L2.  A team whose members report different XCC ids (the hardware promises no placement) falls
// back, for the whole launch, to agent-scope release / acquire fences around the meeting: correct anywhere, slower.
// The launch needs every workgroup resident (grid <= what the occupancy query admits; checked by the host side).
#pragma once
#include "ckks_ntt_core.h"
#include "ckks_ntt_tile16.h"

namespace {

#define TEAM_SIZE 16
#define TEAM_MAX 256
#define TEAM_PAD 64              // a team's words sit 256 bytes from the next team's: its pollers and arrivals have their own L2 line
                                 // and channel (all 64 counters in two lines: every poll of an XCD on ONE channel, 4.2 ms per step)
struct TeamCtl {
    unsigned arrive[TEAM_MAX * TEAM_PAD];   // [team * TEAM_PAD]: monotonic arrival counter; [team * TEAM_PAD + 1]: XCC mask
    unsigned timeout;            // set by a member that gave up waiting (bounded spins): the host reports failure
    unsigned pad_;
    unsigned long long stats[TEAM_MAX][TEAM_SIZE][4];   // per member: s_memtime ticks in column phases, meetings, tile phases; [3] = fast
};

__device__ __forceinline__ unsigned xcc_id() {
    // s_getreg_b32 HW_REG_XCC_ID (id 20), offset 0, size 4
    return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xf;
}

// one meeting of the team: every member calls it once per job; returns when all 16 have.  fast = the team sits on one XCD.
__device__ __forceinline__ bool team_meet(TeamCtl *ctl, int team, unsigned target, bool fast, i64 *sm) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's column stores have reached L2
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned seen;
        if (fast) {
            __hip_atomic_fetch_add(&ctl->arrive[team * TEAM_PAD], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // executes in this XCD's L2
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&ctl->arrive[team * TEAM_PAD], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned spins = 0;
        for (;;) {
            seen = __hip_atomic_load(&ctl->arrive[team * TEAM_PAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: bypasses L1
            if (seen >= target) break;
            __builtin_amdgcn_s_sleep(16);
            if (++spins > (1u << 24)) {   // ~ seconds: a member is not resident, give up loudly instead of hanging the box
                __hip_atomic_store(&ctl->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        if (!fast) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        reinterpret_cast<unsigned *>(sm + NTT16_FLAG)[1] = seen >= target ? 1u : 0u;
    }
    __syncthreads();
    return reinterpret_cast<const unsigned *>(sm + NTT16_FLAG)[1] != 0;
}

// the jobs of one arithmetic class: rows of `rl` in order, this team's polynomials of each
// CP = column-phase flavours (fwd_cols_body's POL: load | store << 3), LD / ST = tile-phase load / store flavours
template <bool DP, int K, int CP, int LD, int ST>
__device__ __forceinline__ bool team_rows(i64 *sm, i64 *a, const PassGeom &gc, const PassGeom &gt, const RowList &rl, int li0,
                                          int poly0, int poly_step, int poly_end, int member, TeamCtl *ctl, int team, bool fast,
                                          unsigned &meetings, const i64 *__restrict__ psi_br, const double *__restrict__ psi_dp,
                                          const i64 *__restrict__ ql, const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                          const i64 *__restrict__ kh) {
    constexpr int chunks = (1 << (16 - K)) / NTT_COL_THREADS;   // 128-column chunks per limb (K = 4: 32)
    for (int li = 0; li < rl.n; ++li) {
        const int crow = __builtin_amdgcn_readfirstlane((int)rl.id[li]);
        for (int poly = poly0; poly < poly_end; poly += poly_step) {
            // column phase: the two 128-column chunks 2 * member, 2 * member + 1 of limb (poly, crow)
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            {
                const int chunk = 2 * member + ((int)threadIdx.x >> 7);
                // fwd_cols_body's block numbering; wave-uniform (a wave lies inside one half of the block): pinned to an SGPR
                const int b = __builtin_amdgcn_readfirstlane(((li0 + li) * gc.batch + poly) * chunks + chunk);
                fwd_cols_body<DP, K, false, CP>(b, a, gc, rl, psi_br, psi_dp, nullptr, ql, qh, kl, kh);
            }
            meetings += TEAM_SIZE;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            if (!team_meet(ctl, team, meetings, fast, sm)) return false;
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
            // tile phase: tile `member` of the same limb (nothing of it is kept alive across the column phase: the kernel has
            // the registers of ONE phase — the column step alone needs 133 VGPRs on its own)
            {
                Ctx c;
                c.m = load_mod(ql, qh, kl, kh, crow);
                c.tw_mont = psi_br + ((i64)crow << gt.logN);
                set_aux<DP>(c, psi_dp, crow, gt.logN);
                c.d = DP ? make_dp_tab(c.m, c.tw_dp) : make_dp(c.m);
                c.relaxed = 0;
                c.inv_reduce = 0;
                fwd_tile16<DP, false, LD, ST>(sm, a + ((i64)(poly * gt.rows + crow) << gt.logN), member, gt, c);
            }
            lds_barrier();   // the waves' store spans are rewritten by the next tile's first exchange (and the flag word above)
            if (threadIdx.x == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long t3 = __builtin_amdgcn_s_memtime();
                ctl->stats[team][member][0] += t1 - t0, ctl->stats[team][member][1] += t2 - t1, ctl->stats[team][member][2] += t3 - t2;
                ctl->stats[team][member][3] = fast ? 1 : 0;
            }
        }
    }
    return true;
}

// grid = 8 * TEAM_SIZE * teams_per_xcd workgroups of 256 threads, all resident.  Team of block b: XCD lane b & 7, group
// (b >> 3) / 16, member (b >> 3) % 16.  XCD lane x owns polynomials [x * ppx, (x + 1) * ppx), its group g takes g, g + tpx, ..
template <int K, int CP, int LD, int ST>
__global__ void __launch_bounds__(NTT16_THREADS, 4) ntt_fwd_teams(i64 *a, PassGeom gc, PassGeom gt, ClassLists cl, int ppx,
                                                                    TeamCtl *ctl, const i64 *__restrict__ psi_br,
                                                                    const double *__restrict__ psi_dp,
                                                                    const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                    const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT16_LDS_WORDS + 1];
    const int xl = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
    const int tpx = (int)(gridDim.x >> 3) / TEAM_SIZE;
    const int g = slot / TEAM_SIZE, member = slot % TEAM_SIZE;
    const int team = g * 8 + xl;
    // membership: where do the 16 members really run?  (agent scope: nothing is known about the placement yet)
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_or(&ctl->arrive[team * TEAM_PAD + 1], 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }
    unsigned meetings = TEAM_SIZE;
    if (!team_meet(ctl, team, meetings, false, sm)) return;
    if (threadIdx.x == 0) {
        const unsigned m = __hip_atomic_load(&ctl->arrive[team * TEAM_PAD + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        reinterpret_cast<unsigned *>(sm + NTT16_FLAG)[1] = (m & (m - 1)) == 0 ? 1u : 0u;
    }
    __syncthreads();
    const bool fast = reinterpret_cast<const unsigned *>(sm + NTT16_FLAG)[1] != 0;
    __syncthreads();
    const int poly0 = xl * ppx + g, poly_end = (xl + 1) * ppx < gc.batch ? (xl + 1) * ppx : gc.batch;
    // integer-class rows first (fwd_cols_body numbers the blocks of a class list from 0: li0 = 0 for both lists)
    if (!team_rows<false, K, CP, LD, ST>(sm, a, gc, gt, cl.in, 0, poly0, tpx, poly_end, member, ctl, team, fast, meetings, psi_br, psi_dp, ql, qh, kl, kh))
        return;
    team_rows<true, K, CP, LD, ST>(sm, a, gc, gt, cl.dp, 0, poly0, tpx, poly_end, member, ctl, team, fast, meetings, psi_br, psi_dp, ql, qh, kl, kh);
}

// ---- the SKEWED form: a member runs the column unit of job i + 1 BEFORE the tile of job i, so the peers' column units of job i
// were finished a whole phase ago when it looks at the flag — no member ever waits in the steady state.  The
// price: two limbs per team are between their phases instead of one.  Arrivals are counted per job (ring of four slots: a
// member is at most two jobs ahead of the slowest one).
struct TeamJob {
    int dp, li, poly, crow;
};
__device__ __forceinline__ TeamJob team_job(int i, const ClassLists &cl, int ppt, int poly0, int step) {
    TeamJob j;
    const int nin = cl.in.n * ppt;
    j.dp = i >= nin;
    const int k = j.dp ? i - nin : i;
    j.li = k / ppt;
    j.poly = poly0 + (k % ppt) * step;
    j.crow = __builtin_amdgcn_readfirstlane((int)(j.dp ? cl.dp.id[j.li] : cl.in.id[j.li]));
    j.li = __builtin_amdgcn_readfirstlane(j.li), j.poly = __builtin_amdgcn_readfirstlane(j.poly), j.dp = __builtin_amdgcn_readfirstlane(j.dp);
    return j;
}

template <bool DP, int K>
__device__ __forceinline__ void team_col_unit(i64 *a, const PassGeom &gc, const RowList &rl, const TeamJob &j, int member,
                                              const i64 *__restrict__ psi_br, const double *__restrict__ psi_dp,
                                              const i64 *__restrict__ ql, const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                              const i64 *__restrict__ kh) {
    constexpr int chunks = (1 << (16 - K)) / NTT_COL_THREADS;
    const int chunk = 2 * member + ((int)threadIdx.x >> 7);
    const int b = __builtin_amdgcn_readfirstlane((j.li * gc.batch + j.poly) * chunks + chunk);
    fwd_cols_body<DP, K>(b, a, gc, rl, psi_br, psi_dp, nullptr, ql, qh, kl, kh);
}

template <bool DP>
__device__ __forceinline__ void team_tile_unit(i64 *sm, i64 *a, const PassGeom &gt, const TeamJob &j, int member,
                                               const i64 *__restrict__ psi_br, const double *__restrict__ psi_dp,
                                               const i64 *__restrict__ ql, const i64 *__restrict__ qh, const i64 *__restrict__ kl,
                                               const i64 *__restrict__ kh) {
    Ctx c;
    c.m = load_mod(ql, qh, kl, kh, j.crow);
    c.tw_mont = psi_br + ((i64)j.crow << gt.logN);
    set_aux<DP>(c, psi_dp, j.crow, gt.logN);
    c.d = DP ? make_dp_tab(c.m, c.tw_dp) : make_dp(c.m);
    c.relaxed = 0;
    c.inv_reduce = 0;
    fwd_tile16<DP, false>(sm, a + ((i64)(j.poly * gt.rows + j.crow) << gt.logN), member, gt, c);
}

template <int K>
__global__ void __launch_bounds__(NTT16_THREADS, 4) ntt_fwd_teams_skew(i64 *a, PassGeom gc, PassGeom gt, ClassLists cl, int ppx,
                                                                         TeamCtl *ctl, const i64 *__restrict__ psi_br,
                                                                         const double *__restrict__ psi_dp,
                                                                         const i64 *__restrict__ ql, const i64 *__restrict__ qh,
                                                                         const i64 *__restrict__ kl, const i64 *__restrict__ kh) {
    __shared__ i64 sm[NTT16_LDS_WORDS + 1];
    const int xl = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
    const int tpx = (int)(gridDim.x >> 3) / TEAM_SIZE;
    const int g = slot / TEAM_SIZE, member = slot % TEAM_SIZE;
    const int team = g * 8 + xl;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_or(&ctl->arrive[team * TEAM_PAD + 1], 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }
    if (!team_meet(ctl, team, TEAM_SIZE, false, sm)) return;
    if (threadIdx.x == 0) {
        const unsigned m = __hip_atomic_load(&ctl->arrive[team * TEAM_PAD + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        reinterpret_cast<unsigned *>(sm + NTT16_FLAG)[1] = (m & (m - 1)) == 0 ? 1u : 0u;
    }
    __syncthreads();
    const bool fast = reinterpret_cast<const unsigned *>(sm + NTT16_FLAG)[1] != 0;
    __syncthreads();
    const int poly0 = xl * ppx + g, poly_end = (xl + 1) * ppx < gc.batch ? (xl + 1) * ppx : gc.batch;
    const int ppt = poly0 < poly_end ? (poly_end - poly0 + tpx - 1) / tpx : 0;
    const int njobs = (cl.in.n + cl.dp.n) * ppt;
    unsigned *slots = &ctl->arrive[team * TEAM_PAD + 4];
    unsigned long long tw = 0;
    auto col = [&](int i) {
        const TeamJob j = team_job(i, cl, ppt, poly0, tpx);
        if (j.dp) team_col_unit<true, K>(a, gc, cl.dp, j, member, psi_br, psi_dp, ql, qh, kl, kh);
        else team_col_unit<false, K>(a, gc, cl.in, j, member, psi_br, psi_dp, ql, qh, kl, kh);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's column stores have reached L2
        __syncthreads();
        if (threadIdx.x == 0) {
            if (fast) __hip_atomic_fetch_add(&slots[i & 3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(&slots[i & 3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    if (njobs > 0) col(0);
    for (int i = 0; i < njobs; ++i) {
        if (i + 1 < njobs) col(i + 1);
        if (threadIdx.x == 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            const unsigned target = TEAM_SIZE * (unsigned)(i / 4 + 1);
            unsigned spins = 0, seen;
            for (;;) {
                seen = __hip_atomic_load(&slots[i & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (seen >= target) break;
                __builtin_amdgcn_s_sleep(16);
                if (++spins > (1u << 24)) {
                    __hip_atomic_store(&ctl->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            if (!fast) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            reinterpret_cast<unsigned *>(sm + NTT16_FLAG)[1] = seen >= target ? 1u : 0u;
            tw += __builtin_amdgcn_s_memtime() - t0;
        }
        __syncthreads();
        if (reinterpret_cast<const unsigned *>(sm + NTT16_FLAG)[1] == 0) return;
        const TeamJob j = team_job(i, cl, ppt, poly0, tpx);
        if (j.dp) team_tile_unit<true>(sm, a, gt, j, member, psi_br, psi_dp, ql, qh, kl, kh);
        else team_tile_unit<false>(sm, a, gt, j, member, psi_br, psi_dp, ql, qh, kl, kh);
        lds_barrier();
    }
    if (threadIdx.x == 0) ctl->stats[team][member][1] = tw, ctl->stats[team][member][3] = fast ? 1 : 0;
}

}  // namespace
