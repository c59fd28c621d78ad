// agent_group.h -- the neighbour walk and the ClearPath search for a GROUP of G lanes per agent
// (device only).  G = 16 (one DPP row; four agents per wave) for the common agent with up to 16
// ClearPath neighbours, G = 64 (the whole wave) for agents in a crowd.
//
// Why groups: 100 000 agents are 1 563 waves if a thread steps an agent -- 1.5 waves per SIMD, every
// dependent load and every long scalar chain exposed (measured: 100-150 us per kernel) -- and 100 000
// waves with most lanes idle if a wave does (round 1: 1 017 wave instructions per agent).  Sixteen
// lanes per agent are 25 000 waves, and the lane-parallel parts of the step (candidate tests, cones,
// ray pairs) are 16-64 wide for the typical agent.
//
// Everything a group does is "group uniform" control flow; cross-lane traffic stays inside the group
// (ballot bits of the group, shuffles with a group-relative source, xor butterflies below G).
#pragma once
#include "agent_thread.h"

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// A value that is the same in every lane of a wave-wide group BY CONSTRUCTION (read from LDS at a uniform
// address, the result of a reduction): tell the compiler, so that it lives in a scalar register and the loops
// it controls stay uniform -- otherwise one such value in a `break` makes the whole loop divergent, and every
// counter inside it a per-lane VGPR under exec masks (measured: 40 % of the search's instructions were SALU
// bookkeeping for branches that no lane ever takes differently).  Rows of 16 lanes: identity.
template <int G> __device__ __forceinline__ int   uni(int v)   { return G == 64 ? __builtin_amdgcn_readfirstlane(v) : v; }
template <int G> __device__ __forceinline__ float uni(float v) { return G == 64 ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))) : v; }

template <int G> struct grp {
    static __device__ __forceinline__ int lane() { return (int)(threadIdx.x & (G - 1)); }
    static __device__ __forceinline__ int base() { return (int)(threadIdx.x & 63 & ~(G - 1)); }
    static __device__ __forceinline__ unsigned long long ballot(bool p)
    {
        const unsigned long long m = __ballot(p);
        if(G == 64) return m;
        return (m >> base()) & ((1ull << (G & 63)) - 1ull);
    }
    static __device__ __forceinline__ bool any(bool p) { return ballot(p) != 0ull; }
    static __device__ __forceinline__ int shfl(int v, int src) { return __shfl(v, base() + src); }
    static __device__ __forceinline__ float shfl(float v, int src) { return __shfl(v, base() + src); }
    // lexicographic (key, idx) arg-min over the group; key = +inf means "no candidate"
    static __device__ __forceinline__ void argmin(float &key, int &idx)
    {
#pragma unroll
        for(int d = G / 2; d >= 1; d >>= 1) {
            const float ok = __shfl_xor(key, d);
            const int   oi = __shfl_xor(idx, d);
            const bool take = (ok < key) || (ok == key && oi < idx);
            if(take) { key = ok; idx = oi; }
        }
    }
};

// inclusive prefix sum inside every row of 16 lanes (DPP row_shr, zero fill)
__device__ __forceinline__ int row_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
    return v;
}

// ---------------------------------------------------------------------------------------------
// neighbour walk, 16 lanes per agent
// ---------------------------------------------------------------------------------------------
// A SEGMENT = the cells of one fine row inside one coarse block that the query box covers:
// contiguous in the cell-sorted pool.  An r = 30 box is at most 5 x 5 fine cells and touches at most
// 2 x 2 coarse blocks: at most 10 segments, in visiting order (coarse block row, coarse block column,
// fine row; bitmap_grid.h:1408-1466) one per lane.
struct seg_table { int incl, len, bo; int total; };

__device__ __forceinline__ seg_table row_segments(const nh_grid &G, const sp_extent &E, int gl)
{
    const int cxc_lo = E.cx_lo >> 3, cxc_hi = E.cx_hi >> 3;
    const int cyc_lo = E.cy_lo >> 3, cyc_hi = E.cy_hi >> 3;
    const int ncb = cxc_hi - cxc_lo + 1;                                  // 1 or 2 block columns
    const int nr0 = min(cyc_lo * 8 + 7, E.cy_hi) - E.cy_lo + 1;           // fine rows in the first block row
    const int nr1 = (cyc_hi > cyc_lo) ? E.cy_hi - (cyc_lo + 1) * 8 + 1 : 0;
    int cb = 0, fy = -1;
    if(gl < ncb * nr0) {
        cb = gl >= nr0 ? 1 : 0;
        fy = E.cy_lo + (gl - cb * nr0);
    }else if(gl < ncb * (nr0 + nr1)) {
        const int s = gl - ncb * nr0;
        cb = s >= nr1 ? 1 : 0;
        fy = (cyc_lo + 1) * 8 + (s - cb * nr1);
    }
    seg_table T; T.len = 0; int bb = 0;
    if(fy >= 0) {
        const int cxc = cxc_lo + cb;
        const int fx0 = max(cxc * 8, E.cx_lo), fx1 = min(cxc * 8 + 8, E.cx_hi + 1);
        bb = G.cell_start[fy * G.grid_w + fx0];
        T.len = G.cell_start[fy * G.grid_w + fx1] - bb;
    }
    T.incl = row_incl_scan(T.len);
    T.total = grp<16>::shfl(T.incl, 15);
    T.bo = bb - (T.incl - T.len);                         // pool index of candidate q of this segment: bo + q
    return T;
}

// pool index of candidate q (q < total): the first segment whose inclusive prefix exceeds q
__device__ __forceinline__ int row_candidate(const seg_table &T, int q)
{
    int lo = 0;
#pragma unroll
    for(int st = 8; st >= 1; st >>= 1) {
        const int pv = grp<16>::shfl(T.incl, lo + st - 1);
        if(pv <= q) lo += st;
    }
    return grp<16>::shfl(T.bo, lo & 15) + q;
}

// separation_force (movement.c:1690, r = 30, cap 128) and find_neighbours (movement.c:2768, r = 10, cap
// 512, 32 + 32) of the entity in pool slot k, in ONE walk: the r = 10 query is the r = 30 walk restricted
// to d <= 10 (its box lies inside the r = 30 box, the visiting key does not depend on the query, and an
// element of a cell outside the r = 10 box fails the distance test anyway).  Once the r = 30 query has
// stopped at its cap the walk restarts on the smaller r = 10 box.  Both caps bind where the reference's
// two separate queries stop.  terms: this group's 16 float2 of LDS.
// Left to the wave-per-agent path (NH_NB_IRREGULAR): a garrisoned entity among the hits
// (filter_garrisoned, position.c:100-119, permutes the list) and queries that take the reference's
// wide linear scan or miss the grid.
__device__ void nbr_walk_row(const nh_grid &G, int k, float scaled_max_force, const double *exp_tab,
                             float2 *terms, const nh_nbr &NB)
{
    typedef grp<16> g;
    const int gl = g::lane();
    const unsigned lt = (1u << gl) - 1u;
    const float4 me4 = G.recA[k];
    const uint32_t mybits = __float_as_uint(me4.w);
    const int uid = (int)(mybits >> NH_PB_UID_SHIFT);
    const v2 me = mkv(me4.x, me4.y);
    const float my_radius = me4.z;
    const int32_t icx = bg_scale(me.x), icy = bg_scale(me.z);
    const int32_t ir30 = bg_scale(30.0f), ir10 = bg_scale(10.0f);
    sp_extent E30, E10;
    if(!sp_query_extent(G, icx, icy, ir30, E30) || !sp_query_extent(G, icx, icy, ir10, E10)
    || E30.wide || E10.wide
    || (E30.cx_hi >> 3) - (E30.cx_lo >> 3) > 1 || (E30.cy_hi >> 3) - (E30.cy_lo >> 3) > 1) {
        if(gl == 0) NB.cnt[uid] = (NH_NB_IRREGULAR | NH_NB_DONE) << 16;
        return;
    }
    const int32_t lim30 = ir30 * ir30, lim10 = ir10 * ir10;
    int raw30 = 0, raw10 = 0, n_dyn = 0, n_stat = 0;
    bool irregular = false;
    f2 acc = {0.0f, 0.0f};
    bool r10_only = false;
    seg_table T = row_segments(G, E30, gl);
    for(int q0 = 0; ; q0 += 16) {
        if(q0 >= T.total) {
            break;
        }
        const int q = q0 + gl;
        const bool valid = q < T.total;
        const int kk = row_candidate(T, valid ? q : 0);
        float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        int32_t d2 = 0x7fffffff;
        if(valid) {
            c = G.recA[kk];
            // (elements clamped into a border cell may be far away: range-check before squaring in
            // 32 bits; r = 30: |d| <= 7680 + 4095 otherwise)
            const int32_t dx = bg_scale(c.x) - icx, dy = bg_scale(c.y) - icy;
            const bool nearby = (uint32_t)(dx + 32767) < 65535u && (uint32_t)(dy + 32767) < 65535u;
            if(nearby) d2 = dx * dx + dy * dy;
        }
        const unsigned m30 = r10_only ? 0u : (unsigned)g::ballot(d2 <= lim30);
        const unsigned m10 = (unsigned)g::ballot(d2 <= lim10);
        const bool acc30 = !r10_only && d2 <= lim30 && raw30 + __popc(m30 & lt) < NH_SEP_CAP;
        const bool acc10 = d2 <= lim10 && raw10 + __popc(m10 & lt) < NH_NEAR_CAP;
        const uint32_t bits = __float_as_uint(c.w);
        if(g::any((acc30 || acc10) && (bits & NH_PB_GARRISONED))) { irregular = true; break; }
        const bool other = kk != k && (bits & NH_PB_MOVABLE) && !((mybits ^ bits) & NH_PB_AIR);
        if(!r10_only && m30) {
            // separation terms lane-parallel, then added in candidate order (skipped candidates add
            // +0, which leaves the never-negative-zero running sum unchanged)
            v2 term = mkv(0.0f, 0.0f);
            if(acc30 && other) {
                v2 t2;
                if(separation_term(me, my_radius, mkv(c.x, c.y), c.z, exp_tab, t2)) term = t2;
            }
            terms[gl] = make_float2(term.x, term.z);
            wave_sync();
#pragma unroll
            for(int j = 0; j < 16; j += 2) {
                const float4 two = *(const float4*)&terms[j];
                acc = acc + f2{two.x, two.y};
                acc = acc + f2{two.z, two.w};
            }
            wave_sync();
        }
        {
            const bool want = acc10 && other && c.z != 0.0f;
            const bool is_s = want && (bits & NH_PB_STATIC), is_d = want && !(bits & NH_PB_STATIC);
            const unsigned ms = (unsigned)g::ballot(is_s), md = (unsigned)g::ballot(is_d);
            if(is_s) {
                const int p = n_stat + __popc(ms & lt);
                if(p < NH_MAX_NEIGHBOURS) nbr_store(NB, uid, 32 + p, c, make_float2(0.0f, 0.0f));   // :2820
            }
            if(is_d) {
                const int p = n_dyn + __popc(md & lt);
                if(p < NH_MAX_NEIGHBOURS) nbr_store(NB, uid, p, c, G.recV[kk]);
            }
            n_stat = min(NH_MAX_NEIGHBOURS, n_stat + __popc(ms));
            n_dyn = min(NH_MAX_NEIGHBOURS, n_dyn + __popc(md));
        }
        raw30 += __popc(m30);
        raw10 += __popc(m10);
        if(raw10 >= NH_NEAR_CAP && (r10_only || raw30 >= NH_SEP_CAP)) break;
        if(!r10_only && raw30 >= NH_SEP_CAP) {
            // the r = 30 query has stopped: what is left is the r = 10 query -- redo it on its own,
            // smaller box (same visiting order, so the lists come out the same)
            r10_only = true;
            raw10 = 0; n_dyn = 0; n_stat = 0;
            T = row_segments(G, E10, gl);
            q0 = -16;
        }
    }
    if(gl != 0) return;
    if(irregular) {
        NB.cnt[uid] = (NH_NB_IRREGULAR | NH_NB_DONE) << 16;
        return;
    }
    v2 sep = mkv(0.0f, 0.0f);
    if(raw30 > 0)                                   // `if(0 == num_near) return 0`, movement.c:1737
        sep = vtrunc(vscale(mkv(acc.x, acc.y), -1.0f), scaled_max_force);
    NB.sep[uid] = make_float2(sep.x, sep.z);
    NB.cnt[uid] = (uint32_t)n_dyn | ((uint32_t)n_stat << 8) | (NH_NB_DONE << 16);
}

// ---------------------------------------------------------------------------------------------
// ClearPath (clearpath.c) on a group of G lanes: G_ClearPath_NewVelocity :694
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool inside_pcr(const float4 *cones, int n_cones, v2 test)
{
    for(int c = 0; c < n_cones; c++)
        if(cone_contains(cones[2 * c], cones[2 * c + 1], test)) return true;
    return false;
}

#define NH_CP_PAD16 8      // float4s of padding per 16-lane scratch: k_cp_rows' workgroup has to own at least k_cp_heavy's LDS (hole inheritance)
// LDS scratch of one ClearPath problem on a group of G lanes (at most G neighbours in total):
//   cones   2 float4 per cone: {apex.x, apex.z, slope(left), slope(right)}, {left.x, left.z, right.x, right.z}
//   ord     cone slots, nearest neighbour first (the order of the inside-obstacle tests)
//   q*      queue of candidate points waiting for the inside-obstacle test (< 2 G pending)
//   col     the rays (columns) whose line passes close enough to des_v to still matter
//   tau/seq the retry shortcut (cp_jump)
//   dyn/stat  the two neighbour lists, 5 floats per entry (remove_furthest edits them)
template <int G> struct cp_lds {
    float4  cones[2 * G];
    int32_t ord[G];
    float   qx[2 * G], qz[2 * G], ql[2 * G];
    int32_t qi[2 * G];
    int32_t col[2 * G];              // rays (columns) in ascending key
    float   ckey[2 * G];             // key of every ray: a lower bound of the distance of its candidates to des_v
    int32_t tau[G], seq[G];          // cp_jump: removal time of every cone, the removal sequence
    float   dyn[(G < 32 ? G : 32) * 5], stat[(G < 32 ? G : 32) * 5];
    float4  tc[G == 64 ? 2 * G : (G == 16 ? NH_CP_PAD16 : 1)];  // wave-wide search: the cones in TEST order (cones[ord[k]] copied: one dependent LDS read less per test step)
};

// compute_vnew :368 keeps the first strictly-smaller distance in candidate order, i.e. the minimum
// of (distance, order index) over the candidates OUTSIDE the combined obstacle.  Consequences:
//  * candidates can be examined in any order;
//  * a candidate that cannot beat the best admissible one found so far needs no inside-obstacle
//    test at all (branch and bound): the distance costs ~15 instructions, the test up to 64 cones;
//  * the candidate of the ray pair (i, j) lies ON LINE j: C_InfiniteLineIntersection (collision.c:
//    820-852) evaluates line 2's own equation at the x it found (out.z = s2 (out.x - p2.x) + p2.z, or
//    out.x = p2.x for a vertical line 2), whatever the conditioning of the pair.  So once a bound
//    exists, a ray j whose line keeps a distance from des_v larger than the bound (plus a margin three
//    orders of magnitude above the rounding of that evaluation) cannot contribute through ANY pair
//    (i, j): the whole column is skipped.  In a crowd that leaves a few columns of 128;
//  * inside_pcr (:249) is an OR over the cones, so the cones may be tested in any order: nearest
//    neighbour first (the widest cones), and a lane whose candidate is decided takes the next
//    candidate from the queue instead of waiting for the slowest lane of its group -- in a crowd
//    nearly every candidate lies inside one of the first few cones.
// The bound is group uniform.  `found` = some candidate outside the obstacle has been seen (what the
// reference's `vec_size(&xpoints) == 0` asks); until then every candidate is tested, so that points
// whose distance is NaN or infinite still count.
struct cp_bound { float len; int idx; v2 pt; int nfound; int sb;
};
#define CP_COL_MARGIN 0.02f
#define CP_SMALL_RAYS 8          // up to this many rays (4 neighbours) a problem is searched without queue and bound

__device__ __forceinline__ bool cp_alive(const cp_bound &B, float len, int idx)
{
    return B.nfound == 0 || len < B.len || (len == B.len && idx < B.idx);
}

// the candidate a lane is testing: cone ord[ci] is next
// (no bool members: `nfound` counts and `ci < 0` means "no candidate" -- flags that live across the
// group-divergent loops as 1-bit lane masks were miscompiled at -O3 once cp_work was inlined: the later
// row of a wave lost its candidates.  Counters stay in VGPRs.)
struct cp_lane { v2 pt; int idx; float len; int ci; };

// Inside-obstacle tests with persistent lanes: every iteration each busy lane tests its candidate
// against ONE cone; a lane whose candidate is decided -- inside a cone: dropped; outside all cones: it
// becomes the bound if it beats it -- takes the next queued candidate.  Runs until the queue is empty
// (finish = false: candidates still in flight stay with their lanes for the next call) or until
// nothing is in flight either (finish = true).

// ---- the same two functions for a WAVE-wide group, written without per-lane branches ------------------------
// In cp_work / cp_push every `if` on a lane's own state (has it a candidate? is it inside?) is a region the wave
// enters with an exec mask -- two or three scalar instructions per region for bookkeeping, and a SIMD issues a
// scalar instruction in a slot it could have given a vector one: the search ran 0.55 scalar instructions per
// vector instruction and 3.6 cycles per instruction of either kind (profiles/archive/r04_*).  Here a lane's state changes
// through selects, the only branches are on wave-uniform values (ballots, queue counters), and the two rare
// expansions (an undecided fast cone test, a candidate beating the bound) sit behind one ballot each.
// Same decisions in the same order as cp_work: a lane without a candidate takes the next queued one, tests it
// against ONE cone per step, nearest first; inside -> dropped, outside all -> it may become the bound.
template <int G>
__device__ __forceinline__ void cp_work_bf(cp_lds<G> &S, const cpent &ent, int n_cones, int &qn_io, cp_lane &L,
                                           cp_bound &B, bool finish)
{
    typedef grp<G> g;
    const int gl = g::lane();
    const unsigned long long lt_mask = (1ull << gl) - 1ull;
    const int qn = uni<G>(qn_io);
    n_cones = uni<G>(n_cones);
    int head = 0;
    float Blen = uni<G>(B.len), Bpx = uni<G>(B.pt.x), Bpz = uni<G>(B.pt.z);
    int Bidx = uni<G>(B.idx), nfound = uni<G>(B.nfound);
    float px = L.pt.x, pz = L.pt.z, len = L.len;
    int idx = L.idx, ci = L.ci;
    for(;;) {
        if(head < qn) {
            const bool need = ci < 0;
            const unsigned long long mn = g::ballot(need);
            const int my = head + (int)__popcll(mn & lt_mask);
            const bool take = need & (my < qn);
            const int at = take ? my : 0;
            const float4 qe = ((const float4*)S.qx)[at];          // (the queue packs an entry into 16 bytes)
            const float qx = qe.x, qz = qe.y, ql = qe.z;
            const int qi = __float_as_int(qe.w);
            const bool alive = (nfound == 0) | (ql < Blen) | ((ql == Blen) & (qi < Bidx));
            px = take ? qx : px; pz = take ? qz : pz; len = take ? ql : len; idx = take ? qi : idx;
            ci = take ? (alive ? 0 : -1) : ci;
            head = min(qn, head + (int)__popcll(mn));
        }
        const unsigned long long busy = g::ballot(ci >= 0);
        if(busy == 0ull) {
            if(head >= qn) break;
            continue;
        }
        if(!finish && head >= qn) break;
        float4 A, Bc;
        if constexpr(G == 64) {
            const int ck = max(ci, 0);
            A = S.tc[2 * ck]; Bc = S.tc[2 * ck + 1];
        }else{
            const int ck = S.ord[max(ci, 0)];
            A = S.cones[2 * ck]; Bc = S.cones[2 * ck + 1];
        }
        const v2 pt = mkv(px, pz);
        int v = cone_test_bf(A, Bc, pt);
        if(g::ballot((v == 2) & (ci >= 0)) != 0ull) {
            NH_COLD_PATH();
            if(v == 2) v = cone_contains_exact(A, Bc, pt) ? 1 : 0;
        }
        const bool act = ci >= 0, in = v == 1;
        const int nci = ci + 1;
        const bool outside = act & !in & (nci >= n_cones);
        ci = act ? ((in | outside) ? -1 : nci) : ci;
        const unsigned long long mo = g::ballot(outside);
        if(mo != 0ull) {
            NH_COLD_PATH();
            float key = (outside && len == len) ? len : __builtin_inff();      // a NaN distance never wins
            int ki = outside ? idx : 0x7fffffff;
            const float mykey = key; const int myidx = ki;
            g::argmin(key, ki);
            key = uni<G>(key); ki = uni<G>(ki);
            const bool better = nfound == 0 || key < Blen || (key == Blen && ki < Bidx);
            nfound++;
            if(better && key < __builtin_inff()) {
                const int owner = __ffsll((unsigned long long)g::ballot(outside && myidx == ki && mykey == key)) - 1;
                Blen = key; Bidx = ki;
                Bpx = uni<G>(g::shfl(px - ent.pos.x, owner)); Bpz = uni<G>(g::shfl(pz - ent.pos.z, owner));
            }
            const bool alive2 = (len < Blen) | ((len == Blen) & (idx < Bidx));
            ci = (ci >= 0 && !alive2) ? -1 : ci;
        }
    }
    L.pt = mkv(px, pz); L.len = len; L.idx = idx; L.ci = ci;
    B.len = Blen; B.idx = Bidx; B.pt = mkv(Bpx, Bpz); B.nfound = nfound;
    qn_io = 0;
    wave_sync();
}

// push this lane's candidate (ok) onto the group's queue; the queue is worked off once G are waiting
template <int G>
__device__ __forceinline__ void cp_push_bf(cp_lds<G> &S, const cpent &ent, int n_cones, bool ok, v2 pt, int idx,
                                           float len, int &qn, cp_lane &L, cp_bound &B)
{
    typedef grp<G> g;
    const int gl = g::lane();
    const unsigned long long mk = g::ballot(ok);
    if(mk != 0ull) {
        if(ok) {
            const int at = qn + (int)__popcll(mk & ((1ull << gl) - 1ull));
            ((float4*)S.qx)[at] = make_float4(pt.x, pt.z, len, __int_as_float(idx));
        }
        qn = uni<G>(qn + (int)__popcll(mk));
        wave_sync();
        if(qn >= G) cp_work_bf<G>(S, ent, n_cones, qn, L, B, false);
    }
}

// attempts of G_ClearPath_NewVelocity's do-while per problem (remove_furthest retries), as a histogram:
// [k] = problems that returned in attempt k (k = 7: seven or more), [8] = total attempts
__device__ unsigned long long nh_cp_attempts[9];

// ---- the retry loop of G_ClearPath_NewVelocity (:704-713), without retrying -------------------------
// do { attempt; if found return; remove_furthest; } while(both lists non-empty): in a jam an agent can
// fail dozens of attempts, each a full search with one neighbour fewer.  But nothing about attempt t
// is unknown after attempt 0:
//  * the removal ORDER only depends on the distances and the list positions (first strict maximum in
//    scan order, vec_del moves the last entry into the hole): simulate it, tau(k) = the removal that
//    takes neighbour k out (BIG: it survives until the loop ends after T_end removals);
//  * the cone of a neighbour and the intersection point of two rays do not depend on the attempt;
//  * so a candidate p made of rays of neighbours a, b exists in attempts t < min(tau(a), tau(b)) and
//    is inside the obstacle while any cone containing it is still there: it is admissible exactly for
//    max{tau(c) : cone c contains p} <= t < min(tau(a), tau(b)); des_v itself from t_d = max tau of
//    the cones containing it.
// The first attempt that succeeds is the minimum of those start times (cones tested latest-removed
// first: the first one that contains p IS the maximum; a candidate is dropped as soon as a cone that
// outlives the best start so far contains it).  Returns that attempt number (the caller replays so
// many removals from S.seq and runs ONE ordinary attempt, which finds the admissible point and
// breaks ties exactly as the reference does), or -1: no attempt succeeds before a list runs empty.
// (part / nparts: this group's share of the candidates when a team searches; the team's answer is
// the minimum of the shares' results, BIG standing for "none")
// Step 4 of cp_jump (below) for a WAVE-wide group, in the layout and style of the search's column phase: lane = row
// (this lane's two rays and their removal times in registers), the column's ray uniform; a lane's candidate state
// changes through selects, the cones come from S.tc in the order they are tested (latest-removed first, S.col = their
// removal times), a queue entry is 16 bytes, and the only branches are on ballots and the queue counters.  Columns
// whose neighbour leaves with the first removal are skipped (their candidates exist in attempt 0 only).  `cur`: the
// best start so far (from des_v); returns the minimum over this share's candidates.
__device__ __forceinline__ int cp_jump_cands_bf(cp_lds<64> &S, const cpent &ent, v2 des_v, int n_cones, int cur,
                                                int part, int nparts)
{
    const int gl = (int)(threadIdx.x & 63);
    const unsigned long long lt_mask = (1ull << gl) - 1ull;
    const int BIG = 1 << 20;
    n_cones = uni<64>(n_cones); cur = uni<64>(cur);
    const int n_rays = 2 * n_cones;
    if(gl < n_cones) {
        const int so = S.ord[gl];
        S.tc[2 * gl] = S.cones[2 * so]; S.tc[2 * gl + 1] = S.cones[2 * so + 1];
        S.col[gl] = S.tau[so];
    }
    float rpx[2], rpz[2], rdx[2], rdz[2], rsl[2];
    int rtau[2];
#pragma unroll
    for(int h = 0; h < 2; h++) {
        const int r = gl + h * 64;
        rpx[h] = rpz[h] = rdx[h] = rdz[h] = rsl[h] = 0.0f; rtau[h] = 0;
        if(r < n_rays) {
            const float4 Ar = S.cones[r & ~1], Br = S.cones[r | 1];
            rpx[h] = Ar.x; rpz[h] = Ar.y;
            rdx[h] = (r & 1) ? Br.z : Br.x; rdz[h] = (r & 1) ? Br.w : Br.y;
            rsl[h] = (r & 1) ? Ar.w : Ar.z;
            rtau[h] = S.tau[r >> 1];
        }
    }
    wave_sync();
    const int nh = n_rays > 64 ? 2 : 1;
    const int n_mine = (n_rays - part + nparts - 1) / nparts;          // columns part, part + nparts, ...
    const int n_proj = part == 0 ? nh : 0;                              // (the projections: the first share's)
    const int n_pass = n_proj + n_mine * nh;
    int qn = 0;
    float px = 0.0f, pz = 0.0f;
    int end = 0, ci = -1;                                               // ci < 0: this lane holds no candidate
#pragma unroll 1
    for(int p = 0;; p++) {
        const bool gen_done = p >= n_pass || cur <= 1;
        if(!gen_done) {
            bool ok = false;
            v2 pt = mkv(0, 0);
            int e = 0;
            if(p < n_proj) {
                const int h = p;
                const v2 point = h ? mkv(rpx[1], rpz[1]) : mkv(rpx[0], rpz[0]);
                const v2 dir = h ? mkv(rdx[1], rdz[1]) : mkv(rdx[0], rdz[0]);
                pt = vadd(point, vscale(dir, vdot(dir, des_v)));
                e = (h ? rtau[1] : rtau[0]) - 1;
                ok = gl + h * 64 < n_rays;
            }else{
                const int q = p - n_proj;
                const int jc = nh == 2 ? q >> 1 : q, h = nh == 2 ? q & 1 : 0;
                const int j = jc * nparts + part;
                const int tj = uni<64>(S.tau[j >> 1]);
                if(tj <= 1) continue;
                const float4 Aj = S.cones[j & ~1], Bj = S.cones[j | 1];
                const v2 p2 = mkv(Aj.x, Aj.y), d2 = (j & 1) ? mkv(Bj.z, Bj.w) : mkv(Bj.x, Bj.y);
                const float s2 = (j & 1) ? Aj.w : Aj.z;
                const int i = gl + h * 64;
                const v2 p1 = h ? mkv(rpx[1], rpz[1]) : mkv(rpx[0], rpz[0]);
                const v2 d1 = h ? mkv(rdx[1], rdz[1]) : mkv(rdx[0], rdz[0]);
                const float s1 = h ? rsl[1] : rsl[0];
                e = min(h ? rtau[1] : rtau[0], tj) - 1;
                const bool mine = (i < n_rays) & (i != j) & (e >= 1);
                float len = 0.0f;
                bool slow = false;
                ok = ray_isect_bf(p1, d1, s1, p2, d2, s2, des_v, ent.pos, pt, len, slow);
                if(__ballot(slow & mine) != 0ull) {
                    NH_COLD_PATH();
                    if(slow & mine) ok = ray_isect(p1, d1, s1, p2, d2, s2, pt);
                }
                ok = ok & mine;
            }
            ok = ok & (e >= 1);                        // it must exist in some attempt after the first
            const unsigned long long mk = __ballot(ok);
            if(ok) {
                const int at = qn + (int)__popcll(mk & lt_mask);
                ((float4*)S.qx)[at] = make_float4(pt.x, pt.z, __int_as_float(e), 0.0f);
            }
            qn = uni<64>(qn + (int)__popcll(mk));
            wave_sync();
            if(qn < 64) continue;
        }
        // work the queue off (persistent lanes, one cone test per busy lane and step)
        int head = 0;
        for(;;) {
            const bool need = ci < 0;
            const unsigned long long mn = __ballot(need);
            const int my = head + (int)__popcll(mn & lt_mask);
            const bool take = need & (my < qn);
            const float4 qe = ((const float4*)S.qx)[take ? my : 0];
            px = take ? qe.x : px; pz = take ? qe.y : pz; end = take ? __float_as_int(qe.z) : end;
            ci = take ? 0 : ci;
            head = uni<64>(min(qn, head + (int)__popcll(mn)));
            if(__ballot(ci >= 0) == 0ull) break;       // (no lane holds one: the queue is empty as well)
            if(!gen_done && head >= qn) break;          // more to generate: the lanes keep what they hold
            const bool act = ci >= 0;
            const int lim = min(cur - 1, end);          // its start has to be <= lim to matter
            const int ck = max(ci, 0);
            const float4 A = S.tc[2 * ck], Bc = S.tc[2 * ck + 1];
            const int tcv = S.col[ck];
            const v2 pt = mkv(px, pz);
            int v = cone_test_bf(A, Bc, pt);
            if(__ballot((v == 2) & act) != 0ull) {
                NH_COLD_PATH();
                if(v == 2) v = cone_contains_exact(A, Bc, pt) ? 1 : 0;
            }
            const bool in = v == 1, dead = lim < 1;
            const int nci = ci + 1;
            const bool last = nci >= n_cones;
            // inside: the latest-removed cone around it (its start, if early enough); outside everything: cannot be
            // (attempt 0 failed), start 0 as the reference's loop would find
            int result = in ? (tcv <= lim ? tcv : BIG) : (last ? 0 : BIG);
            result = (act & !dead) ? result : BIG;
            ci = act ? ((dead | in | last) ? -1 : nci) : -1;
            if(__ballot(result < BIG) != 0ull) {
                int r = result;
#pragma unroll
                for(int d = 32; d >= 1; d >>= 1) r = min(r, __shfl_xor(r, d));
                cur = uni<64>(min(cur, r));
            }
        }
        qn = 0;
        wave_sync();
        if(gen_done) break;
    }
    return cur;
}

template <int G>
__device__ int cp_jump(cp_lds<G> &S, const cpent &ent, v2 des_v, bool have, bool isdyn, int k, bool use,
                       int slot, float dist, int n_dyn, int n_stat, int n_cones, int part, int nparts)
{
    typedef grp<G> g;
    const int gl = g::lane();
    const unsigned long long lt_mask = (1ull << gl) - 1ull;
    const int BIG = 1 << 20;
    // 1. the removal sequence
    int tau = BIG, pos = k, cd = n_dyn, cs = n_stat, step = 0;
    bool present = have;                           // still in its list
    do {
        const bool removable = present && dist == dist;        // NaN never passes `len > max_dist`
        float nk = removable ? -dist : __builtin_inff();
        int ni = removable ? (isdyn ? pos : cd + pos) : 0x7fffffff;
        const int myscan = ni;
        g::argmin(nk, ni);
        if(!(nk < __builtin_inff())) break;        // nothing left to remove
        const bool w_dyn = ni < cd;
        const int w_pos = w_dyn ? ni : ni - cd, last = (w_dyn ? cd : cs) - 1;
        const bool winner = removable && myscan == ni;
        if(gl == 0) S.seq[step] = (w_dyn ? 0 : 256) | w_pos;
        step++;
        if(present && !winner && isdyn == w_dyn && pos == last) pos = w_pos;
        if(winner) { present = false; tau = step; }
        if(w_dyn) cd--; else cs--;
    } while(cd > 0 && cs > 0 && step < G);
    const int t_end = step;                        // attempts exist for t < t_end
    if(t_end <= 1) return -1;
    // 2. tau per cone, cones ordered by tau descending
    wave_sync();
    if(use) S.tau[slot] = tau;
    wave_sync();
    if(use) {
        int rank = 0;
        for(int c = 0; c < n_cones; c++) {
            const int tc = S.tau[c];
            rank += (tc > tau || (tc == tau && c < slot)) ? 1 : 0;
        }
        S.ord[rank] = slot;
    }
    wave_sync();
    // 3. des_v itself
    const v2 des_ws = vadd(ent.pos, des_v);
    int td = 0;
    if(gl < n_cones && cone_contains(S.cones[2 * gl], S.cones[2 * gl + 1], des_ws)) td = S.tau[gl];
#pragma unroll
    for(int d = G / 2; d >= 1; d >>= 1) td = max(td, __shfl_xor(td, d));
    int cur = min(td, t_end);                      // best start so far (group uniform)
    // 4. the candidates: when does each become admissible?
    if constexpr(G == 64) {
        cur = cp_jump_cands_bf(S, ent, des_v, n_cones, cur, part, nparts);
        return cur < t_end ? cur : -1;
    }
    const int n_rays = 2 * n_cones, npairs = n_rays * n_rays;
    const float inv_nr = 1.0f / (float)n_rays;
    int qn = 0;
    v2 l_pt = mkv(0, 0); int l_end = 0, l_ci = -1;         // l_ci < 0: this lane holds no candidate
    for(int c0 = part * G; c0 < n_rays + npairs || qn > 0 || g::any(l_ci >= 0); c0 += nparts * G) {
        // generate (projections first, then the ordered pairs)
        const int c = c0 + gl;
        bool ok = false;
        v2 pt = mkv(0, 0);
        int end = 0;
        if(cur > 1) {
            if(c < n_rays) {
                const float4 Ai = S.cones[c & ~1], Bi = S.cones[c | 1];
                const v2 dir = (c & 1) ? mkv(Bi.z, Bi.w) : mkv(Bi.x, Bi.y), point = mkv(Ai.x, Ai.y);
                pt = vadd(point, vscale(dir, vdot(dir, des_v)));
                end = S.tau[c >> 1] - 1;
                ok = true;
            }else if(c < n_rays + npairs) {
                const int p = c - n_rays;
                int i = (int)((float)p * inv_nr);
                int j = p - i * n_rays;
                if(j < 0) { i--; j += n_rays; }
                if(j >= n_rays) { i++; j -= n_rays; }
                if(i != j) {
                    end = min(S.tau[i >> 1], S.tau[j >> 1]) - 1;
                    if(end >= 1 && end >= 0) {
                        const float4 Ai = S.cones[i & ~1], Bi = S.cones[i | 1];
                        const float4 Aj = S.cones[j & ~1], Bj = S.cones[j | 1];
                        const bool ri = i & 1, rj = j & 1;
                        ok = ray_isect(mkv(Ai.x, Ai.y), ri ? mkv(Bi.z, Bi.w) : mkv(Bi.x, Bi.y), ri ? Ai.w : Ai.z,
                                       mkv(Aj.x, Aj.y), rj ? mkv(Bj.z, Bj.w) : mkv(Bj.x, Bj.y), rj ? Aj.w : Aj.z,
                                       pt);
                    }
                }
            }
            ok = ok && end >= 1;                   // it must exist in some attempt after the first
        }
        const unsigned long long mk = g::ballot(ok);
        if(ok) {
            const int at = qn + __popcll(mk & lt_mask);
            S.qx[at] = pt.x; S.qz[at] = pt.z; S.qi[at] = end;
        }
        qn += __popcll(mk);
        wave_sync();
        const bool gen_done = c0 + nparts * G >= n_rays + npairs || cur <= 1;
        if(qn < G && !gen_done) continue;
        // work the queue off (persistent lanes, one cone test per busy lane and iteration)
        int head = 0;
        for(;;) {
            if(head < qn) {
                const bool need = l_ci < 0;
                const unsigned long long mn = g::ballot(need);
                if(mn) {
                    const int my = head + __popcll(mn & lt_mask);
                    if(need && my < qn) { l_pt = mkv(S.qx[my], S.qz[my]); l_end = S.qi[my]; l_ci = 0; }
                    head = min(qn, head + __popcll(mn));
                }
            }
            if(!g::any(l_ci >= 0)) { if(head >= qn) break; continue; }
            if(!gen_done && head >= qn) break;
            int result = -1;
            if(l_ci >= 0) {
                const int lim = min(cur - 1, l_end);          // its start has to be <= lim to matter
                if(lim < 1) {
                    l_ci = -1;
                }else{
                    const int cs_ = S.ord[l_ci];
                    const int tc = S.tau[cs_];
                    const bool in = cone_contains(S.cones[2 * cs_], S.cones[2 * cs_ + 1], l_pt);
                    l_ci++;
                    if(in) {
                        if(tc <= lim) result = tc;             // the latest-removed cone around it
                        l_ci = -1;
                    }else if(l_ci >= n_cones) {
                        result = 0; l_ci = -1;                 // (outside everything: cannot be, attempt 0 failed)
                    }
                }
            }
            if(g::any(result >= 0)) {
                int r = result >= 0 ? result : BIG;
#pragma unroll
                for(int d = G / 2; d >= 1; d >>= 1) r = min(r, __shfl_xor(r, d));
                cur = min(cur, r);
            }
        }
        qn = 0;
        wave_sync();
        if(gen_done) break;
    }
    return cur < t_end ? cur : -1;
}


// One problem searched by a TEAM: the waves of a workgroup (k_cp_heavy).  Every wave builds the same
// cones and the same column order in its own LDS copy, tests the projections, and takes the columns
// part, part + nparts, ... of that order with a bound of its own; the minima of (distance, order
// index) over the waves' admissible candidates combine to exactly the single-group result (team_min).
// Everything after that is decided on the combined result, so the waves stay in step: the retry
// shortcut (cp_jump over the waves' shares of the candidates, minimum of the start times), the replay
// of the removals (every wave on its own copy) and the attempt that succeeds, again shared.
// The exchange goes through cp_team in LDS, two workgroup barriers per combine.
#define CP_TEAM_MAX 4
struct cp_team { cp_bound B[CP_TEAM_MAX]; int t[CP_TEAM_MAX]; };

template <int G>
__device__ __forceinline__ void team_min(cp_bound &B, cp_team &T, int part, int nparts)
{
    if(grp<G>::lane() == 0) T.B[part] = B;
    __syncthreads();
    cp_bound best = T.B[0];
    int nfound = best.nfound;
    for(int w = 1; w < nparts; w++) {
        const cp_bound o = T.B[w];
        nfound += o.nfound;
        if(o.len < best.len || (o.len == best.len && o.idx < best.idx)) best = o;
    }
    best.nfound = nfound;
    B = best;
    __syncthreads();
}

// S.dyn / S.stat hold the neighbours (n_dyn + n_stat <= G, each <= 32).
//   TEAM   every wave of the workgroup calls this with the same problem in its own S, part = its wave
//          number, nparts = the waves of the workgroup, T = the team's exchange area
template <int G, bool TEAM = false>
__device__ v2 clearpath_grp(const cpent &ent, v2 des_v, int n_dyn, int n_stat, cp_lds<G> &S,
                            int part = 0, int nparts = 1, cp_team *T = nullptr)
{
    typedef grp<G> g;
    n_dyn = uni<G>(n_dyn); n_stat = uni<G>(n_stat);
    if(n_dyn + n_stat == 0) return des_v;          // no obstacle: inside_pcr of nothing is false
    const int gl = g::lane();
    const unsigned long long lt_mask = (1ull << gl) - 1ull;
    bool jumped = false;
    // at most 64 neighbours can be removed; the bound only guards against a NaN-poisoned input
    for(int guard = 0; guard < 66; guard++) {
        // ---- HRVOs for dynamic, VOs for static neighbours -> rays (rays_repr :291) -----------
        // lane = neighbour, dynamic ones first; same_position neighbours are skipped (:216-246)
        const bool isdyn = gl < n_dyn;
        const int  k = isdyn ? gl : gl - n_dyn;
        const bool have = gl < n_dyn + n_stat;
        cpent nb; nb.pos = mkv(0, 0); nb.vel = mkv(0, 0); nb.radius = 0;
        if(have) {
            const float *src = (isdyn ? S.dyn : S.stat) + 5 * k;
            nb.pos = mkv(src[0], src[1]); nb.vel = mkv(src[2], src[3]); nb.radius = src[4];
        }
        const bool use = have && !(vlen(vsub(nb.pos, ent.pos)) < CP_EPS);
        v2 apex = mkv(0, 0), left = mkv(0, 0), right = mkv(0, 0);
        float sl = 0.0f, sr = 0.0f;
        if(use) make_cone(ent, nb, isdyn, apex, left, right, sl, sr);
        const unsigned long long m = g::ballot(use);
        const int slot = __popcll(m & lt_mask);                // hrvos first, then vos, in order
        const int n_cones = __popcll(m);
        const int n_rays = 2 * n_cones;
        // test order of the cones: the one des_v lies DEEPEST inside first (the smaller of its distances to the two
        // side lines, negative outside).  Every candidate that matters lies within the bound of des_v, so the cones
        // that reach furthest around des_v contain most of them; the nearest-neighbour order this replaces needed
        // 7.8 tests per candidate against 5.8 (scripts/cp_model.py; any order gives the same answer: inside_pcr
        // is an OR over the cones).  NaN sorts last (a NaN key would tie with nothing and lose its rank).
        float ndist = __builtin_inff();
        if(use) {
            const v2 q0 = vsub(vadd(ent.pos, des_v), apex);
            const float dpl = q0.z * left.x - q0.x * left.z, dpr = -(q0.z * right.x - q0.x * right.z);
            const float depth = fminf(dpl, dpr);
            ndist = (dpl == dpl && dpr == dpr) ? -depth : 0x1p120f;
        }
        wave_sync();
        if(use) {
            S.cones[2 * slot]     = make_float4(apex.x, apex.z, sl, sr);
            S.cones[2 * slot + 1] = make_float4(left.x, left.z, right.x, right.z);
        }
        S.ql[gl] = ndist;                                   // (the queue is empty here)
        wave_sync();
        if(use) {
            // rank of this neighbour by distance among the neighbours that have a cone
            int rank = 0;
            for(int j = 0; j < n_dyn + n_stat; j++) {
                const float dj = S.ql[j];
                rank += (dj < ndist || (dj == ndist && j < gl)) ? 1 : 0;
            }
            S.ord[rank] = slot;
        }
        wave_sync();

        // des_v admissible as it is?  lane = cone
        const v2 des_ws = vadd(ent.pos, des_v);
        bool in = false;
        if(gl < n_cones) in = cone_contains(S.cones[2 * gl], S.cones[2 * gl + 1], des_ws);
        if(!g::any(in)) {
            if(gl == 0 && part == 0 && guard > 0) { atomicAdd(&nh_cp_attempts[guard < 7 ? guard : 7], 1ull); atomicAdd(&nh_cp_attempts[8], (unsigned long long)guard + 1); }
            return des_v;
        }

        if constexpr(G == 64) {
            // the cones in test order, for the test steps of the wave-wide search
            if(gl < n_cones) { const int so = S.ord[gl]; S.tc[2 * gl] = S.cones[2 * so]; S.tc[2 * gl + 1] = S.cones[2 * so + 1]; }
            wave_sync();
        }
        cp_bound B; B.len = __builtin_inff(); B.idx = 0x7fffffff; B.pt = mkv(0, 0); B.nfound = 0; B.sb = 0;
        const int npairs = n_rays * n_rays;
        if(!TEAM && n_rays <= CP_SMALL_RAYS) {
        // ---- few neighbours (most agents outside a crowd): no queue, no bound -- every lane works its
        // own few candidates through (point, distance, every cone), keeps its best, one arg-min at the
        // end.  A handful of independent chains per lane instead of a dozen group-wide round trips.
        int nf = 0;
        float blen = __builtin_inff(); int bidx = 0x7fffffff; v2 bpt = mkv(0, 0);
        for(int c = gl; c < npairs + n_rays; c += G) {
            bool ok = false;
            v2 pt = mkv(0, 0);
            if(c >= npairs) {
                const int r = c - npairs;
                const float4 Ai = S.cones[r & ~1], Bi = S.cones[r | 1];
                const v2 dir = (r & 1) ? mkv(Bi.z, Bi.w) : mkv(Bi.x, Bi.y), point = mkv(Ai.x, Ai.y);
                pt = vadd(point, vscale(dir, vdot(dir, des_v)));
                ok = true;
            }else{
                const int i = c / n_rays, j = c - i * n_rays;
                if(i != j) {
                    const float4 Ai = S.cones[i & ~1], Bi = S.cones[i | 1];
                    const float4 Aj = S.cones[j & ~1], Bj = S.cones[j | 1];
                    const bool ri = i & 1, rj = j & 1;
                    ok = ray_isect(mkv(Ai.x, Ai.y), ri ? mkv(Bi.z, Bi.w) : mkv(Bi.x, Bi.y), ri ? Ai.w : Ai.z,
                                   mkv(Aj.x, Aj.y), rj ? mkv(Bj.z, Bj.w) : mkv(Bj.x, Bj.y), rj ? Aj.w : Aj.z,
                                   pt);
                }
            }
            if(ok) {
                bool inside = false;
                for(int k2 = 0; k2 < n_cones; k2++)
                    inside = inside || cone_contains(S.cones[2 * k2], S.cones[2 * k2 + 1], pt);
                if(!inside) {
                    const v2 curr = vsub(pt, ent.pos);
                    const float len = vlen(vsub(des_v, curr));
                    nf++;
                    if(len < blen || (len == blen && c < bidx)) { blen = len; bidx = c; bpt = curr; }
                }
            }
        }
        // (a NaN or infinite distance never wins: blen stays +inf there, the answer stays 0 as :368-386)
        float key = blen; int ki = bidx;
        g::argmin(key, ki);
        const int owner = __ffsll((unsigned long long)g::ballot(bidx == ki && blen == key)) - 1;
        B.nfound = g::any(nf > 0) ? 1 : 0;
        if(key < __builtin_inff()) {
            B.len = uni<G>(key); B.idx = uni<G>(ki);
            B.pt = mkv(uni<G>(g::shfl(bpt.x, owner)), uni<G>(g::shfl(bpt.z, owner)));
        }
        }else{
        cp_lane L; L.pt = mkv(0, 0); L.idx = 0; L.len = 0.0f; L.ci = -1;
        int qn = 0;                                            // pending candidates (group uniform)
        // ---- the projections of des_v on every ray (:344; order index npairs + ray).  Visited first:
        // they are the closest point of each ray, so the bound tightens at once.
        for(int c0 = 0; c0 < n_rays; c0 += G) {
            const int c = c0 + gl;
            bool ok = false;
            v2 pt = mkv(0, 0);
            float len = 0.0f;
            if(c < n_rays) {
                const float4 Ai = S.cones[c & ~1], Bi = S.cones[c | 1];
                const v2 dir = (c & 1) ? mkv(Bi.z, Bi.w) : mkv(Bi.x, Bi.y), point = mkv(Ai.x, Ai.y);
                const float plen = vdot(dir, des_v);
                pt = vadd(point, vscale(dir, plen));
                len = vlen(vsub(des_v, vsub(pt, ent.pos)));
                ok = cp_alive(B, len, npairs + c);
            }
            cp_push_bf<G>(S, ent, n_cones, ok, pt, npairs + c, len, qn, L, B);
        }
        cp_work_bf<G>(S, ent, n_cones, qn, L, B, true);

        // ---- the ray pairs (:321; order index i * n_rays + j), column by column, NEAREST LINE FIRST:
        // a column's candidates all lie on its line, so its key -- the distance of des_v to that line,
        // less the margins -- bounds them from below.  Columns are visited in ascending key; the search
        // stops at the first column whose key exceeds the bound: every later one is farther still.
        // In a crowd the best admissible velocity lies on one of the few lines next to des_v.
        unsigned long long cov0 = 0ull, cov1 = 0ull;           // rays whose column holds nothing admissible (sorted last)
        {
            // keys of this lane's (up to two) rays
            float mykey[2], mykeyc[2] = {__builtin_inff(), __builtin_inff()};
#pragma unroll
            for(int h = 0; h < 2; h++) {
                const int j = gl + h * G;
                mykey[h] = __builtin_inff();
                if(j < n_rays) {
                    const float4 Aj = S.cones[j & ~1], Bj = S.cones[j | 1];
                    const v2 dj = (j & 1) ? mkv(Bj.z, Bj.w) : mkv(Bj.x, Bj.y);
                    const v2 rel = vsub(des_v, vsub(mkv(Aj.x, Aj.y), ent.pos));
                    // distance of des_v to the RAY, not to its line: a candidate of column j passed the sign tests
                    // of C_RayRayIntersection2D (collision.c:862-871), so (pt - apex) agrees with the direction
                    // component by component (or is zero) -- its foot on line j has a parameter >= 0.  A ray that
                    // points away from des_v is as far from it as its apex is.  (The decision t >= 0 may be off by
                    // the rounding of t: there |rel| and the line distance differ by t^2 / 2 dist, far below the
                    // margins; the native square root's ulp hides in the factor.)
                    const float dline = fabsf(dj.x * rel.z - dj.z * rel.x);
                    const float tpar = dj.x * rel.x + dj.z * rel.z;
                    const float dapex = nh_sqrt_native(rel.x * rel.x + rel.z * rel.z) * 0.999f;
                    const float dist = (tpar >= 0.0f) ? dline : fmaxf(dline, dapex);
                    // prunable once dist > B.len + CP_COL_MARGIN + 2e-3 (B.len + |rel|_1): the second
                    // margin covers rays with |dir.x| < 1/1024, which are intersected as the exactly
                    // vertical line x = apex.x (collision.c:823-831), up to 1/1024 per unit of distance
                    // off the real line.  Solved for B.len, rounded down.
                    const float k = (dist - CP_COL_MARGIN - 2e-3f * (fabsf(rel.x) + fabsf(rel.z))) * 0.99f;
                    mykey[h] = (k == k && dline == dline) ? k : -__builtin_inff();   // NaN (fmaxf would hide it): never pruned
                    mykeyc[h] = mykey[h];
                    S.ckey[j] = mykey[h];
                }
            }
            float *keyc = (float*)S.tau;                       // (tau, seq: the retry shortcut's, free during the search)
#pragma unroll
            for(int h = 0; h < 2; h++) { const int j = gl + h * G; if(j < n_rays) keyc[j] = mykeyc[h]; }
            cov0 = g::ballot(mykeyc[0] == __builtin_inff() && gl < n_rays);
            cov1 = g::ballot(mykeyc[1] == __builtin_inff() && gl + G < n_rays);
            wave_sync();
#pragma unroll
            for(int h = 0; h < 2; h++) {
                const int j = gl + h * G;
                if(j < n_rays) {
                    int rank = 0;
                    for(int k = 0; k < n_rays; k++) {
                        const float kk = keyc[k];
                        rank += (kk < mykeyc[h] || (kk == mykeyc[h] && k < j)) ? 1 : 0;
                    }
                    S.col[rank] = j;
                }
            }
            wave_sync();
        }
        // ---- the cones that can still matter.  Every candidate that can beat the bound lies within B.len of
        // des_v; des_v is inside the obstacle.  A cone that contains such a point but not des_v has a side ray
        // crossing the segment between the two, i.e. within B.len of des_v (the region inside_pcr accepts is the
        // cone turned inwards by asin(1/1024) per side: r/1024 at distance r, inside the key's second margin):
        // that ray's column key is <= B.len.  So the inside-obstacle tests of the column phase only need the
        // cones that contain des_v or own a live column -- in a jam half of them -- in the same order.
        int n_test = n_cones;
        if(G == 64 && B.nfound) {
            bool relevant = false;
            if(gl < n_cones) relevant = in || !(S.ckey[2 * gl] > B.len) || !(S.ckey[2 * gl + 1] > B.len);
            const int slot_r = gl < n_cones ? S.ord[gl] : 0;                // rank gl -> cone slot
            const bool keep = gl < n_cones && (g::shfl((int)relevant, slot_r) != 0);
            const unsigned long long mk = g::ballot(keep);
            wave_sync();
            if(keep) S.ord[__popcll(mk & lt_mask)] = slot_r;
            n_test = __popcll(mk);
            wave_sync();
            if constexpr(G == 64) {
                if(gl < n_test) { const int so = S.ord[gl]; S.tc[2 * gl] = S.cones[2 * so]; S.tc[2 * gl + 1] = S.cones[2 * so + 1]; }
                wave_sync();
            }
        }
                              // keys, column order, cone compaction
        {
            // ---- lane = row: this lane's (up to two) rays stay in registers, the column's ray is the same for the
            // whole group.  No index arithmetic, a quarter of the LDS reads, and the bound is consulted per column.
            float rpx[2], rpz[2], rdx[2], rdz[2], rsl[2];
#pragma unroll
            for(int h = 0; h < 2; h++) {
                const int r = gl + h * G;
                rpx[h] = rpz[h] = rdx[h] = rdz[h] = rsl[h] = 0.0f;
                if(r < n_rays) {
                    const float4 Ar = S.cones[r & ~1], Br = S.cones[r | 1];
                    rpx[h] = Ar.x; rpz[h] = Ar.y;
                    rdx[h] = (r & 1) ? Br.z : Br.x; rdz[h] = (r & 1) ? Br.w : Br.y;
                    rsl[h] = (r & 1) ? Ar.w : Ar.z;
                }
            }
            const int n_mine = (n_rays - part + nparts - 1) / nparts;      // columns part, part + nparts, ...
            for(int jc = 0; jc < n_mine; jc++) {
                const int j = uni<G>(S.col[jc * nparts + part]);
                if(((j < G ? cov0 >> j : cov1 >> (j - G)) & 1ull) != 0ull) break;         // (covered columns sort last)
                if(B.nfound && uni<G>(S.ckey[j]) > B.len) break;
                const float4 Aj = S.cones[j & ~1], Bj = S.cones[j | 1];
                const v2 p2 = mkv(Aj.x, Aj.y), d2 = (j & 1) ? mkv(Bj.z, Bj.w) : mkv(Bj.x, Bj.y);
                const float s2 = (j & 1) ? Aj.w : Aj.z;
                const int nh = n_rays > G ? 2 : 1;
#pragma unroll 1
                for(int h = 0; h < nh; h++) {              // (not unrolled: ONE copy of the queue code in the loop)
                    const int i = gl + h * G;
                    const int idx = i * n_rays + j;
                    const v2 p1 = h ? mkv(rpx[1], rpz[1]) : mkv(rpx[0], rpz[0]);
                    const v2 d1 = h ? mkv(rdx[1], rdz[1]) : mkv(rdx[0], rdz[0]);
                    const float s1 = h ? rsl[1] : rsl[0];
                    bool ok = false;
                    v2 pt = mkv(0, 0);
                    float len = 0.0f;
                    {
                        bool slow = false;
                        ok = ray_isect_bf(p1, d1, s1, p2, d2, s2, des_v, ent.pos, pt, len, slow);
                        const bool mine = (i < n_rays) & (i != j);
                        if(g::ballot(slow & mine) != 0ull) {          // (a quotient that needs its division, an odd distance)
                            NH_COLD_PATH();
                            if(slow & mine) {
                                ok = ray_isect(p1, d1, s1, p2, d2, s2, pt);
                                len = vlen(vsub(des_v, vsub(pt, ent.pos)));
                            }
                        }
                        ok = ok & mine & cp_alive(B, len, idx);
                    }
                    cp_push_bf<G>(S, ent, n_test, ok, pt, idx, len, qn, L, B);
                }
            }
            cp_work_bf<G>(S, ent, n_test, qn, L, B, true);
        }
        }
        if(TEAM) team_min<G>(B, *T, part, nparts);
        if(B.nfound) {
            if(gl == 0 && part == 0 && guard > 0) { atomicAdd(&nh_cp_attempts[guard < 7 ? guard : 7], 1ull); atomicAdd(&nh_cp_attempts[8], (unsigned long long)guard + 1); }
            return B.pt;                   // (only NaN / infinite distances: ret stays 0, as :368-386)
        }

        // ---- no admissible point: remove_furthest (:390) and retry while both lists non-empty
        float dist = -__builtin_inff();
        if(have) dist = vlen(vsub(ent.pos, nb.pos));
        if(!jumped) {
            // the first failure: find the attempt that will succeed and go there in one step
            jumped = true;
            int t = uni<G>(cp_jump<G>(S, ent, des_v, have, isdyn, k, use, slot, dist, n_dyn, n_stat, n_cones, part, nparts));
            if(TEAM) {
                if(gl == 0) T->t[part] = t < 0 ? (1 << 20) : t;
                __syncthreads();
                t = T->t[0];
                for(int w = 1; w < nparts; w++) t = min(t, T->t[w]);
                if(t >= (1 << 20)) t = -1;
                t = uni<G>(t);
                __syncthreads();
            }
            if(t < 0) {
                if(gl == 0 && part == 0) { atomicAdd(&nh_cp_attempts[0], 1ull); atomicAdd(&nh_cp_attempts[8], 2ull); }
                return mkv(0.0f, 0.0f);
            }
            wave_sync();
            if(gl == 0) {
                int cd = n_dyn, cs = n_stat;
                for(int r = 0; r < t; r++) {
                    const int e = S.seq[r], p = e & 255;
                    if(e < 256) { cd--; for(int q = 0; q < 5; q++) S.dyn[5 * p + q] = S.dyn[5 * cd + q]; }
                    else        { cs--; for(int q = 0; q < 5; q++) S.stat[5 * p + q] = S.stat[5 * cs + q]; }
                }
            }
            wave_sync();
            for(int r = 0; r < t; r++) { if(uni<G>(S.seq[r]) < 256) n_dyn--; else n_stat--; }
            continue;
        }
        // (not reached in practice: the attempt after a jump succeeds; kept as the plain loop)
        // first strict maximum in scan order (dynamic entries precede static ones) == min over (-dist, lane)
        float nk = -dist; int ni = gl;
        if(!have || !(dist == dist)) { nk = __builtin_inff(); }   // NaN never passes `len > max_dist`
        g::argmin(nk, ni);
        nk = uni<G>(nk); ni = uni<G>(ni);
        wave_sync();
        if(nk < __builtin_inff() && gl == 0) {
            if(ni < n_dyn) { for(int q = 0; q < 5; q++) S.dyn[5 * ni + q] = S.dyn[5 * (n_dyn - 1) + q]; }
            else           { const int s = ni - n_dyn; for(int q = 0; q < 5; q++) S.stat[5 * s + q] = S.stat[5 * (n_stat - 1) + q]; }
        }
        wave_sync();
        if(nk < __builtin_inff()) { if(ni < n_dyn) n_dyn--; else n_stat--; }
        if(!(n_dyn > 0 && n_stat > 0)) {
            if(gl == 0 && part == 0) { atomicAdd(&nh_cp_attempts[0], 1ull); atomicAdd(&nh_cp_attempts[8], (unsigned long long)guard + 1); }
            return mkv(0.0f, 0.0f);
        }
    }
    return mkv(0.0f, 0.0f);
}

// ---------------------------------------------------------------------------------------------
// ClearPath for an agent with at most four neighbours on a row of 16 lanes, ONE attempt, no queue,
// no bound, no retry logic: the kernel of the sparse crowd (k_cp_small) -- few registers, 128 bytes
// of LDS per agent, twice the waves per SIMD of the general search.  Lane k < n holds neighbour k
// (dynamic ones first).  found = false: no admissible candidate (the caller hands the agent to the
// general search, which knows how to retry).  Same arithmetic, same order index, same tie-break as
// clearpath_grp.
// ---------------------------------------------------------------------------------------------
__device__ v2 clearpath_small_row(const cpent &ent, v2 des_v, const cpent &nb, bool isdyn, bool have,
                                  float4 *cones, bool &found)
{
    typedef grp<16> g;
    const int gl = g::lane();
    const unsigned long long lt_mask = (1ull << gl) - 1ull;
    found = true;
    const bool use = have && !(vlen(vsub(nb.pos, ent.pos)) < CP_EPS);
    v2 apex = mkv(0, 0), left = mkv(0, 0), right = mkv(0, 0);
    float sl = 0.0f, sr = 0.0f;
    if(use) make_cone(ent, nb, isdyn, apex, left, right, sl, sr);
    const unsigned long long m = g::ballot(use);
    const int slot = __popcll(m & lt_mask);
    const int n_cones = __popcll(m), n_rays = 2 * n_cones, npairs = n_rays * n_rays;
    wave_sync();
    if(use) {
        cones[2 * slot]     = make_float4(apex.x, apex.z, sl, sr);
        cones[2 * slot + 1] = make_float4(left.x, left.z, right.x, right.z);
    }
    wave_sync();
    const v2 des_ws = vadd(ent.pos, des_v);
    bool in = false;
    if(gl < n_cones) in = cone_contains(cones[2 * gl], cones[2 * gl + 1], des_ws);
    if(!g::any(in)) return des_v;

    int nf = 0;
    float blen = __builtin_inff(); int bidx = 0x7fffffff; v2 bpt = mkv(0, 0);
    for(int c = gl; c < npairs + n_rays; c += 16) {
        bool ok = false;
        v2 pt = mkv(0, 0);
        if(c >= npairs) {
            const int r = c - npairs;
            const float4 Ai = cones[r & ~1], Bi = cones[r | 1];
            const v2 dir = (r & 1) ? mkv(Bi.z, Bi.w) : mkv(Bi.x, Bi.y), point = mkv(Ai.x, Ai.y);
            pt = vadd(point, vscale(dir, vdot(dir, des_v)));
            ok = true;
        }else{
            const int i = c / n_rays, j = c - i * n_rays;
            if(i != j) {
                const float4 Ai = cones[i & ~1], Bi = cones[i | 1];
                const float4 Aj = cones[j & ~1], Bj = cones[j | 1];
                const bool ri = i & 1, rj = j & 1;
                ok = ray_isect(mkv(Ai.x, Ai.y), ri ? mkv(Bi.z, Bi.w) : mkv(Bi.x, Bi.y), ri ? Ai.w : Ai.z,
                               mkv(Aj.x, Aj.y), rj ? mkv(Bj.z, Bj.w) : mkv(Bj.x, Bj.y), rj ? Aj.w : Aj.z, pt);
            }
        }
        if(ok) {
            bool inside = false;
            for(int k2 = 0; k2 < n_cones; k2++)
                inside = inside || cone_contains(cones[2 * k2], cones[2 * k2 + 1], pt);
            if(!inside) {
                const v2 curr = vsub(pt, ent.pos);
                const float len = vlen(vsub(des_v, curr));
                nf++;
                if(len < blen || (len == blen && c < bidx)) { blen = len; bidx = c; bpt = curr; }
            }
        }
    }
    float key = blen; int ki = bidx;
    g::argmin(key, ki);
    found = g::any(nf > 0);
    v2 res = mkv(0.0f, 0.0f);                  // (only NaN / infinite distances: the answer stays 0, :368-386)
    if(key < __builtin_inff()) {
        const int owner = __ffsll((unsigned long long)g::ballot(bidx == ki && blen == key)) - 1;
        res = mkv(g::shfl(bpt.x, owner), g::shfl(bpt.z, owner));
    }
    return res;
}

// neighbour records (from the walk) -> S.dyn / S.stat: lane j copies record j (the lanes of a group read one
// contiguous run of the entity's row)
template <int G>
__device__ __forceinline__ void cp_load_lists(const nh_grid &Gd, const nh_nbr &NB, int uid, int n_dyn, int n_stat,
                                              cp_lds<G> &S)
{
    (void)Gd;
    const int gl = grp<G>::lane();
    wave_sync();
    if(gl < n_dyn + n_stat) {
        const bool isdyn = gl < n_dyn;
        const int j = isdyn ? gl : gl - n_dyn;
        const float *src = NB.rec + (size_t)uid * NB.stride + 5 * (isdyn ? j : 32 + j);
        float *dst = (isdyn ? S.dyn : S.stat) + 5 * j;
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3]; dst[4] = src[4];
    }
    wave_sync();
}
