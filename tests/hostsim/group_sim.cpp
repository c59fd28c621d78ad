// tests/hostsim/group_sim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// The ClearPath search of a GROUP (permafrost-engine_amd/csrc/agent_group.h: clearpath_grp<G> with everything under
// it -- cone construction, ranks, projections, the column phase, the queue and its branch-free form, cone compaction,
// the retry shortcut cp_jump and its branch-free candidate phase, the replay of removals) compiled for the host and
// run on the lockstep emulator of wave_emu.h: the SAME source the kernels k_cp_rows (G = 16) and k_cp_heavy /
// k_agent_full (G = 64, one wave per problem) execute, checked against the reference build without a GPU.  Nothing of
// this is linked into libnavhip.so.
#define NH_HOSTSIM 1
#include "wave_emu.h"
#include "agent_group.h"

namespace {
template <int G> struct job {
    cpent e; v2 des; int nd, ns; cp_lds<G> *S; v2 out;
};
template <int G> void body(void *p)
{
    job<G> *J = (job<G>*)p;
    const v2 r = clearpath_grp<G>(J->e, J->des, J->nd, J->ns, *J->S);
    if(grp<G>::lane() == 0) J->out = r;
}
template <int G> int run_all(int nq, const float *ent, const float *des_v, const float *dyn, const int32_t *n_dyn,
                             const float *stat, const int32_t *n_stat, float *out, long *collectives)
{
    cp_lds<G> *S = new cp_lds<G>();
    long total = 0;
    for(int q = 0; q < nq; q++) {
        job<G> J;
        J.e.pos = mkv(ent[5 * q], ent[5 * q + 1]); J.e.vel = mkv(ent[5 * q + 2], ent[5 * q + 3]); J.e.radius = ent[5 * q + 4];
        J.des = mkv(des_v[2 * q], des_v[2 * q + 1]);
        J.nd = n_dyn[q]; J.ns = n_stat[q]; J.S = S; J.out = mkv(0, 0);
        if(J.nd + J.ns > G || J.nd > 32 || J.ns > 32) { delete S; return 2; }
        memset((void*)S, 0xff, sizeof(*S));                                  // (nothing may be read before it is written)
        for(int i = 0; i < J.nd * 5; i++) S->dyn[i] = dyn[(size_t)q * 160 + i];    // as k_clearpath<G> loads them
        for(int i = 0; i < J.ns * 5; i++) S->stat[i] = stat[(size_t)q * 160 + i];
        long c = 0;
        const char *err = emu::run(G, body<G>, &J, &c);
        if(err) { fprintf(stderr, "group_sim: problem %d: %s\n", q, err); delete S; return 1; }
        total += c;
        out[2 * q] = J.out.x; out[2 * q + 1] = J.out.z;
    }
    if(collectives) *collectives = total;
    delete S;
    return 0;
}
}  // namespace

// ---- the velocity step of a whole snapshot through the device's own group code --------------------------------------
// The spatial hash is built serially here (the four k_sp_* passes are plain scatter / scan code with GPU tests of their
// own); everything per agent is the device source: pool_record, nbr_walk_row on a 16-lane group, mid_thread,
// cp_load_lists + clearpath_grp on a 16- or 64-lane group (the split of k_agent_mid's work lists), post_thread.
// The cohesion term is an input (k_cohesion is a kernel of its own).  out_disp[uid] = DISP_* of agent_thread.h;
// agents whose neighbour walk is "irregular" (DISP_FULL: the wave-per-agent gather of k_agent_full) are not stepped.
#include <algorithm>
#include <cmath>
#include <vector>
static const double k_exp2_64_g[64] = { NH_EXP2_64_TABLE };

struct groupsim_map {
    int32_t chunk_w, chunk_h;
    const uint8_t  *cost[NAVHIP_NAV_LAYER_MAX];
    const uint16_t *blockers[NAVHIP_NAV_LAYER_MAX];
};

namespace {
struct walk_job { const nh_grid *G; int k; float smf; float2 *terms; const nh_nbr *NB; };
void walk_body(void *p)
{
    walk_job *J = (walk_job*)p;
    nbr_walk_row(*J->G, J->k, J->smf, k_exp2_64_g, J->terms, *J->NB);
}
template <int G> struct step_job {
    const nh_step_params *P; const nh_nbr *NB; int uid, nd, ns; cpent e; v2 des; cp_lds<G> *S; v2 out;
};
template <int G> void step_body(void *p)
{
    step_job<G> *J = (step_job<G>*)p;
    cp_load_lists<G>(J->P->grid, *J->NB, J->uid, J->nd, J->ns, *J->S);
    const v2 r = clearpath_grp<G>(J->e, J->des, J->nd, J->ns, *J->S);
    if(grp<G>::lane() == 0) J->out = r;
}
}  // namespace

extern "C" int groupsim_agent_step(const groupsim_map *map, const navhip_world *w, const float *coh_xz,
                                   const navhip_step_out *out, uint8_t *out_disp, long *collectives)
{
    nh_step_params P;
    memset(&P, 0, sizeof(P));
    P.map.w = map->chunk_w; P.map.h = map->chunk_h;
    for(int l = 0; l < NAVHIP_NAV_LAYER_MAX; l++) {
        P.map.layers[l].cost = map->cost[l];
        P.map.layers[l].blockers = map->blockers[l];
    }
    P.map_x = w->map_pos_x; P.map_z = w->map_pos_z;
    P.n_ents = w->n_ents; P.n_flocks = w->n_flocks; P.hz = w->hz;
    P.work_begin = w->work_begin; P.work_end = w->work_end;
    if(P.work_begin == 0 && P.work_end == 0) P.work_end = w->n_ents;
    P.pos_xz = w->pos_xz; P.vel_xz = w->vel_xz; P.radius = w->radius; P.max_speed = w->max_speed;
    P.speed = w->speed; P.flags = w->flags; P.state = w->state; P.has_dest_los = w->has_dest_los;
    P.flock = w->flock; P.vdes_xz = w->vdes_xz; P.flock_target_xz = w->flock_target_xz;
    P.flock_offsets = w->flock_offsets; P.flock_members = w->flock_members;
    P.flock_field_slot = w->flock_field_slot; P.field_pool = w->field_pool;
    P.form_ready = w->form_ready; P.cell_pos_xz = w->cell_pos_xz;
    P.form_cohesion_xz = w->form_cohesion_xz; P.form_align_xz = w->form_align_xz;
    P.form_drag_xz = w->form_drag_xz;
    P.arrival_sink_xz = w->arrival_sink_xz; P.arrival_flags = w->arrival_flags;
    const int n = w->n_ents;
    nh_grid &G = P.grid;
    G.origin_x = (int32_t)lrintf(w->grid_xmin * 256.0f); G.origin_y = (int32_t)lrintf(w->grid_zmin * 256.0f);
    const int32_t span_x = (int32_t)lrintf(w->grid_xmax * 256.0f) - G.origin_x;
    const int32_t span_y = (int32_t)lrintf(w->grid_zmax * 256.0f) - G.origin_y;
    G.grid_w = std::max(1, (int)(((uint32_t)span_x + 4095u) >> 12));
    G.grid_h = std::max(1, (int)(((uint32_t)span_y + 4095u) >> 12));
    G.n = n;
    const int ncells = G.grid_w * G.grid_h;
    std::vector<int32_t> cell(n), cell_start(ncells + 1, 0), pool_of(n, -1), fill(ncells, 0);
    for(int i = 0; i < n; i++) {
        cell[i] = sp_cell_of(G, bg_scale(w->pos_xz[2 * i]), bg_scale(w->pos_xz[2 * i + 1]));
        cell_start[cell[i] + 1]++;
    }
    for(int c = 0; c < ncells; c++) cell_start[c + 1] += cell_start[c];
    std::vector<float4> recA(n);
    std::vector<float2> recV(n);
    const nh_pack_src src = {w->vel_xz, w->radius, w->flags, w->state, w->arrival_sink_xz, w->arrival_flags};
    for(int i = n - 1; i >= 0; i--) {                 // descending uid inside every cell (bg_insert + bg_cleanup)
        const int slot = cell_start[cell[i]] + fill[cell[i]]++;
        pool_record(i, w->pos_xz, src, P.work_begin, P.work_end, recA[slot], recV[slot]);
        pool_of[i] = slot;
    }
    G.cell_start = cell_start.data(); G.recA = recA.data(); G.recV = recV.data(); G.pool_of = pool_of.data();

    std::vector<float2> sep(n);
    std::vector<uint32_t> cnt(n, 0);
    std::vector<float> rec((size_t)64 * 5 * n, 0.0f);
    nh_nbr NB = {sep.data(), cnt.data(), rec.data(), 64 * 5};
    const float smf = (float)((double)(0.75f / (float)P.hz) * 20.0);
    const double thresh = ((double)(0.75f / (float)P.hz) * 20.0) * 0.01;
    long total = 0, c = 0;
    float2 terms[16];
    for(int k = 0; k < n; k++) {
        if(nh_f2u(recA[k].w) & NH_PB_IDLE) continue;
        walk_job J = {&G, k, smf, terms, &NB};
        const char *err = emu::run(16, walk_body, &J, &c);
        if(err) { fprintf(stderr, "group_sim: neighbour walk of pool slot %d: %s\n", k, err); return 1; }
        total += c;
    }
    nh_step_outs O = {out->vel_xz, out->new_pos_xz, out->vdes_xz, out->vpref_xz, out->status};
    cp_lds<16> *S16 = new cp_lds<16>();
    cp_lds<64> *S64 = new cp_lds<64>();
    int rc = 0;
    for(int uid = P.work_begin; uid < P.work_end && !rc; uid++) {
        nh_mid_rec R;
        v2 out_vel;
        const int disp = mid_thread(P, uid, NB, coh_xz, smf, thresh, R, out_vel);
        out_disp[uid] = (uint8_t)disp;
        if(O.vdes_xz)  { O.vdes_xz[2 * uid] = R.vdes[0]; O.vdes_xz[2 * uid + 1] = R.vdes[1]; }
        if(O.vpref_xz) { O.vpref_xz[2 * uid] = R.vpref[0]; O.vpref_xz[2 * uid + 1] = R.vpref[1]; }
        const v2 me = mkv(P.pos_xz[2 * uid], P.pos_xz[2 * uid + 1]);
        if(disp == DISP_DONE) {
            post_thread(P, uid, me, P.state[uid], P.flags[uid], P.radius[uid], out_vel, R.vel_cap, R.status, O);
            continue;
        }
        if(disp < DISP_ROW0 || disp > DISP_HEAVY) continue;        // DISP_FULL: the irregular gather, not stepped here
        const uint32_t cn = cnt[uid];
        const int nd = (int)(cn & 0xff), ns = (int)((cn >> 8) & 0xff);
        cpent e; e.pos = me; e.vel = mkv(P.vel_xz[2 * uid], P.vel_xz[2 * uid + 1]); e.radius = P.radius[uid];
        v2 nv;
        const char *err;
        if(disp <= DISP_ROW3) {
            step_job<16> J = {&P, &NB, uid, nd, ns, e, mkv(R.vpref[0], R.vpref[1]), S16, mkv(0, 0)};
            memset((void*)S16, 0xff, sizeof(*S16));
            err = emu::run(16, step_body<16>, &J, &c);
            nv = J.out;
        }else{
            step_job<64> J = {&P, &NB, uid, nd, ns, e, mkv(R.vpref[0], R.vpref[1]), S64, mkv(0, 0)};
            memset((void*)S64, 0xff, sizeof(*S64));
            err = emu::run(64, step_body<64>, &J, &c);
            nv = J.out;
        }
        if(err) { fprintf(stderr, "group_sim: search of entity %d: %s\n", uid, err); rc = 1; break; }
        total += c;
        post_thread(P, uid, me, P.state[uid], P.flags[uid], e.radius, nv, R.vel_cap, R.status, O);
    }
    delete S16; delete S64;
    if(collectives) *collectives = total;
    return rc;
}

// [k] = problems that returned in attempt k (k = 7: seven or more), [8] = total attempts (agent_group.h: nh_cp_attempts)
extern "C" void groupsim_attempts(unsigned long long out[9], int reset)
{
    memcpy(out, nh_cp_attempts, sizeof(nh_cp_attempts));
    if(reset) memset(nh_cp_attempts, 0, sizeof(nh_cp_attempts));
}

// one problem searched by a TEAM of CP_TEAM waves (what k_cp_heavy runs below CP_SOLO_MIN problems, k_clearpath_team)
namespace {
enum { CP_TEAM = 4 };
struct team_job { cpent e; v2 des; int nd, ns; cp_lds<64> *S; cp_team *T; v2 out; const float *dyn, *stat; };
void team_body(void *p)
{
    team_job *J = (team_job*)p;
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    cp_lds<64> &S = J->S[wib];
    for(int i = lane; i < J->nd * 5; i += 64) S.dyn[i] = J->dyn[i];
    for(int i = lane; i < J->ns * 5; i += 64) S.stat[i] = J->stat[i];
    wave_sync();
    const v2 r = clearpath_grp<64, true>(J->e, J->des, J->nd, J->ns, S, wib, CP_TEAM, J->T);
    if(threadIdx.x == 0) J->out = r;
}
}  // namespace

extern "C" int groupsim_clearpath_team(int nq, const float *ent, const float *des_v, const float *dyn, const int32_t *n_dyn,
                                       const float *stat, const int32_t *n_stat, float *out, long *collectives)
{
    cp_lds<64> *S = new cp_lds<64>[CP_TEAM];
    cp_team T;
    long total = 0, c = 0;
    for(int q = 0; q < nq; q++) {
        team_job J;
        J.e.pos = mkv(ent[5 * q], ent[5 * q + 1]); J.e.vel = mkv(ent[5 * q + 2], ent[5 * q + 3]); J.e.radius = ent[5 * q + 4];
        J.des = mkv(des_v[2 * q], des_v[2 * q + 1]);
        J.nd = n_dyn[q]; J.ns = n_stat[q]; J.S = S; J.T = &T; J.out = mkv(0, 0);
        J.dyn = dyn + (size_t)q * 160; J.stat = stat + (size_t)q * 160;
        if(J.nd > 32 || J.ns > 32) { delete[] S; return 2; }
        memset((void*)S, 0xff, sizeof(cp_lds<64>) * CP_TEAM);
        memset((void*)&T, 0xff, sizeof(T));
        const char *err = emu::run(64 * CP_TEAM, team_body, &J, &c);
        if(err) { fprintf(stderr, "group_sim: team problem %d: %s\n", q, err); delete[] S; return 1; }
        total += c;
        out[2 * q] = J.out.x; out[2 * q + 1] = J.out.z;
    }
    if(collectives) *collectives = total;
    delete[] S;
    return 0;
}

extern "C" int groupsim_clearpath(int G, int nq, const float *ent, const float *des_v, const float *dyn, const int32_t *n_dyn,
                                  const float *stat, const int32_t *n_stat, float *out, long *collectives)
{
    if(G == 16) return run_all<16>(nq, ent, des_v, dyn, n_dyn, stat, n_stat, out, collectives);
    if(G == 64) return run_all<64>(nq, ent, des_v, dyn, n_dyn, stat, n_stat, out, collectives);
    return 3;
}
