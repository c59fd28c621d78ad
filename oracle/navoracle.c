/* oracle/navoracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's algorithm for the navigation hot path (SURVEY.md
 * section 8a), written from the reference's behaviour; every function cites the reference
 * file:line (under the reference's src/) it follows.  Sequential, one agent / one chunk field
 * at a time, same C types and expression order as the reference so that float results are
 * reproducible.
 *
 * Pinning: the reference has no golden vectors or unit tests for this path, so
 * this restatement is pinned against OUTPUTS OF THE REFERENCE ITSELF: oracle/_ref/libpfref.so
 * (the reference's own translation units compiled in place) in tests/test_oracle_cpu.py, and
 * the committed fixtures under tests/golden/ that scripts/make_golden.py generated from it.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.  It uses
 * the PODs of include/navhip.h (the boundary under test) for its inputs so that the same
 * request / snapshot records feed both sides; it shares no code with the product.
 *
 * Build: gcc -std=c99 -O2 -fPIC -shared -ffp-contract=off  (oracle/build_oracle.py)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdbool.h>
#include <pthread.h>
#include <time.h>

#include "navhip.h"

#define RES            64
#define CELLS          4096
#define COST_IMPASS    0xff
#define ISLAND_NONE    0xffff
#define FACTION_NONE   0xf
#define MAX_FACTIONS   15
#define NLAYERS        12

/* the planes of `struct nav_chunk` (nav_data.h:118-158) the path reads, per layer, in the packed
 * upload layout [h][w][64][64] (N_CopyCostBasePacked, nav.c:2432) */
typedef struct no_map {
    int32_t         w, h;
    const uint8_t  *cost[NLAYERS];
    const uint16_t *blockers[NLAYERS];
    const uint16_t *local_islands[NLAYERS];
    const uint8_t  *factions[NLAYERS];       /* [chunk][15][64][64] */
    const uint16_t *islands[NLAYERS];        /* global island ids (nav_chunk.islands) */
} no_map;

/* ===========================================================================================
 * chunk flow fields
 * =========================================================================================== */

/* field_tile_passable (field.c:117) / field_tile_passable_no_enemies (field.c:179) */
static bool tile_passable(const no_map *m, int layer, int chunk, int r, int c, int faction_id,
                          unsigned enemies)
{
    size_t i = ((size_t)chunk << 12) + (size_t)r * RES + c;
    if(m->cost[layer][i] == COST_IMPASS)
        return false;
    uint16_t blk = m->blockers[layer] ? m->blockers[layer][i] : 0;
    if(faction_id == FACTION_NONE)
        return blk == 0;
    bool enemies_only = true;
    if(m->factions[layer]) {
        const uint8_t *fp = m->factions[layer] + ((size_t)chunk * MAX_FACTIONS << 12) + r * RES + c;
        for(int f = 0; f < MAX_FACTIONS; f++) {
            if(fp[(size_t)f << 12] && !(enemies & (1u << f))) {
                enemies_only = false;
                break;
            }
        }
    }
    if(enemies_only)
        return true;
    return blk == 0;
}

/* binary min-heap on a float priority (pqueue.h:112-191); any correct PQ yields the same
 * integration values, ties included, because the values are a fixpoint of the relaxation */
typedef struct { float prio; uint16_t cell; } pq_ent;
typedef struct { pq_ent *a; int n, cap; } pq_t;

static void pq_push(pq_t *q, float prio, int cell)
{
    if(q->n == q->cap) {
        q->cap = q->cap ? q->cap * 2 : 256;
        q->a = realloc(q->a, (size_t)q->cap * sizeof(pq_ent));
    }
    int i = q->n++;
    while(i > 0) {
        int p = (i - 1) / 2;
        if(q->a[p].prio <= prio) break;
        q->a[i] = q->a[p];
        i = p;
    }
    q->a[i].prio = prio;
    q->a[i].cell = (uint16_t)cell;
}

static int pq_pop(pq_t *q)
{
    int ret = q->a[0].cell;
    pq_ent last = q->a[--q->n];
    int i = 0;
    for(;;) {
        int l = 2 * i + 1, r = l + 1, s = i;
        float best = last.prio;
        if(l < q->n && q->a[l].prio < best) { s = l; best = q->a[l].prio; }
        if(r < q->n && q->a[r].prio < best) { s = r; }
        if(s == i) break;
        q->a[i] = q->a[s];
        i = s;
    }
    if(q->n > 0) q->a[i] = last;
    return ret;
}

/* field_tile_adjacent_to_next_iid (field.c:1131): scan the tiles of the `next` portal for one
 * at Manhattan distance 1 (M_Tile_Distance, tile.c:414) that lies on local island next_iid */
static bool adjacent_to_next_iid(const no_map *m, const navhip_field_req *rq, int r, int c)
{
    const uint16_t *li = m->local_islands[rq->layer];
    int next_chunk = rq->next_chunk_r * m->w + rq->next_chunk_c;
    for(int r2 = rq->next_r0; r2 <= rq->next_r1; r2++) {
    for(int c2 = rq->next_c0; c2 <= rq->next_c1; c2++) {
        int dr = (rq->next_chunk_r * RES + r2) - (rq->chunk_r * RES + r);
        int dc = (rq->next_chunk_c * RES + c2) - (rq->chunk_c * RES + c);
        if(abs(dr) + abs(dc) == 1) {
            if(li[((size_t)next_chunk << 12) + r2 * RES + c2] == rq->next_iid)
                return true;
        }
    }}
    return false;
}

/* field_flow_dir (field.c:355): min over the 4 cardinals, diagonals admitted only when both
 * side tiles are finite, then first match in the order N,S,E,W,NW,NE,SW,SE */
static int flow_dir(const float *f, int r, int c)
{
    const int rdim = RES, cdim = RES;
    float min_cost = INFINITY;
#define F(rr, cc) f[(rr) * rdim + (cc)]
#define MINF(a, b) ((a) < (b) ? (a) : (b))
    if(r > 0)          min_cost = MINF(min_cost, F(r - 1, c));
    if(r < rdim - 1)   min_cost = MINF(min_cost, F(r + 1, c));
    if(c > 0)          min_cost = MINF(min_cost, F(r, c - 1));
    if(c < cdim - 1)   min_cost = MINF(min_cost, F(r, c + 1));
    if(r > 0 && c > 0 && F(r - 1, c) < INFINITY && F(r, c - 1) < INFINITY)
        min_cost = MINF(min_cost, F(r - 1, c - 1));
    if(r > 0 && c < cdim - 1 && F(r - 1, c) < INFINITY && F(r, c + 1) < INFINITY)
        min_cost = MINF(min_cost, F(r - 1, c + 1));
    if(r < rdim - 1 && c > 0 && F(r + 1, c) < INFINITY && F(r, c - 1) < INFINITY)
        min_cost = MINF(min_cost, F(r + 1, c - 1));
    if(r < rdim - 1 && c < cdim - 1 && F(r + 1, c) < INFINITY && F(r, c + 1) < INFINITY)
        min_cost = MINF(min_cost, F(r + 1, c + 1));

    if(r > 0 && F(r - 1, c) == min_cost)                          return NAVHIP_FD_N;
    else if(r < rdim - 1 && F(r + 1, c) == min_cost)              return NAVHIP_FD_S;
    else if(c < cdim - 1 && F(r, c + 1) == min_cost)              return NAVHIP_FD_E;
    else if(c > 0 && F(r, c - 1) == min_cost)                     return NAVHIP_FD_W;
    else if(r > 0 && c > 0 && F(r - 1, c - 1) == min_cost)        return NAVHIP_FD_NW;
    else if(r > 0 && c < cdim - 1 && F(r - 1, c + 1) == min_cost) return NAVHIP_FD_NE;
    else if(r < rdim - 1 && c > 0 && F(r + 1, c - 1) == min_cost) return NAVHIP_FD_SW;
    else if(r < rdim - 1 && c < rdim - 1 && F(r + 1, c + 1) == min_cost) return NAVHIP_FD_SE;
    return NAVHIP_FD_NONE;      /* the reference asserts here */
#undef F
#undef MINF
}

static int field_nearest_pathable(const no_map *m, const navhip_field_req *rq, uint8_t *inout_dirs,
                                  float *out_integ);
static int closest_tiles_local(const no_map *m, int layer, int chunk, int target, unsigned local_iid,
                               unsigned global_iid, int *out, int maxout);

/* N_FlowFieldInit (field.c:2020, unless NAVHIP_REQ_INOUT) + N_FlowFieldUpdate (field.c:2030) for
 * TARGET_TILE / TARGET_PORTAL.  Returns 0, or -1 for a malformed request. */
int no_field_update(const no_map *m, const navhip_field_req *rq, uint8_t *inout_dirs,
                    float *out_integ)
{
    if(rq->layer >= NLAYERS || !m->cost[rq->layer] || rq->chunk_r >= m->h || rq->chunk_c >= m->w)
        return -1;
    navhip_field_req live;
    if((rq->flags & NAVHIP_REQ_LIVE_IIDS) && rq->type == NAVHIP_TARGET_PORTAL && m->local_islands[rq->layer]) {
        /* island ids re-read from the current labels: the first tile of each portal that has one; a
         * portal without any leads nowhere and the request is skipped (include/navhip.h) */
        const uint16_t *li0 = m->local_islands[rq->layer];
        live = *rq;
        live.port_iid = live.next_iid = NAVHIP_ISLAND_NONE;
        const uint16_t *lp = li0 + ((size_t)(rq->chunk_r * m->w + rq->chunk_c) << 12);
        for(int r = rq->port_r0; r <= rq->port_r1 && live.port_iid == NAVHIP_ISLAND_NONE; r++)
            for(int c = rq->port_c0; c <= rq->port_c1 && live.port_iid == NAVHIP_ISLAND_NONE; c++)
                live.port_iid = lp[r * RES + c];
        const uint16_t *ln = li0 + ((size_t)(rq->next_chunk_r * m->w + rq->next_chunk_c) << 12);
        for(int r = rq->next_r0; r <= rq->next_r1 && live.next_iid == NAVHIP_ISLAND_NONE; r++)
            for(int c = rq->next_c0; c <= rq->next_c1 && live.next_iid == NAVHIP_ISLAND_NONE; c++)
                live.next_iid = ln[r * RES + c];
        if(live.port_iid == NAVHIP_ISLAND_NONE || live.next_iid == NAVHIP_ISLAND_NONE)
            return 0;
        rq = &live;
    }
    const int layer = rq->layer, chunk = rq->chunk_r * m->w + rq->chunk_c;
    const int faction_id = rq->faction_id;
    const unsigned enemies = rq->enemies;
    const uint8_t *cost = m->cost[layer] + ((size_t)chunk << 12);
    if(rq->type == NAVHIP_TARGET_NEAREST_PATHABLE)
        return field_nearest_pathable(m, rq, inout_dirs, out_integ);
    const bool island_nearest = (rq->flags & NAVHIP_REQ_ISLAND_NEAREST) != 0;

    if(!(rq->flags & NAVHIP_REQ_INOUT) && !island_nearest)
        memset(inout_dirs, NAVHIP_FD_NONE, CELLS);                     /* N_FlowFieldInit */

    static __thread float integ[CELLS];
    for(int i = 0; i < CELLS; i++) integ[i] = INFINITY;                /* field.c:2059-2063 */
    pq_t q = {0};
    static __thread int frontier0[CELLS];
    int nfront = 0;
#define SEED(i_) do { if(island_nearest) frontier0[nfront++] = (i_); \
                      else { pq_push(&q, 0.0f, (i_)); integ[(i_)] = 0.0f; } } while(0)

    /* field_initial_frontier, field.c:1372 */
    if(rq->type == NAVHIP_TARGET_TILE) {                               /* field.c:1096 */
        if(tile_passable(m, layer, chunk, rq->tile_r, rq->tile_c, faction_id, enemies)) {
            int i = rq->tile_r * RES + rq->tile_c;
            SEED(i);
        }
    }else if(rq->type == NAVHIP_TARGET_PORTAL) {                       /* field.c:1160 */
        const uint16_t *li = m->local_islands[layer];
        if(!li) { free(q.a); return -1; }
        for(int r = rq->port_r0; r <= rq->port_r1; r++) {
        for(int c = rq->port_c0; c <= rq->port_c1; c++) {
            if(!tile_passable(m, layer, chunk, r, c, faction_id, enemies))
                continue;
            if(rq->port_iid != ISLAND_NONE && li[((size_t)chunk << 12) + r * RES + c] != rq->port_iid)
                continue;
            if(!adjacent_to_next_iid(m, rq, r, c))
                continue;
            SEED(r * RES + c);
        }}
    }else{
        free(q.a);
        return -1;
    }
#undef SEED

    if(island_nearest) {
        /* N_FlowFieldUpdateIslandToNearest, field.c:2307: move the frontier onto the tiles of
         * local island aux_iid nearest to the target's own frontier */
        if(!m->local_islands[layer]) { free(q.a); return -1; }
        const unsigned local_iid = rq->aux_iid;
        const uint16_t *li = m->local_islands[layer] + ((size_t)chunk << 12);
        const uint16_t *gi = m->islands[layer] ? m->islands[layer] + ((size_t)chunk << 12) : NULL;
        if(nfront == 0 && rq->type == NAVHIP_TARGET_TILE)               /* ignoreblock retry :2352 */
            frontier0[nfront++] = rq->tile_r * RES + rq->tile_c;
        static __thread int newf[CELLS * 2], tmp[CELLS];
        int min_mh = 1 << 30, nnew = 0;
        for(int i = 0; i < nfront; i++) {
            int cur = frontier0[i];
            unsigned cur_giid = gi ? gi[cur] : ISLAND_NONE;
            if(li[cur] == local_iid) {
                if(min_mh > 0) nnew = 0;
                min_mh = 0;
                newf[nnew++] = cur;
                continue;
            }
            int nextra = closest_tiles_local(m, layer, chunk, cur, local_iid, cur_giid, tmp, CELLS);
            if(!nextra) continue;
            int mh = abs((tmp[0] >> 6) - (cur >> 6)) + abs((tmp[0] & 63) - (cur & 63));
            if(mh < min_mh) { min_mh = mh; nnew = 0; }
            if(mh > min_mh) continue;
            for(int k = 0; k < nextra && nnew < CELLS * 2; k++) newf[nnew++] = tmp[k];
        }
        for(int i = 0; i < nnew; i++) {
            pq_push(&q, 0.0f, newf[i]);
            integ[newf[i]] = 0.0f;
        }
    }

    /* field_build_integration, field.c:539: Dijkstra, 4-connected (field_neighbours_grid :203
     * skips diagonals), step cost = cost_base of the neighbour, passable neighbours only */
    static const int dr4[4] = {-1, 0, 0, 1}, dc4[4] = {0, -1, 1, 0};
    while(q.n > 0) {
        int cur = pq_pop(&q);
        int r = cur >> 6, c = cur & 63;
        for(int k = 0; k < 4; k++) {
            int nr = r + dr4[k], nc = c + dc4[k];
            if(nr < 0 || nr >= RES || nc < 0 || nc >= RES) continue;
            if(!tile_passable(m, layer, chunk, nr, nc, faction_id, enemies)) continue;
            float total = integ[cur] + (float)cost[nr * RES + nc];
            if(total < integ[nr * RES + nc]) {
                integ[nr * RES + nc] = total;
                pq_push(&q, total, nr * RES + nc);
            }
        }
    }
    free(q.a);

    /* field_build_flow, field.c:734: unreached cells are left untouched, cost-0 cells NONE */
    for(int r = 0; r < RES; r++) {
    for(int c = 0; c < RES; c++) {
        float v = integ[r * RES + c];
        if(v == INFINITY) continue;
        if(v == 0.0f) { inout_dirs[r * RES + c] = NAVHIP_FD_NONE; continue; }
        inout_dirs[r * RES + c] = (uint8_t)flow_dir(integ, r, c);
    }}

    /* field_fixup -> field_fixup_portal_edges, field.c:1408,830 */
    if(rq->type == NAVHIP_TARGET_PORTAL) {
        int d;
        if(rq->next_chunk_r < rq->chunk_r)      d = NAVHIP_FD_N;
        else if(rq->next_chunk_r > rq->chunk_r) d = NAVHIP_FD_S;
        else if(rq->next_chunk_c < rq->chunk_c) d = NAVHIP_FD_W;
        else                                    d = NAVHIP_FD_E;
        for(int i = 0; i < CELLS; i++)
            if(integ[i] == 0.0f) inout_dirs[i] = (uint8_t)d;
    }
    if(out_integ) memcpy(out_integ, integ, sizeof(float) * CELLS);
    return 0;
}

/* field_tile_passable (field.c:117): faction agnostic */
static bool plain_passable(const no_map *m, int layer, int chunk, int i)
{
    size_t k = ((size_t)chunk << 12) + i;
    if(m->cost[layer][k] == COST_IMPASS) return false;
    return !(m->blockers[layer] && m->blockers[layer][k] > 0);
}

/* N_FlowFieldUpdateToNearestPathable (field.c:2247): field_passable_frontier (:1441) BFS through the
 * non-passable region of `start`, field_build_integration_nonpass (:643), in-place bake of the
 * tiles with a finite non-zero value */
static int field_nearest_pathable(const no_map *m, const navhip_field_req *rq, uint8_t *inout_dirs,
                                  float *out_integ)
{
    const int layer = rq->layer, chunk = rq->chunk_r * m->w + rq->chunk_c;
    const uint8_t *cost = m->cost[layer] + ((size_t)chunk << 12);
    static __thread float integ[CELLS];
    static __thread uint8_t visited[CELLS];
    static __thread int queue[CELLS];
    for(int i = 0; i < CELLS; i++) integ[i] = INFINITY;
    memset(visited, 0, sizeof(visited));
    pq_t q = {0};
    int head = 0, tail = 0;
    int start = rq->tile_r * RES + rq->tile_c;
    queue[tail++] = start;
    visited[start] = 1;
    while(head < tail) {
        int cur = queue[head++];
        if(plain_passable(m, layer, chunk, cur)) {
            pq_push(&q, 0.0f, cur);
            integ[cur] = 0.0f;
            continue;
        }
        static const int d4[4][2] = {{0, -1}, {0, 1}, {-1, 0}, {1, 0}};
        for(int k = 0; k < 4; k++) {
            int nr = (cur >> 6) + d4[k][0], nc = (cur & 63) + d4[k][1];
            if(nr < 0 || nr >= RES || nc < 0 || nc >= RES) continue;     /* tile_outside_region */
            if(visited[nr * RES + nc]) continue;
            visited[nr * RES + nc] = 1;
            queue[tail++] = nr * RES + nc;
        }
    }
    static const int dr4[4] = {-1, 0, 0, 1}, dc4[4] = {0, -1, 1, 0};
    while(q.n > 0) {
        int cur = pq_pop(&q);
        int r = cur >> 6, c = cur & 63;
        for(int k = 0; k < 4; k++) {
            int nr = r + dr4[k], nc = c + dc4[k];
            if(nr < 0 || nr >= RES || nc < 0 || nc >= RES) continue;
            int ni = nr * RES + nc;
            if(plain_passable(m, layer, chunk, ni)) continue;
            float total = integ[cur] + (float)cost[ni];
            if(total < integ[ni]) {
                integ[ni] = total;
                pq_push(&q, total, ni);
            }
        }
    }
    free(q.a);
    for(int r = 0; r < RES; r++) {
    for(int c = 0; c < RES; c++) {
        float v = integ[r * RES + c];
        if(v == INFINITY || v == 0.0f) continue;
        inout_dirs[r * RES + c] = (uint8_t)flow_dir(integ, r, c);
    }}
    if(out_integ) memcpy(out_integ, integ, sizeof(float) * CELLS);
    return 0;
}

/* field_closest_tiles_local (field.c:1010): BFS over the whole chunk from `target` in Manhattan
 * order; the qualifying tiles at the first distance where any qualifies */
static int closest_tiles_local(const no_map *m, int layer, int chunk, int target, unsigned local_iid,
                               unsigned global_iid, int *out, int maxout)
{
    static __thread uint8_t visited[CELLS];
    static __thread int queue[CELLS];
    memset(visited, 0, sizeof(visited));
    const uint16_t *li = m->local_islands[layer] + ((size_t)chunk << 12);
    const uint16_t *gi = m->islands[layer] ? m->islands[layer] + ((size_t)chunk << 12) : NULL;
    int head = 0, tail = 0, ret = 0, first = -1;
    queue[tail++] = target;
    visited[target] = 1;
    while(head < tail) {
        int cur = queue[head++];
        static const int d4[4][2] = {{0, -1}, {0, 1}, {-1, 0}, {1, 0}};
        for(int k = 0; k < 4; k++) {
            int nr = (cur >> 6) + d4[k][0], nc = (cur & 63) + d4[k][1];
            if(nr < 0 || nr >= RES || nc < 0 || nc >= RES) continue;
            if(visited[nr * RES + nc]) continue;
            visited[nr * RES + nc] = 1;
            queue[tail++] = nr * RES + nc;
        }
        int mh = abs((cur >> 6) - (target >> 6)) + abs((cur & 63) - (target & 63));
        if(first > -1 && mh > first) break;
        if(!plain_passable(m, layer, chunk, cur)) continue;
        if(global_iid != ISLAND_NONE && gi && gi[cur] != global_iid) continue;
        if(local_iid != ISLAND_NONE && li[cur] != local_iid) continue;
        if(first == -1) first = mh;
        out[ret++] = cur;
        if(ret == maxout) break;
    }
    return ret;
}

int no_build_fields(const no_map *m, const navhip_field_req *reqs, int n, uint8_t *inout_dirs,
                    float *out_integ)
{
    for(int i = 0; i < n; i++) {
        int rc = no_field_update(m, &reqs[i], inout_dirs + (size_t)i * CELLS,
                                 out_integ ? out_integ + (size_t)i * CELLS : NULL);
        if(rc) return rc;
    }
    return 0;
}

/* ===========================================================================================
 * vec2 arithmetic (pf_math.c:58-94) -- float ops, sqrt evaluated in double then rounded
 * =========================================================================================== */
typedef struct { float x, z; } v2;

static v2 mkv(float x, float z) { v2 r = {x, z}; return r; }
static v2 vadd(v2 a, v2 b) { return mkv(a.x + b.x, a.z + b.z); }
static v2 vsub(v2 a, v2 b) { return mkv(a.x - b.x, a.z - b.z); }
static v2 vscale(v2 a, float s) { return mkv(a.x * s, a.z * s); }
static float vdot(v2 a, v2 b) { return a.x * b.x + a.z * b.z; }
static float vlen(v2 a) { return (float)sqrt(a.x * a.x + a.z * a.z); }
static v2 vnormal(v2 a) { float l = vlen(a); return mkv(a.x / l, a.z / l); }

/* vec2_truncate, movement.c:643 */
static v2 vtrunc(v2 a, float max_len)
{
    if(vlen(a) > max_len) {
        a = vnormal(a);
        a = vscale(a, max_len);
    }
    return a;
}

#define EPS (1.0 / 1024)      /* clearpath.c:76, collision.c EPSILON */

/* ===========================================================================================
 * region flow fields: field_build_integration_region (field.c:582) + field_build_flow_unaligned
 * (:800, N_CellArrivalFieldCreate :2445 / N_GroupArrivalFieldCreate :2525) or
 * field_build_flow_region (:763, field_update_enemies/entity/zone :1537,:1615,:1822)
 * =========================================================================================== */
static int flow_dir_dim(const float *f, int rdim, int cdim, int r, int c)
{
    float min_cost = INFINITY;
#define F(rr, cc) f[(rr) * rdim + (cc)]
#define MINF(a, b) ((a) < (b) ? (a) : (b))
    if(r > 0)          min_cost = MINF(min_cost, F(r - 1, c));
    if(r < rdim - 1)   min_cost = MINF(min_cost, F(r + 1, c));
    if(c > 0)          min_cost = MINF(min_cost, F(r, c - 1));
    if(c < cdim - 1)   min_cost = MINF(min_cost, F(r, c + 1));
    if(r > 0 && c > 0 && F(r - 1, c) < INFINITY && F(r, c - 1) < INFINITY)
        min_cost = MINF(min_cost, F(r - 1, c - 1));
    if(r > 0 && c < cdim - 1 && F(r - 1, c) < INFINITY && F(r, c + 1) < INFINITY)
        min_cost = MINF(min_cost, F(r - 1, c + 1));
    if(r < rdim - 1 && c > 0 && F(r + 1, c) < INFINITY && F(r, c - 1) < INFINITY)
        min_cost = MINF(min_cost, F(r + 1, c - 1));
    if(r < rdim - 1 && c < cdim - 1 && F(r + 1, c) < INFINITY && F(r, c + 1) < INFINITY)
        min_cost = MINF(min_cost, F(r + 1, c + 1));
    if(r > 0 && F(r - 1, c) == min_cost)                          return NAVHIP_FD_N;
    else if(r < rdim - 1 && F(r + 1, c) == min_cost)              return NAVHIP_FD_S;
    else if(c < cdim - 1 && F(r, c + 1) == min_cost)              return NAVHIP_FD_E;
    else if(c > 0 && F(r, c - 1) == min_cost)                     return NAVHIP_FD_W;
    else if(r > 0 && c > 0 && F(r - 1, c - 1) == min_cost)        return NAVHIP_FD_NW;
    else if(r > 0 && c < cdim - 1 && F(r - 1, c + 1) == min_cost) return NAVHIP_FD_NE;
    else if(r < rdim - 1 && c > 0 && F(r + 1, c - 1) == min_cost) return NAVHIP_FD_SW;
    else if(r < rdim - 1 && c < rdim - 1 && F(r + 1, c + 1) == min_cost) return NAVHIP_FD_SE;
    return NAVHIP_FD_NONE;
#undef F
#undef MINF
}

int no_region_field(const no_map *m, const navhip_region_req *rq, const int16_t *seeds,
                    const int16_t *overlay, uint8_t *inout)
{
    const int rdim = rq->rdim, cdim = rq->cdim, layer = rq->layer;
    if(rdim != cdim || rdim > 128 || (rdim & 1) || !m->cost[layer]) return -1;
    const int H = m->h * RES, W = m->w * RES;
    static __thread float integ[128 * 128];
    static __thread uint8_t mask[128 * 128];
    for(int i = 0; i < rdim * cdim; i++) integ[i] = INFINITY;
    memset(mask, 0, (size_t)rdim * cdim);
    pq_t q = {0};
    for(uint32_t k = 0; k < rq->overlay_count; k++) {            /* build_overlay_mask :571 */
        int dr = overlay[2 * (rq->overlay_begin + k)] - rq->base_abs_r;
        int dc = overlay[2 * (rq->overlay_begin + k) + 1] - rq->base_abs_c;
        if(dr >= 0 && dr < rdim && dc >= 0 && dc < cdim) mask[dr * rdim + dc] = 1;
    }
    for(uint32_t k = 0; k < rq->seed_count; k++) {
        int dr = seeds[2 * (rq->seed_begin + k)] - rq->base_abs_r;
        int dc = seeds[2 * (rq->seed_begin + k) + 1] - rq->base_abs_c;
        if(dr < 0 || dr >= rdim || dc < 0 || dc >= cdim) continue;
        pq_push(&q, 0.0f, dr * rdim + dc);
        integ[dr * rdim + dc] = 0.0f;
    }
    static const int dr4[4] = {-1, 0, 0, 1}, dc4[4] = {0, -1, 1, 0};
    while(q.n > 0) {                                             /* field_build_integration_region */
        int cur = pq_pop(&q);
        int r = cur / rdim, c = cur % rdim;
        for(int k = 0; k < 4; k++) {
            int nr = r + dr4[k], nc = c + dc4[k];
            int ar = rq->base_abs_r + nr, ac = rq->base_abs_c + nc;
            if(ar < 0 || ar >= H || ac < 0 || ac >= W) continue;              /* M_Tile_RelativeDesc */
            int chunk = (ar / RES) * m->w + (ac / RES);
            if(!tile_passable(m, layer, chunk, ar % RES, ac % RES,
                              rq->enemies ? 0 : FACTION_NONE, rq->enemies)) continue;
            if(nr < 0 || nr >= rdim || nc < 0 || nc >= cdim) continue;        /* tile_outside_region */
            if(mask[nr * rdim + nc]) continue;
            float total = integ[cur] + (float)m->cost[layer][((size_t)chunk << 12) + (ar % RES) * RES + (ac % RES)];
            if(total < integ[nr * rdim + nc]) {
                integ[nr * rdim + nc] = total;
                pq_push(&q, total, nr * rdim + nc);
            }
        }
    }
    free(q.a);
    if(rq->out_mode == 0) {                                      /* field_build_flow_unaligned :800 */
        memset(inout, 0, (size_t)rdim * cdim / 2);
        for(int r = 0; r < rdim; r++) {
        for(int c = 0; c < cdim; c++) {
            float v = integ[r * rdim + c];
            if(v == INFINITY) continue;
            int dir = (v == 0.0f) ? NAVHIP_FD_NONE : flow_dir_dim(integ, rdim, cdim, r, c);
            size_t bi = (size_t)r * (rdim / 2) + (c - (c % 2)) / 2;          /* set_flow_cell :786 */
            if(c % 2 == 1) inout[bi] = (uint8_t)((inout[bi] & 0xf0) | dir);
            else           inout[bi] = (uint8_t)((inout[bi] & 0x0f) | (dir << 4));
        }}
    }else{                                                       /* field_build_flow_region :763 */
        int lim_r = rdim < RES ? rdim : RES, lim_c = cdim < RES ? cdim : RES;
        for(int r = 0; r < lim_r; r++) {
        for(int c = 0; c < lim_c; c++) {
            int ir = r + rq->roff, ic = c + rq->coff;
            float v = integ[ir * rdim + ic];
            if(v == INFINITY) continue;
            inout[r * RES + c] = (v == 0.0f) ? NAVHIP_FD_NONE : (uint8_t)flow_dir_dim(integ, rdim, cdim, ir, ic);
        }}
    }
    return 0;
}

int no_build_region_fields(const no_map *m, const navhip_region_req *reqs, int n, const int16_t *seeds,
                           const int16_t *overlay, uint8_t *inout, size_t stride)
{
    for(int i = 0; i < n; i++) {
        int rc = no_region_field(m, &reqs[i], seeds, overlay, inout + (size_t)i * stride);
        if(rc) return rc;
    }
    return 0;
}

/* ===========================================================================================
 * line-of-sight fields: N_LOSFieldCreate (field.c:2085)
 * =========================================================================================== */

/* the reference's binary heap, lib/public/pqueue.h: 1-based nodes, strict comparisons, push sifts
 * the new node up past parents with a LARGER priority only (:150-171), pop moves the last node to
 * the root and sifts the hole down preferring the left child on ties (:112-133,173-183).  The LOS
 * wavefront depends on this exact pop order among equal priorities. */
typedef struct { float prio; int cell; } rpq_node;
typedef struct { rpq_node nodes[CELLS * 2 + 2]; int size; } rpq_t;

static void rpq_push(rpq_t *q, float prio, int cell)
{
    int curr = q->size + 1, parent = curr / 2;
    while(curr > 1 && q->nodes[parent].prio > prio) {
        q->nodes[curr] = q->nodes[parent];
        curr = parent;
        parent = parent / 2;
    }
    q->nodes[curr].prio = prio;
    q->nodes[curr].cell = cell;
    q->size++;
}

static int rpq_pop(rpq_t *q)
{
    int out = q->nodes[1].cell;
    q->nodes[1] = q->nodes[q->size--];
    int root = 1;
    while(root != q->size + 1) {
        int target = q->size + 1, l = root * 2, r = l + 1;
        if(l <= q->size && q->nodes[l].prio < q->nodes[target].prio) target = l;
        if(r <= q->size && q->nodes[r].prio < q->nodes[target].prio) target = r;
        q->nodes[root] = q->nodes[target];
        root = target;
    }
    return out;
}

static bool rpq_contains(const rpq_t *q, int cell)
{
    for(int i = 1; i <= q->size; i++)
        if(q->nodes[i].cell == cell) return true;
    return false;
}

/* field_create_wavefront_blocked_line, field.c:463 */
static void los_blocked_line(const navhip_los_req *rq, float map_x, float map_z, int corner_r,
                             int corner_c, uint8_t *los)
{
    float tbx = map_x - rq->target_chunk_c * 256 - rq->target_tile_c * 4;
    float tbz = map_z + rq->target_chunk_r * 256 + rq->target_tile_r * 4;
    float cbx = map_x - rq->chunk_c * 256 - corner_c * 4;
    float cbz = map_z + rq->chunk_r * 256 + corner_r * 4;
    float bw = 4, bh = 4;
    v2 target_center = mkv(tbx - bw / 2.0f, tbz + bh / 2.0f);
    v2 corner_center = mkv(cbx - bw / 2.0f, cbz + bh / 2.0f);
    v2 slope = vnormal(vsub(target_center, corner_center));
    int dx = abs((int)(slope.x * 1000));
    int dy = -abs((int)(slope.z * 1000));
    int sx = slope.x > 0.0f ? 1 : -1;
    int sy = slope.z < 0.0f ? 1 : -1;
    int err = dx + dy, e2;
    int r = corner_r, c = corner_c;
    do {
        los[r * RES + c] |= 2;
        e2 = 2 * err;
        if(e2 >= dy) { err += dy; c += sx; }
        if(e2 <= dx) { err += dx; r += sy; }
    } while(r >= 0 && r < RES && c >= 0 && c < RES);
}

/* N_LOSFieldCreate, field.c:2085.  prev / out: 4096 bytes, bit 0 visible, bit 1 wavefront_blocked. */
int no_los_field(const no_map *m, const navhip_los_req *rq, const uint8_t *prev, uint8_t *out,
                 float map_x, float map_z)
{
    if(rq->layer >= NLAYERS || !m->cost[rq->layer]) return -1;
    const int layer = rq->layer, chunk = rq->chunk_r * m->w + rq->chunk_c;
    const uint8_t *cost = m->cost[layer] + ((size_t)chunk << 12);
    const uint16_t *bl = m->blockers[layer] ? m->blockers[layer] + ((size_t)chunk << 12) : NULL;
    static __thread float integ[CELLS];
    static __thread rpq_t q;
    q.size = 0;
    memset(out, 0, CELLS);
    for(int i = 0; i < CELLS; i++) integ[i] = INFINITY;

    if(rq->prev_dr == 0 && rq->prev_dc == 0) {                /* case 1: the destination chunk */
        int t = rq->target_tile_r * RES + rq->target_tile_c;
        rpq_push(&q, 0.0f, t);
        integ[t] = 0.0f;
    }else{                                                    /* case 2: carry the shared edge */
        if(!prev) return -1;
        bool horizontal;
        int curr_edge, prev_edge;
        if(rq->prev_dr < 0)      { horizontal = false; curr_edge = 0;       prev_edge = RES - 1; }
        else if(rq->prev_dr > 0) { horizontal = false; curr_edge = RES - 1; prev_edge = 0; }
        else if(rq->prev_dc < 0) { horizontal = true;  curr_edge = 0;       prev_edge = RES - 1; }
        else                     { horizontal = true;  curr_edge = RES - 1; prev_edge = 0; }
        for(int k = 0; k < RES; k++) {
            int ci = horizontal ? k * RES + curr_edge : curr_edge * RES + k;
            int pi = horizontal ? k * RES + prev_edge : prev_edge * RES + k;
            out[ci] = prev[pi] & 3;
            if(out[ci] & 2)
                los_blocked_line(rq, map_x, map_z, ci >> 6, ci & 63, out);
            if(out[ci] & 1) {
                rpq_push(&q, 0.0f, ci);
                integ[ci] = 0.0f;
            }
        }
    }

    while(q.size > 0) {
        int cur = rpq_pop(&q);
        int r = cur >> 6, c = cur & 63;
        /* field_neighbours_grid_los, field.c:304 */
        int nb[4], ncost[4], nn = 0;
        for(int dr = -1; dr <= 1; dr++) {
        for(int dc = -1; dc <= 1; dc++) {
            int ar = r + dr, ac = c + dc;
            if(ar < 0 || ar >= RES || ac < 0 || ac >= RES) continue;
            if(dr == 0 && dc == 0) continue;
            if(dr == dc || dr == -dc) continue;
            if(out[ar * RES + ac] & 2) continue;
            nb[nn] = ar * RES + ac;
            ncost[nn] = cost[ar * RES + ac];
            if(!tile_passable(m, layer, chunk, ar, ac, rq->faction_id, rq->enemies))
                ncost[nn] = COST_IMPASS;
            nn++;
        }}
        for(int i = 0; i < nn; i++) {
            int ni = nb[i], nr = ni >> 6, nc = ni & 63;
            if(ncost[i] > 1) {
                /* field_is_los_corner, field.c:435 */
                bool corner = false;
#define RAWBLK(rr, cc) (cost[(rr) * RES + (cc)] == COST_IMPASS || (bl && bl[(rr) * RES + (cc)] > 0))
                if(nr > 0 && nr < RES - 1 && (RAWBLK(nr - 1, nc) ^ RAWBLK(nr + 1, nc))) corner = true;
                if(!corner && nc > 0 && nc < RES - 1 && (RAWBLK(nr, nc - 1) ^ RAWBLK(nr, nc + 1))) corner = true;
#undef RAWBLK
                if(!corner) continue;
                los_blocked_line(rq, map_x, map_z, nr, nc, out);
            }else{
                float new_cost = integ[cur] + 1;
                out[ni] |= 1;
                if(new_cost < integ[ni]) {
                    integ[ni] = new_cost;
                    if(!rpq_contains(&q, ni))
                        rpq_push(&q, new_cost, ni);
                }
            }
        }
    }
    /* field_pad_wavefront, field.c:519 */
    for(int r = 0; r < RES; r++) {
    for(int c = 0; c < RES; c++) {
        if(!(out[r * RES + c] & 2)) continue;
        for(int rr = r - 1; rr <= r + 1; rr++) {
        for(int cc = c - 1; cc <= c + 1; cc++) {
            if(rr < 0 || rr > RES - 1 || cc < 0 || cc > RES - 1) continue;
            out[rr * RES + cc] &= (uint8_t)~1;
        }}
    }}
    return 0;
}

int no_build_los(const no_map *m, const navhip_los_req *reqs, int n, const uint8_t *prev,
                 uint8_t *out, float map_x, float map_z)
{
    for(int i = 0; i < n; i++) {
        int rc = no_los_field(m, &reqs[i], prev ? prev + (size_t)i * CELLS : NULL,
                              out + (size_t)i * CELLS, map_x, map_z);
        if(rc) return rc;
    }
    return 0;
}

/* ===========================================================================================
 * spatial index (lib/public/bitmap_grid.h), as the harness builds it: bg_ent_init over the
 * grid bounds, insert uids 0..n-1, bg_ent_cleanup
 * =========================================================================================== */
typedef struct no_grid {
    int32_t origin_x, origin_y;
    int     grid_w, grid_h, n;
    int32_t *cell_start;          /* [ncells + 1] */
    int32_t *ids, *xs, *ys;       /* the clean pool: cells row-major (bitmap_grid.h:1506-1535) */
} no_grid;

static int32_t bg_scale(float v) { return (int32_t)lrintf(v * 256.0f); }    /* BG_SCALE_F :196 */

static int grid_cell(const no_grid *g, int32_t ix, int32_t iy)
{
    int cx = (ix - g->origin_x) >> 12, cy = (iy - g->origin_y) >> 12;    /* BG_CELL_LOG2_INT */
    if(cx < 0) cx = 0;
    if(cy < 0) cy = 0;
    if(cx >= g->grid_w) cx = g->grid_w - 1;
    if(cy >= g->grid_h) cy = g->grid_h - 1;
    return cy * g->grid_w + cx;
}

static int grid_build(no_grid *g, const navhip_world *w)
{
    memset(g, 0, sizeof(*g));
    /* bg_<name>_init, bitmap_grid.h:959-990 */
    g->origin_x = bg_scale(w->grid_xmin);
    g->origin_y = bg_scale(w->grid_zmin);
    int32_t span_x = bg_scale(w->grid_xmax) - g->origin_x;
    int32_t span_y = bg_scale(w->grid_zmax) - g->origin_y;
    if(span_x <= 0 || span_y <= 0) return -1;
    g->grid_w = (int)(((uint32_t)span_x + 4095u) >> 12);
    g->grid_h = (int)(((uint32_t)span_y + 4095u) >> 12);
    if(g->grid_w < 1) g->grid_w = 1;
    if(g->grid_h < 1) g->grid_h = 1;
    g->n = w->n_ents;
    int ncells = g->grid_w * g->grid_h, n = g->n;
    g->cell_start = calloc((size_t)ncells + 1, sizeof(int32_t));
    g->ids = malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    g->xs = malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    g->ys = malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int32_t *cell = malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for(int i = 0; i < n; i++) {
        cell[i] = grid_cell(g, bg_scale(w->pos_xz[2 * i]), bg_scale(w->pos_xz[2 * i + 1]));
        g->cell_start[cell[i] + 1]++;
    }
    for(int c = 0; c < ncells; c++) g->cell_start[c + 1] += g->cell_start[c];
    /* bg_insert pushes at the HEAD of the cell's overflow chain (:1102-1121) and cleanup copies
     * the chain head first (:1515-1521): inserting 0..n-1 leaves each cell in DESCENDING uid
     * order.  Fill every cell from its end while walking uids upwards. */
    int32_t *fill = malloc(sizeof(int32_t) * (size_t)ncells);
    for(int c = 0; c < ncells; c++) fill[c] = g->cell_start[c + 1];
    for(int i = 0; i < n; i++) {
        int slot = --fill[cell[i]];
        g->ids[slot] = i;
        g->xs[slot] = bg_scale(w->pos_xz[2 * i]);
        g->ys[slot] = bg_scale(w->pos_xz[2 * i + 1]);
    }
    free(fill);
    free(cell);
    return 0;
}

static void grid_free(no_grid *g)
{
    free(g->cell_start); free(g->ids); free(g->xs); free(g->ys);
    memset(g, 0, sizeof(*g));
}

/* bg_<name>_inrange_circle, bitmap_grid.h:1376: inclusive int64 distance test on the x256
 * fixed-point coordinates; coarse 8x8 blocks row-major, fine rows, cells left to right */
static int grid_query(const no_grid *g, float x, float z, float range, uint32_t *out, int maxout)
{
    if(maxout <= 0 || range < 0.0f) return 0;
    int32_t icx = bg_scale(x), icy = bg_scale(z), ir = bg_scale(range);
    int64_t ir2 = (int64_t)ir * (int64_t)ir;
    int32_t imnx = icx - ir, imxx = icx + ir, imny = icy - ir, imxy = icy + ir;
    /* _bg_cell_extent :1236 */
    if(imxx < g->origin_x || imxy < g->origin_y) return 0;
    int32_t span_x = (int32_t)((uint32_t)g->grid_w << 12), span_y = (int32_t)((uint32_t)g->grid_h << 12);
    if(imnx >= g->origin_x + span_x || imny >= g->origin_y + span_y) return 0;
    int cx_lo = (imnx - g->origin_x) >> 12, cx_hi = (imxx - g->origin_x) >> 12;
    int cy_lo = (imny - g->origin_y) >> 12, cy_hi = (imxy - g->origin_y) >> 12;
    if(cx_lo < 0) cx_lo = 0;
    if(cy_lo < 0) cy_lo = 0;
    if(cx_hi >= g->grid_w) cx_hi = g->grid_w - 1;
    if(cy_hi >= g->grid_h) cy_hi = g->grid_h - 1;

    int written = 0;
    int64_t extent = (int64_t)(cx_hi - cx_lo + 1) * (int64_t)(cy_hi - cy_lo + 1);
    int64_t total = (int64_t)g->grid_w * (int64_t)g->grid_h;
    if(extent * 4 >= total * 3) {                 /* wide-query path on a clean pool :1389-1397 */
        for(int k = 0; k < g->n; k++) {
            int64_t dx = (int64_t)g->xs[k] - icx, dy = (int64_t)g->ys[k] - icy;
            if(dx * dx + dy * dy <= ir2) {
                out[written++] = (uint32_t)g->ids[k];
                if(written >= maxout) return written;
            }
        }
        return written;
    }
    for(int cyc = cy_lo >> 3; cyc <= (cy_hi >> 3); cyc++) {
    for(int cxc = cx_lo >> 3; cxc <= (cx_hi >> 3); cxc++) {
        int fy0 = cyc * 8, fy1 = fy0 + 8, fx0 = cxc * 8, fx1 = fx0 + 8;
        if(fy0 < cy_lo) fy0 = cy_lo;
        if(fy1 > cy_hi + 1) fy1 = cy_hi + 1;
        if(fx0 < cx_lo) fx0 = cx_lo;
        if(fx1 > cx_hi + 1) fx1 = cx_hi + 1;
        for(int fy = fy0; fy < fy1; fy++) {
        for(int fx = fx0; fx < fx1; fx++) {
            int ci = fy * g->grid_w + fx;
            for(int k = g->cell_start[ci]; k < g->cell_start[ci + 1]; k++) {
                int64_t dx = (int64_t)g->xs[k] - icx, dy = (int64_t)g->ys[k] - icy;
                if(dx * dx + dy * dy <= ir2) {
                    out[written++] = (uint32_t)g->ids[k];
                    if(written >= maxout) return written;
                }
            }
        }}
    }}
    return written;
}

/* filter_garrisoned, position.c:100 */
static int filter_garrisoned(const uint32_t *flags, uint32_t *cand, int count)
{
    int ret = count;
    for(int i = count - 1; i >= 0; i--) {
        if(flags[cand[i]] & NAVHIP_ENTITY_FLAG_GARRISONED) {
            cand[i] = cand[ret - 1];
            ret--;
        }
    }
    return ret;
}

/* G_Pos_EntsInCircleFrom, position.c:379 */
static int ents_in_circle(const no_grid *g, const uint32_t *flags, v2 p, float range,
                          uint32_t *out, int maxout)
{
    int n = grid_query(g, p.x, p.z, range, out, maxout);
    return flags ? filter_garrisoned(flags, out, n) : n;
}

int no_spatial_query(const navhip_world *w, const float *query_xz, int nq, float range,
                     int maxout, int32_t *out_counts, uint32_t *out_ids)
{
    no_grid g;
    if(grid_build(&g, w)) return -1;
    for(int q = 0; q < nq; q++)
        out_counts[q] = grid_query(&g, query_xz[2 * q], query_xz[2 * q + 1], range,
                                   out_ids + (size_t)q * maxout, maxout);
    grid_free(&g);
    return 0;
}

/* ===========================================================================================
 * ClearPath (game/clearpath.c) + the two collision.c primitives under it
 * =========================================================================================== */
typedef struct { v2 pos, vel; float radius; } cpent;
typedef struct { v2 point, dir; } line2d;

/* C_InfiniteLineIntersection, collision.c:820 (the vertical-l2 branch really adds l2.point.z) */
static bool line_isect(line2d l1, line2d l2, v2 *out)
{
    float s1 = fabs(l1.dir.x) < EPS ? NAN : (l1.dir.z / l1.dir.x);
    float s2 = fabs(l2.dir.x) < EPS ? NAN : (l2.dir.z / l2.dir.x);
    if(isnan(s1) && isnan(s2)) return false;
    if(fabs(s1 - s2) < EPS) return false;
    if(isnan(s1) && !isnan(s2)) {
        out->x = l1.point.x;
        out->z = (l1.point.x - l2.point.x) * s2 + l2.point.z;
    }else if(!isnan(s1) && isnan(s2)) {
        out->x = l2.point.x;
        out->z = (l2.point.x - l1.point.x) * s1 + l2.point.z;
    }else{
        out->x = (s1 * l1.point.x - s2 * l2.point.x + l2.point.z - l1.point.z) / (s1 - s2);
        out->z = s2 * (out->x - l2.point.x) + l2.point.z;
    }
    return true;
}

/* C_RayRayIntersection2D, collision.c:854 */
static bool ray_isect(line2d l1, line2d l2, v2 *out)
{
    v2 p;
    if(!line_isect(l1, l2, &p)) return false;
    if((p.x - l1.point.x) / l1.dir.x < 0.0f) return false;
    if((p.z - l1.point.z) / l1.dir.z < 0.0f) return false;
    if((p.x - l2.point.x) / l2.dir.x < 0.0f) return false;
    if((p.z - l2.point.z) / l2.dir.z < 0.0f) return false;
    *out = p;
    return true;
}

/* compute_vo_edges, clearpath.c:130 */
static void vo_edges(cpent ent, cpent nb, v2 *out_right, v2 *out_left)
{
    v2 e2n = vnormal(vsub(nb.pos, ent.pos));
    v2 right = mkv(-e2n.z, e2n.x);
    right = vscale(right, nb.radius + ent.radius + 0.0f);      /* CLEARPATH_BUFFER_RADIUS 0 */
    v2 right_tangent = vadd(nb.pos, right), left_tangent = vsub(nb.pos, right);
    *out_right = vnormal(vsub(right_tangent, ent.pos));
    *out_left = vnormal(vsub(left_tangent, ent.pos));
}

/* inside_pcr, clearpath.c:249; rays[] = (left, right) pairs */
static bool inside_pcr(const line2d *rays, int n_rays, v2 test)
{
    for(int i = 0; i < n_rays; i += 2) {
        v2 ptt = vsub(test, rays[i].point);
        if(vlen(ptt) < EPS) continue;
        ptt = vnormal(ptt);
        float left_det = (ptt.z * rays[i].dir.x) - (ptt.x * rays[i].dir.z);
        if(left_det < EPS) continue;
        ptt = vsub(test, rays[i + 1].point);
        if(vlen(ptt) < EPS) continue;
        ptt = vnormal(ptt);
        float right_det = (ptt.z * rays[i + 1].dir.x) - (ptt.x * rays[i + 1].dir.z);
        if(right_det > -EPS) continue;
        return true;
    }
    return false;
}

/* clearpath_new_velocity, clearpath.c:552 */
static bool cp_new_velocity(cpent ent, v2 des_v, const cpent *dyn, int n_dyn, const cpent *stat,
                            int n_stat, v2 *out)
{
    line2d rays[2 * 64];
    int n_rays = 0;
    for(int i = 0; i < n_dyn; i++) {                           /* compute_all_hrvos :232 */
        if(vlen(vsub(dyn[i].pos, ent.pos)) < EPS) continue;   /* same_position :123 */
        /* compute_hrvo :180 over compute_rvo :166 */
        v2 right, left;
        vo_edges(ent, dyn[i], &right, &left);
        v2 rvo_apex = vadd(ent.pos, vscale(vadd(ent.vel, dyn[i].vel), 0.5f));
        v2 centerline = vadd(left, right);
        v2 vo_apex = vadd(ent.pos, dyn[i].vel);
        v2 apex;
        float det = (centerline.x * ent.vel.z) - (centerline.z * ent.vel.x);
        if(det > EPS) {
            line2d l1 = {rvo_apex, left}, l2 = {vo_apex, right};
            apex = rvo_apex;        /* the reference asserts the lines meet; keep a defined value */
            line_isect(l1, l2, &apex);
        }else if(det < -EPS) {
            line2d l1 = {rvo_apex, right}, l2 = {vo_apex, left};
            apex = rvo_apex;
            line_isect(l1, l2, &apex);
        }else{
            apex = rvo_apex;
        }
        rays[n_rays].point = apex;     rays[n_rays].dir = left;        /* rays_repr :291 */
        rays[n_rays + 1].point = apex; rays[n_rays + 1].dir = right;
        n_rays += 2;
    }
    for(int i = 0; i < n_stat; i++) {                          /* compute_all_vos :216 */
        if(vlen(vsub(stat[i].pos, ent.pos)) < EPS) continue;
        v2 right, left;
        vo_edges(ent, stat[i], &right, &left);                 /* compute_vo :153 */
        v2 apex = vadd(ent.pos, stat[i].vel);
        rays[n_rays].point = apex;     rays[n_rays].dir = left;
        rays[n_rays + 1].point = apex; rays[n_rays + 1].dir = right;
        n_rays += 2;
    }

    v2 des_ws = vadd(ent.pos, des_v);
    if(!inside_pcr(rays, n_rays, des_ws)) {
        *out = des_v;
        return true;
    }

    /* compute_vo_xpoints :321, compute_vdes_proj_points :344, compute_vnew :368 fused: the
     * candidates are visited in the order the reference pushes them, first strict minimum wins */
    float min_dist = INFINITY;
    v2 ret = mkv(0.0f, 0.0f);
    int npoints = 0;
    for(int i = 0; i < n_rays; i++) {
    for(int j = 0; j < n_rays; j++) {
        if(i == j) continue;
        v2 pt;
        if(!ray_isect(rays[i], rays[j], &pt)) continue;
        if(inside_pcr(rays, n_rays, pt)) continue;
        npoints++;
        v2 curr = vsub(pt, ent.pos);
        float len = vlen(vsub(des_v, curr));
        if(len < min_dist) { min_dist = len; ret = curr; }
    }}
    for(int i = 0; i < n_rays; i++) {
        float len = vdot(rays[i].dir, des_v);
        v2 proj = vadd(rays[i].point, vscale(rays[i].dir, len));
        if(inside_pcr(rays, n_rays, proj)) continue;
        npoints++;
        v2 curr = vsub(proj, ent.pos);
        float l2 = vlen(vsub(des_v, curr));
        if(l2 < min_dist) { min_dist = l2; ret = curr; }
    }
    if(npoints == 0) return false;
    *out = ret;
    return true;
}

/* G_ClearPath_NewVelocity, clearpath.c:694 (+ remove_furthest :390, vec del = swap with last) */
static v2 cp_solve(cpent ent, v2 des_v, cpent *dyn, int n_dyn, cpent *stat, int n_stat)
{
    do {
        v2 ret;
        if(cp_new_velocity(ent, des_v, dyn, n_dyn, stat, n_stat, &ret))
            return ret;
        float max_dist = -INFINITY;
        int del_list = -1, del_idx = -1;
        for(int l = 0; l < 2; l++) {
            const cpent *v = l == 0 ? dyn : stat;
            int n = l == 0 ? n_dyn : n_stat;
            for(int j = 0; j < n; j++) {
                float len = vlen(vsub(ent.pos, v[j].pos));
                if(len > max_dist) { max_dist = len; del_list = l; del_idx = j; }
            }
        }
        if(max_dist > -INFINITY) {
            if(del_list == 0) dyn[del_idx] = dyn[--n_dyn];
            else              stat[del_idx] = stat[--n_stat];
        }
    } while(n_dyn > 0 && n_stat > 0);
    return mkv(0.0f, 0.0f);
}

int no_clearpath(int nq, const float *ent, const float *des_v, const float *dyn,
                 const int32_t *n_dyn, const float *stat, const int32_t *n_stat, float *out)
{
    for(int q = 0; q < nq; q++) {
        cpent e = {mkv(ent[5 * q], ent[5 * q + 1]), mkv(ent[5 * q + 2], ent[5 * q + 3]), ent[5 * q + 4]};
        cpent d[32], s[32];
        if(n_dyn[q] < 0 || n_dyn[q] > 32 || n_stat[q] < 0 || n_stat[q] > 32) return -1;
        for(int k = 0; k < n_dyn[q]; k++) {
            const float *p = dyn + ((size_t)q * 32 + k) * 5;
            d[k].pos = mkv(p[0], p[1]); d[k].vel = mkv(p[2], p[3]); d[k].radius = p[4];
        }
        for(int k = 0; k < n_stat[q]; k++) {
            const float *p = stat + ((size_t)q * 32 + k) * 5;
            s[k].pos = mkv(p[0], p[1]); s[k].vel = mkv(p[2], p[3]); s[k].radius = p[4];
        }
        v2 r = cp_solve(e, mkv(des_v[2 * q], des_v[2 * q + 1]), d, n_dyn[q], s, n_stat[q]);
        out[2 * q] = r.x; out[2 * q + 1] = r.z;
    }
    return 0;
}

/* ===========================================================================================
 * tile lookups, flow sampling
 * =========================================================================================== */
typedef struct { int chunk_r, chunk_c, tile_r, tile_c; } tiledesc;

#define CLAMPI(v, lo, hi) ((v) < (lo) ? (lo) : (v) > (hi) ? (hi) : (v))

/* M_Tile_DescForPoint2D, tile.c:547, at the nav resolution n_res (nav.c:247): 64x64 tiles of
 * 4 wu per 256-wu chunk */
static bool tile_for_point(const no_map *m, float map_x, float map_z, v2 p, tiledesc *out)
{
    float width = (float)(size_t)(m->w * 256), height = (float)(size_t)(m->h * 256);
    if(p.x > map_x || p.x < map_x - width) return false;
    if(p.z < map_z || p.z > map_z + height) return false;
    int chunk_r = (int)(fabs(map_z - p.z) / 256);
    int chunk_c = (int)(fabs(map_x - p.x) / 256);
    chunk_r = CLAMPI(chunk_r, 0, m->h - 1);
    chunk_c = CLAMPI(chunk_c, 0, m->w - 1);
    float base_x = map_x - (chunk_c * 256);
    float base_z = map_z + (chunk_r * 256);
    int tile_r = (int)(fabs(base_z - p.z) / 4);
    int tile_c = (int)(fabs(base_x - p.x) / 4);
    out->chunk_r = chunk_r; out->chunk_c = chunk_c;
    out->tile_r = CLAMPI(tile_r, 0, 63);
    out->tile_c = CLAMPI(tile_c, 0, 63);
    return true;
}

/* Entity_NavLayerWithRadius, entity.c:554 */
static int nav_layer_for(uint32_t flags, float radius)
{
    int base = (flags & NAVHIP_ENTITY_FLAG_WATER) ? 4 : (flags & NAVHIP_ENTITY_FLAG_AIR) ? 8 : 0;
    if(radius >= 15.0f) return base + 3;
    if(radius >= 10.0f) return base + 2;
    if(radius >= 5.0f)  return base + 1;
    return base;
}

/* N_PositionPathable nav.c:4055 / N_PositionBlocked nav.c:4070 (off-map: the reference asserts;
 * reported as not pathable / not blocked) */
static bool pos_pathable(const no_map *m, const navhip_world *w, int layer, v2 p)
{
    tiledesc t;
    if(!tile_for_point(m, w->map_pos_x, w->map_pos_z, p, &t)) return false;
    return m->cost[layer][((size_t)(t.chunk_r * m->w + t.chunk_c) << 12) + t.tile_r * RES + t.tile_c]
           != COST_IMPASS;
}

static bool pos_blocked(const no_map *m, const navhip_world *w, int layer, v2 p)
{
    tiledesc t;
    if(!tile_for_point(m, w->map_pos_x, w->map_pos_z, p, &t)) return false;
    if(!m->blockers[layer]) return false;
    return m->blockers[layer][((size_t)(t.chunk_r * m->w + t.chunk_c) << 12) + t.tile_r * RES + t.tile_c] > 0;
}

/* N_FlowDir, field.c:2428 */
static v2 flow_dir_vec(int dir)
{
    const float d = (float)(1.0f / sqrt(2.0f));
    switch(dir) {
    case NAVHIP_FD_NW: return mkv( d, -d);
    case NAVHIP_FD_N:  return mkv( 0.0f, -1.0f);
    case NAVHIP_FD_NE: return mkv(-d, -d);
    case NAVHIP_FD_W:  return mkv( 1.0f, 0.0f);
    case NAVHIP_FD_E:  return mkv(-1.0f, 0.0f);
    case NAVHIP_FD_SW: return mkv( d,  d);
    case NAVHIP_FD_S:  return mkv( 0.0f, 1.0f);
    case NAVHIP_FD_SE: return mkv(-d,  d);
    default:           return mkv(0.0f, 0.0f);
    }
}

/* N_DesiredPointSeekVelocity (nav.c:3468) on a cache HIT + n_interpolated_flow_dir (nav.c:3407).
 * The (dest, chunk) -> field mapping of the field cache is the flock_field_slot table; a miss or
 * an FD_NONE base tile is reported through *status for the host planner (nav.c:3483-3554). */
static v2 sample_flow(const no_map *m, const navhip_world *w, int flock, v2 pos, unsigned *status)
{
    tiledesc t;
    if(flock < 0 || !w->flock_field_slot || !w->field_pool
    || !tile_for_point(m, w->map_pos_x, w->map_pos_z, pos, &t)) {
        *status |= NAVHIP_ST_FIELD_MISS;
        return mkv(0.0f, 0.0f);
    }
    const int nchunks = m->w * m->h;
    const int32_t *slots = w->flock_field_slot + (size_t)flock * nchunks;
    int slot = slots[t.chunk_r * m->w + t.chunk_c];
    if(slot < 0) {
        *status |= NAVHIP_ST_FIELD_MISS;
        return mkv(0.0f, 0.0f);
    }
    const uint8_t *base_ff = w->field_pool + ((size_t)slot << 12);
    int base_dir = base_ff[t.tile_r * RES + t.tile_c] & 0xf;
    if(base_dir == NAVHIP_FD_NONE) *status |= NAVHIP_ST_FIELD_NONE;

    /* M_Tile_Bounds, tile.c:356 */
    float bx = w->map_pos_x - t.chunk_c * 256 - t.tile_c * 4;
    float bz = w->map_pos_z + t.chunk_r * 256 + t.tile_r * 4;
    float bw = 4, bh = 4;
    v2 centre = mkv(bx - bw / 2.0f, bz + bh / 2.0f);
    float dx = pos.x - centre.x, dz = pos.z - centre.z;
    int dc = (dx < 0.0f) ? 1 : -1;
    int dr = (dz > 0.0f) ? 1 : -1;
    float wc = (float)fmin(fabs(dx) / bw, 1.0f);
    float wr = (float)fmin(fabs(dz) / bh, 1.0f);
    const int   sdc[4] = {0, dc, 0, dc};
    const int   sdr[4] = {0, 0, dr, dr};
    const float sw[4]  = {(1.0f - wc) * (1.0f - wr), wc * (1.0f - wr), (1.0f - wc) * wr, wc * wr};

    v2 acc = mkv(0.0f, 0.0f);
    float wsum = 0.0f;
    for(int i = 0; i < 4; i++) {
        if(sw[i] <= 0.0f) continue;
        /* M_Tile_RelativeDesc, tile.c:391 */
        int abs_r = t.chunk_r * RES + t.tile_r + sdr[i];
        int abs_c = t.chunk_c * RES + t.tile_c + sdc[i];
        if(abs_r < 0 || abs_r >= m->h * RES || abs_c < 0 || abs_c >= m->w * RES) continue;
        int cr = abs_r / RES, cc = abs_c / RES, tr = abs_r % RES, tc = abs_c % RES;
        const uint8_t *ff = base_ff;
        if(cr != t.chunk_r || cc != t.chunk_c) {
            int s2 = slots[cr * m->w + cc];
            if(s2 < 0) continue;
            ff = w->field_pool + ((size_t)s2 << 12);
        }
        int dir = ff[tr * RES + tc] & 0xf;
        if(dir == NAVHIP_FD_NONE) continue;
        v2 scaled = vscale(flow_dir_vec(dir), sw[i]);
        acc = vadd(acc, scaled);
        wsum += sw[i];
    }
    if(wsum < 1e-6f || vlen(acc) < 1e-6f)
        return flow_dir_vec(base_dir);
    return vnormal(acc);
}

/* ===========================================================================================
 * movement step (game/movement.c)
 * =========================================================================================== */
typedef struct step_ctx {
    const no_map *m;
    const navhip_world *w;
    const no_grid *g;
    float  scaled_max_force_f;       /* SCALED_MAX_FORCE (movement.c:93) converted to float */
    double scaled_max_force;
} step_ctx;

static v2 wpos(const navhip_world *w, int uid) { return mkv(w->pos_xz[2 * uid], w->pos_xz[2 * uid + 1]); }
static v2 wvel(const navhip_world *w, int uid) { return mkv(w->vel_xz[2 * uid], w->vel_xz[2 * uid + 1]); }

static bool state_still(int s) { return s == NAVHIP_STATE_ARRIVED || s == NAVHIP_STATE_WAITING; }   /* :652 */

/* arrive_force_point, movement.c:1546 */
static v2 arrive_force_point(const step_ctx *c, int uid, v2 target, v2 vdes, bool los)
{
    const navhip_world *w = c->w;
    v2 desired;
    if(los) {
        desired = vsub(target, wpos(w, uid));
        float distance = vlen(desired);
        desired = vnormal(desired);
        desired = vscale(desired, w->max_speed[uid] / w->hz);
        if(distance < 10.0f)                                        /* ARRIVE_SLOWING_RADIUS */
            desired = vscale(desired, distance / 10.0f);
    }else{
        desired = vscale(vdes, w->max_speed[uid] / w->hz);
    }
    return vtrunc(vsub(desired, wvel(w, uid)), c->scaled_max_force_f);
}

/* cohesion_force, movement.c:1653: exp-weighted centroid of the WHOLE flock, in member order */
static v2 cohesion_force(const step_ctx *c, int uid, int flock)
{
    const navhip_world *w = c->w;
    v2 com = mkv(0.0f, 0.0f);
    size_t count = 0;
    v2 me = wpos(w, uid);
    for(int j = w->flock_offsets[flock]; j < w->flock_offsets[flock + 1]; j++) {
        int curr = w->flock_members[j];
        if(curr == uid) continue;
        v2 cp = wpos(w, curr);
        v2 diff = vsub(cp, me);
        float t = (vlen(diff) - 50.0f * 0.75) / 50.0f;             /* COHESION_NEIGHBOUR_RADIUS */
        float scale = exp(-6.0f * t);
        cp = vscale(cp, scale);
        com = vadd(com, cp);
        count++;
    }
    if(count == 0) return mkv(0.0f, 0.0f);
    com = vscale(com, 1.0f / count);
    return vtrunc(vsub(com, me), c->scaled_max_force_f);
}

/* separation_force, movement.c:1690 */
static v2 separation_force(const step_ctx *c, int uid)
{
    const navhip_world *w = c->w;
    v2 ret = mkv(0.0f, 0.0f);
    uint32_t ent_flags = w->flags[uid];
    uint32_t near_ents[128];
    v2 me = wpos(w, uid);
    int num_near = ents_in_circle(c->g, w->flags, me, 30.0f, near_ents, 128);   /* SEPARATION_NEIGHB_RADIUS */
    for(int i = 0; i < num_near; i++) {
        uint32_t curr = near_ents[i];
        uint32_t flags = w->flags[curr];
        if((int)curr == uid) continue;
        if(!(flags & NAVHIP_ENTITY_FLAG_MOVABLE)) continue;
        if((ent_flags & NAVHIP_ENTITY_FLAG_AIR) != (flags & NAVHIP_ENTITY_FLAG_AIR)) continue;
        float radius = w->radius[uid] + w->radius[curr] + 0.0f;   /* SEPARATION_BUFFER_DIST */
        v2 diff = vsub(wpos(w, (int)curr), me);
        if(vlen(diff) < EPS) continue;
        float eq = 0.85f, steep = 20.0f;
        float t = (vlen(diff) - radius * eq) / vlen(diff);
        float a = -steep * t;
        float scale = exp(a < 40.0f ? a : 40.0f);
        diff = vscale(diff, scale);
        ret = vadd(ret, diff);
    }
    if(num_near == 0) return mkv(0.0f, 0.0f);
    ret = vscale(ret, -1.0f);
    return vtrunc(ret, c->scaled_max_force_f);
}

/* nullify_impass_components, movement.c:1831 (N_TileDims = 4 x 4 wu, nav.c:4653) */
static v2 nullify_impass(const step_ctx *c, int uid, v2 f)
{
    const navhip_world *w = c->w;
    int layer = nav_layer_for(w->flags[uid], w->radius[uid]);
    v2 pos = wpos(w, uid);
    v2 left = mkv(pos.x + 4.0f, pos.z), right = mkv(pos.x - 4.0f, pos.z);
    v2 top = mkv(pos.x, pos.z + 4.0f), bot = mkv(pos.x, pos.z - 4.0f);
    bool on_blocked = pos_blocked(c->m, w, layer, pos);
    if(f.x > 0 && (!pos_pathable(c->m, w, layer, left) || (!on_blocked && pos_blocked(c->m, w, layer, left))))
        f.x = 0.0f;
    if(f.x < 0 && (!pos_pathable(c->m, w, layer, right) || (!on_blocked && pos_blocked(c->m, w, layer, right))))
        f.x = 0.0f;
    if(f.z > 0 && (!pos_pathable(c->m, w, layer, top) || (!on_blocked && pos_blocked(c->m, w, layer, top))))
        f.z = 0.0f;
    if(f.z < 0 && (!pos_pathable(c->m, w, layer, bot) || (!on_blocked && pos_blocked(c->m, w, layer, bot))))
        f.z = 0.0f;
    return f;
}

/* point_seek_total_force :1745 + point_seek_vpref :1870 (arrival module inactive: the seek
 * target is flock->target_xz) */
static v2 point_seek_vpref(const step_ctx *c, int uid, int flock, v2 vdes, bool los, float speed)
{
    const navhip_world *w = c->w;
    v2 target = flock >= 0 ? mkv(w->flock_target_xz[2 * flock], w->flock_target_xz[2 * flock + 1])
                           : wpos(w, uid);
    v2 steer = mkv(0.0f, 0.0f);
    for(int prio = 0; prio < 3; prio++) {
        switch(prio) {
        case 0: {
            v2 arrive = arrive_force_point(c, uid, target, vdes, los);
            v2 cohesion = flock >= 0 ? cohesion_force(c, uid, flock) : mkv(0.0f, 0.0f);
            v2 separation = separation_force(c, uid);
            arrive = vscale(arrive, 0.5f);                 /* MOVE_ARRIVE_FORCE_SCALE */
            cohesion = vscale(cohesion, 0.15f);            /* MOVE_COHESION_FORCE_SCALE */
            separation = vscale(separation, 0.6f);         /* SEPARATION_FORCE_SCALE */
            v2 ret = mkv(0.0f, 0.0f);
            ret = vadd(ret, arrive);
            ret = vadd(ret, separation);
            ret = vadd(ret, cohesion);
            steer = vtrunc(ret, c->scaled_max_force_f);
            break;
        }
        case 1: steer = separation_force(c, uid); break;
        case 2: steer = arrive_force_point(c, uid, target, vdes, los); break;
        }
        steer = nullify_impass(c, uid, steer);
        if(vlen(steer) > c->scaled_max_force * 0.01)
            break;
    }
    v2 accel = vscale(steer, 1.0f / 1.0f);                  /* ENTITY_MASS */
    v2 new_vel = vadd(wvel(w, uid), accel);
    return vtrunc(new_vel, speed / w->hz);
}

/* enemy_seek_vpref :1946 over enemy_seek_total_force :1815 / arrive_force_enemies :1593 */
static v2 enemy_seek_vpref(const step_ctx *c, int uid, float speed, v2 vdes)
{
    const navhip_world *w = c->w;
    v2 desired = vscale(vdes, w->max_speed[uid] / w->hz);
    v2 arrive = vtrunc(vsub(desired, wvel(w, uid)), c->scaled_max_force_f);
    v2 separation = separation_force(c, uid);
    arrive = vscale(arrive, 0.5f);
    separation = vscale(separation, 0.6f);
    v2 ret = mkv(0.0f, 0.0f);
    ret = vadd(ret, arrive);
    ret = vadd(ret, separation);
    v2 steer = vtrunc(ret, c->scaled_max_force_f);
    v2 accel = vscale(steer, 1.0f / 1.0f);
    return vtrunc(vadd(wvel(w, uid), accel), speed / w->hz);
}

/* find_neighbours, movement.c:2768 (arrival slots inactive) */
static void find_neighbours(const step_ctx *c, int uid, cpent *dyn, int *n_dyn, cpent *stat, int *n_stat)
{
    const navhip_world *w = c->w;
    uint32_t ent_flags = w->flags[uid];
    uint32_t near_ents[512];
    int num_near = ents_in_circle(c->g, w->flags, wpos(w, uid), 10.0f, near_ents, 512);
    *n_dyn = *n_stat = 0;
    for(int i = 0; i < num_near; i++) {
        uint32_t curr = near_ents[i];
        uint32_t flags = w->flags[curr];
        if((int)curr == uid) continue;
        if(!(flags & NAVHIP_ENTITY_FLAG_MOVABLE)) continue;
        if(w->radius[curr] == 0.0f) continue;
        if((ent_flags & NAVHIP_ENTITY_FLAG_AIR) != (flags & NAVHIP_ENTITY_FLAG_AIR)) continue;
        cpent nd = {wpos(w, (int)curr), wvel(w, (int)curr), w->radius[curr]};
        if(state_still(w->state[curr]) || vlen(nd.vel) < 0.3f) {        /* CLEARPATH_STILL_SPEED */
            nd.vel = mkv(0.0f, 0.0f);
            if(*n_stat < 32) stat[(*n_stat)++] = nd;                    /* MAX_NEIGHBOURS */
        }else{
            if(*n_dyn < 32) dyn[(*n_dyn)++] = nd;
        }
    }
}

/* arrive_force_cell, movement.c:1574 */
static v2 arrive_force_cell(const step_ctx *c, int uid, v2 cell_xz, v2 vdes)
{
    const navhip_world *w = c->w;
    v2 desired = vsub(cell_xz, wpos(w, uid));
    float distance = vlen(desired);
    if(distance < 10.0f)
        desired = vscale(desired, distance / 10.0f);
    else
        desired = vscale(vdes, w->max_speed[uid] / w->hz);
    return desired;
}

/* cell_arrival_seek_vpref :1908 over cell_seek_total_force :1773, and formation_seek_vpref :1985
 * over formation_point_seek_total_force :1962; the formation forces are host inputs */
static v2 formation_vpref(const step_ctx *c, int uid, int flock, v2 vdes, bool to_cell)
{
    const navhip_world *w = c->w;
    v2 cell = mkv(w->cell_pos_xz[2 * uid], w->cell_pos_xz[2 * uid + 1]);
    v2 cohesion = mkv(w->form_cohesion_xz[2 * uid], w->form_cohesion_xz[2 * uid + 1]);
    v2 alignment = mkv(w->form_align_xz[2 * uid], w->form_align_xz[2 * uid + 1]);
    v2 drag = mkv(w->form_drag_xz[2 * uid], w->form_drag_xz[2 * uid + 1]);
    v2 target = flock >= 0 ? mkv(w->flock_target_xz[2 * flock], w->flock_target_xz[2 * flock + 1]) : wpos(w, uid);
    bool los = w->has_dest_los[uid] != 0;
    float speed = w->speed[uid];
    v2 steer = mkv(0.0f, 0.0f);
    for(int prio = 0; prio < 3; prio++) {
        switch(prio) {
        case 0: {
            v2 arrive = to_cell ? arrive_force_cell(c, uid, cell, vdes)
                                : arrive_force_point(c, uid, target, vdes, los);
            v2 separation = separation_force(c, uid);
            v2 co = vscale(cohesion, 0.15f), al = vscale(alignment, 0.15f);
            arrive = vscale(arrive, 0.5f);
            separation = vscale(separation, 0.6f);
            v2 ret = mkv(0.0f, 0.0f);
            ret = vadd(ret, arrive);
            ret = vadd(ret, separation);
            if(to_cell) {
                v2 delta = vsub(cell, wpos(w, uid));
                if(vlen(delta) > 30.0f) {                        /* CELL_ARRIVAL_RADIUS */
                    ret = vadd(ret, co);
                    ret = vadd(ret, al);
                }
            }else{
                ret = vadd(ret, co);
            }
            steer = vtrunc(ret, c->scaled_max_force_f);
            break;
        }
        case 1: steer = separation_force(c, uid); break;
        case 2: steer = to_cell ? arrive_force_cell(c, uid, cell, vdes)
                                : arrive_force_point(c, uid, target, vdes, los);
                break;
        }
        steer = nullify_impass(c, uid, steer);
        if(vlen(steer) > c->scaled_max_force * 0.01)
            break;
    }
    v2 accel = vscale(steer, 1.0f / 1.0f);
    v2 new_vel = vtrunc(vadd(wvel(w, uid), accel), speed / w->hz);
    if(vlen(drag) > (1.0f / 1024))
        new_vel = vtrunc(new_vel, (speed * 0.75) / w->hz);
    return new_vel;
}

static v2 load_vdes(const step_ctx *c, int uid, v2 me, unsigned *status)
{
    const navhip_world *w = c->w;
    if(w->vdes_xz && !isnan(w->vdes_xz[2 * uid]))
        return mkv(w->vdes_xz[2 * uid], w->vdes_xz[2 * uid + 1]);
    return sample_flow(c->m, w, w->flock[uid], me, status);
}

static bool state_point_seek(int s)
{
    return s == NAVHIP_STATE_MOVING || s == NAVHIP_STATE_SURROUND_ENTITY
        || s == NAVHIP_STATE_ENTER_ENTITY_RANGE;
}

/* move_velocity_work (movement.c:3395) for one entity + the position accept test of
 * entity_compute_update (movement.c:2336-2358), with the output conventions of
 * navhip_agent_step (include/navhip.h) */
static void step_one(const step_ctx *c, int uid, const navhip_step_out *o)
{
    const navhip_world *w = c->w;
    const int state = w->state[uid];
    const uint32_t flags = w->flags[uid];
    unsigned status = 0;
    v2 out_vel = mkv(0.0f, 0.0f), vdes = mkv(0.0f, 0.0f), vpref = mkv(0.0f, 0.0f);
    const bool active = !state_still(state);
    const v2 me = wpos(w, uid);

    if(active && !(flags & NAVHIP_ENTITY_FLAG_COMBAT_HELD)) {
        bool supported = true;
        if(state == NAVHIP_STATE_TURNING) {
            vpref = mkv(0.0f, 0.0f);
        }else if(state == NAVHIP_STATE_SEEK_ENEMIES || state_point_seek(state)) {
            vdes = load_vdes(c, uid, me, &status);
            if(state == NAVHIP_STATE_SEEK_ENEMIES)
                vpref = enemy_seek_vpref(c, uid, w->speed[uid], vdes);
            else
                vpref = point_seek_vpref(c, uid, w->flock[uid], vdes, w->has_dest_los[uid] != 0,
                                         w->speed[uid]);
        }else if(w->form_ready && (state == NAVHIP_STATE_MOVING_IN_FORMATION
                                  || state == NAVHIP_STATE_ARRIVING_TO_CELL)) {
            if(!w->form_ready[uid]) {                       /* movement.c:3425,3437 */
                vpref = mkv(0.0f, 0.0f);
            }else{
                vdes = load_vdes(c, uid, me, &status);
                vpref = formation_vpref(c, uid, w->flock[uid], vdes, state == NAVHIP_STATE_ARRIVING_TO_CELL);
            }
        }else{
            supported = false;
            status |= NAVHIP_ST_UNSUPPORTED;
        }
        if(supported) {
            cpent dyn[32], stat[32];
            int n_dyn, n_stat;
            find_neighbours(c, uid, dyn, &n_dyn, stat, &n_stat);
            cpent ent = {me, wvel(w, uid), w->radius[uid]};
            v2 nv = cp_solve(ent, vpref, dyn, n_dyn, stat, n_stat);
            out_vel = vtrunc(nv, w->max_speed[uid] / w->hz);
        }
    }

    v2 new_pos = me;
    /* entity_compute_update returns before the position update for a garrisoned entity
     * (movement.c:2341-2348) */
    if(active && !(flags & NAVHIP_ENTITY_FLAG_GARRISONED)) {
        int layer = nav_layer_for(flags, w->radius[uid]);
        v2 cand = vadd(me, out_vel);
        bool on_blocked = pos_blocked(c->m, w, layer, me);
        if(vlen(out_vel) > 0 && pos_pathable(c->m, w, layer, cand)
        && (on_blocked || !pos_blocked(c->m, w, layer, cand))) {
            new_pos = cand;
            status |= NAVHIP_ST_MOVED;
        }
    }
    o->vel_xz[2 * uid] = out_vel.x; o->vel_xz[2 * uid + 1] = out_vel.z;
    if(o->new_pos_xz) { o->new_pos_xz[2 * uid] = new_pos.x; o->new_pos_xz[2 * uid + 1] = new_pos.z; }
    if(o->vdes_xz)    { o->vdes_xz[2 * uid] = vdes.x;    o->vdes_xz[2 * uid + 1] = vdes.z; }
    if(o->vpref_xz)   { o->vpref_xz[2 * uid] = vpref.x;  o->vpref_xz[2 * uid + 1] = vpref.z; }
    if(o->status) o->status[uid] = (uint8_t)status;
}

typedef struct { const step_ctx *c; const navhip_step_out *o; int begin, end; } step_job;

static void *step_thread(void *arg)
{
    step_job *j = arg;
    for(int uid = j->begin; uid < j->end; uid++)
        step_one(j->c, uid, j->o);
    return NULL;
}

/* The velocity step for entities [work_begin, work_end) (0,0 = all), fork-joined over
 * `nthreads` contiguous slabs like move_submit_cpu_work (movement.c:3746-3774). */
int no_agent_step(const no_map *m, const navhip_world *w, const navhip_step_out *o, int nthreads)
{
    if(!m || !w || !o || !o->vel_xz || w->n_ents < 0) return -1;
    if(w->hz != 20 && w->hz != 10 && w->hz != 5 && w->hz != 1) return -1;
    if(w->n_ents == 0) return 0;
    no_grid g;
    if(grid_build(&g, w)) return -1;
    step_ctx c = {m, w, &g, 0, 0};
    c.scaled_max_force = (0.75f / w->hz * 20.0);
    c.scaled_max_force_f = (float)c.scaled_max_force;
    int b = w->work_begin, e = w->work_end;
    if(b == 0 && e == 0) e = w->n_ents;
    if(nthreads < 1) nthreads = 1;
    if(nthreads > 256) nthreads = 256;
    if(nthreads == 1) {
        step_job j = {&c, o, b, e};
        step_thread(&j);
    }else{
        pthread_t th[256];
        step_job jobs[256];
        int per = (e - b + nthreads - 1) / nthreads;
        for(int t = 0; t < nthreads; t++) {
            int jb = b + t * per, je = jb + per;
            if(jb > e) jb = e;
            if(je > e) je = e;
            jobs[t] = (step_job){&c, o, jb, je};
            pthread_create(&th[t], NULL, step_thread, &jobs[t]);
        }
        for(int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    grid_free(&g);
    return 0;
}

/* individual terms, for stage-by-stage parity tests */
int no_agent_forces(const no_map *m, const navhip_world *w, int uid, const float vdes[2],
                    float out_arrive[2], float out_cohesion[2], float out_separation[2])
{
    no_grid g;
    if(grid_build(&g, w)) return -1;
    step_ctx c = {m, w, &g, 0, 0};
    c.scaled_max_force = (0.75f / w->hz * 20.0);
    c.scaled_max_force_f = (float)c.scaled_max_force;
    int flock = w->flock[uid];
    v2 target = flock >= 0 ? mkv(w->flock_target_xz[2 * flock], w->flock_target_xz[2 * flock + 1]) : wpos(w, uid);
    v2 a = arrive_force_point(&c, uid, target, mkv(vdes[0], vdes[1]), w->has_dest_los[uid] != 0);
    v2 co = flock >= 0 ? cohesion_force(&c, uid, flock) : mkv(0.0f, 0.0f);
    v2 s = separation_force(&c, uid);
    out_arrive[0] = a.x; out_arrive[1] = a.z;
    out_cohesion[0] = co.x; out_cohesion[1] = co.z;
    out_separation[0] = s.x; out_separation[1] = s.z;
    grid_free(&g);
    return 0;
}

/* ===========================================================================================
 * dynamic obstacles: N_BlockersIncref / N_BlockersDecref (nav.c:4663,4685) and the local-island
 * relabel that follows a blocker change (n_update_local_islands, nav.c:967)
 * =========================================================================================== */

/* C_LineCircleIntersection, collision.c:960 */
static bool line_circle(float ax, float az, float bx, float bz, v2 center, float radius)
{
    float cx = center.x, cz = center.z;
    float dx = bx - ax, dz = bz - az;
    float A = pow(dx, 2) + pow(dz, 2);
    float B = 2 * (dx * (ax - cx) + dz * (az - cz));
    float C = pow(ax - cx, 2) + pow(az - cz, 2) - pow(radius, 2);
    float det = pow(B, 2) - (4 * A * C);
    float t;
    if(det < 0.0f || A < (1.0f / 1024.0f)) {
        return false;
    }else if(det == 0.0f) {
        t = -B / (2 * A);
    }else{
        float t1 = (-B + sqrt(det)) / (2 * A);
        float t2 = (-B - sqrt(det)) / (2 * A);
        t = t1 < t2 ? t1 : t2;
    }
    if(t < 0.0f || t > 1.0f)
        return false;
    return true;
}

/* C_PointInsideRect2D, collision.c:756 */
static bool point_in_rect(v2 p, v2 a, v2 b, v2 c, v2 d)
{
    (void)c;
    v2 ap = vsub(p, a), ab = vsub(b, a), ad = vsub(d, a);
    float ap_ab = vdot(ap, ab), ap_ad = vdot(ap, ad);
    return (ap_ab >= 0.0f && ap_ab <= vdot(ab, ab)) && (ap_ad >= 0.0f && ap_ad <= vdot(ad, ad));
}

/* C_CircleRectIntersection, collision.c:997; rect = {x, z, width, height}, x decreasing rightwards */
static bool circle_rect(v2 center, float radius, float rx, float rz, float rw, float rh)
{
    v2 corners[4] = {mkv(rx - rw, rz), mkv(rx, rz), mkv(rx, rz + rh), mkv(rx - rw, rz + rh)};
    if(point_in_rect(center, corners[0], corners[1], corners[2], corners[3]))
        return true;
    for(int i = 0; i < 4; i++)
        if(vlen(vsub(corners[i], center)) <= radius) return true;
    for(int i = 0; i < 4; i++) {
        int j = (i + 1) % 4;
        if(line_circle(corners[i].x, corners[i].z, corners[j].x, corners[j].z, center, radius))
            return true;
    }
    return false;
}

/* M_Tile_AllUnderCircle, tile.c:687 (nav resolution: 4-wu tiles) */
static int tiles_under_circle(const no_map *m, float map_x, float map_z, v2 center, float radius,
                              tiledesc *out, int maxout)
{
    tiledesc tile;
    if(!tile_for_point(m, map_x, map_z, center, &tile))
        return 0;
    int tile_len = 4;
    int ntiles = ceil(radius / tile_len);
    int ret = 0;
    for(int dr = -ntiles; dr <= ntiles; dr++) {
    for(int dc = -ntiles; dc <= ntiles; dc++) {
        /* M_Tile_RelativeDesc, tile.c:391 */
        int abs_r = tile.chunk_r * RES + tile.tile_r + dr;
        int abs_c = tile.chunk_c * RES + tile.tile_c + dc;
        if(abs_r < 0 || abs_r >= m->h * RES || abs_c < 0 || abs_c >= m->w * RES)
            continue;
        tiledesc curr = {abs_r / RES, abs_c / RES, abs_r % RES, abs_c % RES};
        /* M_Tile_Bounds, tile.c:356 */
        float bx = map_x - curr.chunk_c * 256 - curr.tile_c * 4;
        float bz = map_z + curr.chunk_r * 256 + curr.tile_r * 4;
        if(!circle_rect(center, radius, bx, bz, 4, 4))
            continue;
        out[ret++] = curr;
        if(ret == maxout)
            return ret;
    }}
    return ret;
}

/* M_Tile_Contour, tile.c:759 */
static int tiles_contour(const no_map *m, int ntds, const tiledesc *tds, tiledesc *out, int maxout)
{
    if(ntds == 0) return 0;
    int minr = 1 << 30, minc = 1 << 30, maxr = -(1 << 30), maxc = -(1 << 30);
    for(int i = 0; i < ntds; i++) {
        int absr = tds[i].chunk_r * RES + tds[i].tile_r, absc = tds[i].chunk_c * RES + tds[i].tile_c;
        if(absr < minr) minr = absr;
        if(absc < minc) minc = absc;
        if(absr > maxr) maxr = absr;
        if(absc > maxc) maxc = absc;
    }
    int dr = maxr - minr + 1, dc = maxc - minc + 1;
    int width = dc + 2, height = dr + 2;
    uint8_t *marked = calloc((size_t)width * height, 1);
    for(int i = 0; i < ntds; i++) {
        int absr = tds[i].chunk_r * RES + tds[i].tile_r, absc = tds[i].chunk_c * RES + tds[i].tile_c;
        marked[(absr - minr + 1) * width + (absc - minc + 1)] = 1;
    }
    int ret = 0;
    for(int r = minr - 1; r <= maxr + 1; r++) {
    for(int c = minc - 1; c <= maxc + 1; c++) {
        if(r < 0 || r >= m->h * RES) continue;
        if(c < 0 || c >= m->w * RES) continue;
        int relr = r - minr + 1, relc = c - minc + 1;
        bool contour = false;
        if(marked[relr * width + relc]) continue;
        if(ret == maxout) goto out;
        if((relr > 0 && marked[(relr - 1) * width + relc])
        || (relr < dr && marked[(relr + 1) * width + relc])
        || (relc > 0 && marked[relr * width + (relc - 1)])
        || (relc < dc && marked[relr * width + (relc + 1)]))
            contour = true;
        if((relr > 0 && relc > 0 && marked[(relr - 1) * width + (relc - 1)])
        || (relr > 0 && relc < dc && marked[(relr - 1) * width + (relc + 1)])
        || (relr < dr && relc > 0 && marked[(relr + 1) * width + (relc - 1)])
        || (relr < dr && relc < dc && marked[(relr + 1) * width + (relc + 1)]))
            contour = true;
        if(contour) {
            tiledesc td = {r / RES, c / RES, r % RES, c % RES};
            out[ret++] = td;
        }
    }}
out:
    free(marked);
    return ret;
}

/* n_update_blockers, nav.c:1017 (dirty: the chunk changed between occupied / non-occupied) */
static void update_blockers(const no_map *m, int layer, int faction_id, const tiledesc *tds, int ntds,
                            int ref_delta, uint8_t *dirty)
{
    uint16_t *bl = (uint16_t*)m->blockers[layer];
    uint8_t *fa = (uint8_t*)m->factions[layer];
    if(!bl) return;
    for(int i = 0; i < ntds; i++) {
        int chunk = tds[i].chunk_r * m->w + tds[i].chunk_c;
        size_t cell = ((size_t)chunk << 12) + tds[i].tile_r * RES + tds[i].tile_c;
        int prev = bl[cell];
        bl[cell] = (uint16_t)(bl[cell] + ref_delta);
        if(fa)
            fa[((size_t)chunk * MAX_FACTIONS << 12) + ((size_t)faction_id << 12) + tds[i].tile_r * RES + tds[i].tile_c] += ref_delta;
        int val = bl[cell];
        if(!!val != !!prev && dirty)
            dirty[(size_t)layer * m->w * m->h + chunk] = 1;
    }
}

/* n_update_blockers_circle_{ground,water,air}, nav.c:1051-1133, for the layers base..base+3 */
static void update_blockers_circle(const no_map *m, int base, v2 xz, float range, int faction_id,
                                   float map_x, float map_z, int ref_delta, uint8_t *dirty)
{
    static __thread tiledesc tds[1024], o3[1024], o5[1024], o7[1024];
    int ntds = tiles_under_circle(m, map_x, map_z, xz, range, tds, 1024);
    update_blockers(m, base + 0, faction_id, tds, ntds, ref_delta, dirty);

    int n3 = tiles_contour(m, ntds, tds, o3, 1024);
    update_blockers(m, base + 1, faction_id, tds, ntds, ref_delta, dirty);
    update_blockers(m, base + 1, faction_id, o3, n3, ref_delta, dirty);

    int n5 = tiles_contour(m, n3, o3, o5, 1024);
    update_blockers(m, base + 2, faction_id, tds, ntds, ref_delta, dirty);
    update_blockers(m, base + 2, faction_id, o3, n3, ref_delta, dirty);
    update_blockers(m, base + 2, faction_id, o5, n5, ref_delta, dirty);

    int n7 = tiles_contour(m, n5, o5, o7, 1024);
    update_blockers(m, base + 3, faction_id, tds, ntds, ref_delta, dirty);
    update_blockers(m, base + 3, faction_id, o3, n3, ref_delta, dirty);
    update_blockers(m, base + 3, faction_id, o5, n5, ref_delta, dirty);
    update_blockers(m, base + 3, faction_id, o7, n7, ref_delta, dirty);
}

/* N_BlockersIncref / N_BlockersDecref for a list of circles, applied in order to the (mutable)
 * blockers / factions planes of `m`; layers without a blockers plane are skipped.
 * dirty: [12][h*w] bytes or NULL. */
int no_blockers_circles(const no_map *m, const navhip_circle *circles, int n, float map_x,
                        float map_z, uint8_t *dirty)
{
    for(int i = 0; i < n; i++) {
        const navhip_circle *c = &circles[i];
        v2 xz = mkv(c->x, c->z);
        if(c->flags & NAVHIP_ENTITY_FLAG_AIR) {
            update_blockers_circle(m, 8, xz, c->radius, c->faction_id, map_x, map_z, c->delta, dirty);
        }else{
            update_blockers_circle(m, 4, xz, c->radius, c->faction_id, map_x, map_z, c->delta, dirty);
            update_blockers_circle(m, 0, xz, c->radius, c->faction_id, map_x, map_z, c->delta, dirty);
        }
    }
    return 0;
}

/* n_update_local_islands (nav.c:967) + n_visit_island_local (nav.c:901) for every chunk of one
 * layer (only: per-chunk filter or NULL).  out: [h][w][64][64] u16. */
int no_local_islands(const no_map *m, int layer, uint16_t *out, const uint8_t *only)
{
    if(!m->cost[layer]) return -1;
    static __thread int queue[CELLS * 5];
    for(int chunk = 0; chunk < m->w * m->h; chunk++) {
        if(only && !only[chunk]) continue;
        const uint8_t *cost = m->cost[layer] + ((size_t)chunk << 12);
        const uint16_t *bl = m->blockers[layer] ? m->blockers[layer] + ((size_t)chunk << 12) : NULL;
        uint16_t *li = out + ((size_t)chunk << 12);
        memset(li, 0xff, CELLS * sizeof(uint16_t));
        int local_iid = 0;
        for(int r = 0; r < RES; r++) {
        for(int c = 0; c < RES; c++) {
            if(li[r * RES + c] != ISLAND_NONE) continue;
            if(cost[r * RES + c] == COST_IMPASS) continue;
            if(bl && bl[r * RES + c] > 0) continue;
            int id = ++local_iid;
            int head = 0, tail = 0;
            queue[tail++] = r * RES + c;
            while(head < tail) {
                int cur = queue[head++];
                int cr = cur >> 6, cc = cur & 63;
                if(cost[cur] == COST_IMPASS) continue;
                if(bl && bl[cur] > 0) continue;
                if(li[cur] != ISLAND_NONE) continue;
                li[cur] = (uint16_t)id;
                static const int d4[4][2] = {{0, -1}, {0, 1}, {-1, 0}, {1, 0}};
                for(int k = 0; k < 4; k++) {
                    int nr = cr + d4[k][0], nc = cc + d4[k][1];
                    if(nr < 0 || nr >= RES || nc < 0 || nc >= RES) continue;     /* other chunk */
                    if(li[nr * RES + nc] != ISLAND_NONE) continue;   /* (cheap pre-filter) */
                    if(tail < CELLS * 5) queue[tail++] = nr * RES + nc;
                }
            }
        }}
    }
    return 0;
}

/* ===========================================================================================
 * timing helpers for bench.py's cpu_baseline leg ("port" kind)
 * =========================================================================================== */
static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

typedef struct { const no_map *m; const navhip_field_req *reqs; int begin, end, reps; uint8_t *scratch; } fjob;

static void *field_thread(void *arg)
{
    fjob *j = arg;
    for(int r = 0; r < j->reps; r++)
        for(int i = j->begin; i < j->end; i++)
            no_field_update(j->m, &j->reqs[i], j->scratch, NULL);
    return NULL;
}

double no_field_bench(const no_map *m, const navhip_field_req *reqs, int n, int reps, int nthreads)
{
    if(nthreads < 1) nthreads = 1;
    if(nthreads > 256) nthreads = 256;
    pthread_t th[256];
    fjob jobs[256];
    uint8_t *scratch = malloc((size_t)nthreads * CELLS);
    int per = (n + nthreads - 1) / nthreads;
    double t0 = now_s();
    for(int t = 0; t < nthreads; t++) {
        int b = t * per, e = b + per;
        if(b > n) b = n;
        if(e > n) e = n;
        jobs[t] = (fjob){m, reqs, b, e, reps, scratch + (size_t)t * CELLS};
        pthread_create(&th[t], NULL, field_thread, &jobs[t]);
    }
    for(int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    double dt = now_s() - t0;
    free(scratch);
    return dt;
}

double no_agent_bench(const no_map *m, const navhip_world *w, const navhip_step_out *o, int nthreads)
{
    double t0 = now_s();
    no_agent_step(m, w, o, nthreads);
    return now_s() - t0;
}
