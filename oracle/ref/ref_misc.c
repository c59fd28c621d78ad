/* oracle/ref/ref_misc.c -- TEST INFRASTRUCTURE ONLY.
 * Direct drivers for G_ClearPath_NewVelocity (clearpath.c:694) and the bitmap-grid
 * spatial index behind G_Pos_EntsInCircleFrom (position.c:379, bitmap_grid.h:1376),
 * both linked from the reference's own objects. */
#include "mem.h"
#define MEM_FILE_SYS MEM_SYS_GAME
#define MEM_FILE_SUB 0
#include "game/clearpath.h"
#include "game/position.h"
#include "pfref.h"

#include <stdlib.h>
#include <string.h>

static void fill_vec(vec_cp_ent_t *v, const float *src, int n)
{
    vec_cp_ent_init(v);
    vec_cp_ent_resize(v, n > 0 ? n : 1);
    for(int i = 0; i < n; i++) {
        struct cp_ent e = {
            .xz_pos = (vec2_t){src[i * 5 + 0], src[i * 5 + 1]},
            .xz_vel = (vec2_t){src[i * 5 + 2], src[i * 5 + 3]},
            .radius = src[i * 5 + 4]
        };
        vec_cp_ent_push(v, e);
    }
}

void pfref_clearpath_new_velocity(const float ent[5], const float des_v[2],
                                  const float *dyn, int n_dyn,
                                  const float *stat, int n_stat, float out[2])
{
    vec_cp_ent_t vd, vs;
    fill_vec(&vd, dyn, n_dyn);
    fill_vec(&vs, stat, n_stat);
    struct cp_ent e = {
        .xz_pos = (vec2_t){ent[0], ent[1]}, .xz_vel = (vec2_t){ent[2], ent[3]}, .radius = ent[4]
    };
    vec2_t r = G_ClearPath_NewVelocity(e, 0, (vec2_t){des_v[0], des_v[1]}, vd, vs, false);
    out[0] = r.x;
    out[1] = r.z;
    vec_cp_ent_destroy(&vd);
    vec_cp_ent_destroy(&vs);
}

static bool uid_eq(const uint32_t *a, const uint32_t *b) { return *a == *b; }

void pfref_spatial_query(float xmin, float xmax, float zmin, float zmax,
                         const float *pos_xz, int n,
                         const float *query_xz, int nq, float range, int maxout,
                         int32_t *out_counts, uint32_t *out_ids)
{
    bg_ent_t tree;
    bg_ent_init(&tree, xmin, xmax, zmin, zmax, uid_eq);
    bg_ent_reserve(&tree, n > 0 ? n : 1);
    for(int i = 0; i < n; i++)
        bg_ent_insert(&tree, pos_xz[2 * i], pos_xz[2 * i + 1], (uint32_t)i);
    bg_ent_cleanup(&tree);          /* G_Pos_CopyBitmapGrid packs before snapshotting */
    for(int q = 0; q < nq; q++) {
        /* G_Pos_EntsInCircleFrom = this call + the garrisoned-flag filter (position.c:379-386) */
        out_counts[q] = bg_ent_inrange_circle(&tree, query_xz[2 * q], query_xz[2 * q + 1], range,
            out_ids + (size_t)q * maxout, maxout);
    }
    bg_ent_destroy(&tree);
}
