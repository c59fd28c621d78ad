// agent_internal.h -- device-side views used by the agent-step kernels.
#pragma once
#include "navhip_internal.h"

// bg_<name>_t geometry (bitmap_grid.h:959-990) + the cell-sorted element pool
struct nh_grid {
    int32_t origin_x, origin_y;      // BG_SCALE_F(xmin), BG_SCALE_F(ymin)
    int     grid_w, grid_h;          // ceil(span / 16 wu)
    int     n;
    const int32_t *cell_start;       // [ncells+1]
    const int32_t *sorted_id;        // [n] records[] of the packed pool
    const int32_t *sx, *sy;          // [n] xs[] / ys[] of the packed pool (fixed point x256)
    const float4  *rec;              // [n][2] packed entity records of the inserted entities (agent
                                     // step only): {pos.x, pos.z, radius, flags} | {vel.x, vel.z, state, -}
};

// structure-of-arrays sources of the entity records (all null: no records)
struct nh_pack_src {
    const float    *vel_xz, *radius;
    const uint32_t *flags;
    const uint8_t  *state;
};

struct nh_spatial_scratch {
    int32_t *ent_ix, *ent_iy, *ent_cell;     // [n]
    int32_t *cell_count, *cell_fill;         // [ncells]
    int32_t *cell_start;                     // [ncells+1]
    int32_t *sorted_id, *sx, *sy;            // [n]
    int32_t *block_sum;                      // [ceil(ncells/1024)] scan scratch
    int32_t *box;                            // [4] bounding box of the stepped slab (optional filter)
    float4  *rec;                            // [n][2] entity records
    nh_pack_src src;
};

struct nh_step_params {
    nh_map_view map;
    nh_grid     grid;
    float       map_x, map_z;
    int         n_ents, n_flocks, hz;
    int         n_members;           // upper bound of flock_offsets[n_flocks] (launch size)
    int         work_begin, work_end;
    const float    *pos_xz, *vel_xz, *radius, *max_speed, *speed;
    const uint32_t *flags;
    const uint8_t  *state, *has_dest_los;
    const int32_t  *flock;
    const float    *vdes_xz;
    const float    *flock_target_xz;
    const int32_t  *flock_offsets, *flock_members, *flock_field_slot;
    const uint8_t  *field_pool;
    const uint8_t  *form_ready;
    const float    *cell_pos_xz, *form_cohesion_xz, *form_align_xz, *form_drag_xz;
};

struct nh_step_outs {
    float   *vel_xz, *new_pos_xz, *vdes_xz, *vpref_xz;
    uint8_t *status;
};

void nh_launch_spatial_build(const nh_grid &G, const float *d_pos_xz, nh_spatial_scratch &S,
                             int slab_begin, int slab_end, hipStream_t s);
size_t nh_cohesion_scratch_bytes(int n_flocks, int n_members);
void nh_cohesion_scratch_reset(int32_t *scratch, int n_flocks, int n_members, hipStream_t s);
// returns true when the lane regrouping for the next tick is still to be launched
// (nh_launch_cohesion_regroup, after the caller's "cohesion done" event)
bool nh_launch_cohesion(const nh_step_params &P, int32_t *scratch, float *d_coh, int *parity, hipStream_t s);
void nh_launch_cohesion_regroup(const nh_step_params &P, int32_t *scratch, int *parity, hipStream_t s);
size_t nh_pre_rec_bytes();
void nh_launch_agent_pre(const nh_step_params &P, void *d_pre, const nh_step_outs &O, hipStream_t s);
void nh_launch_agent_step(const nh_step_params &P, float *d_coh, const void *d_pre, const nh_step_outs &O,
                          hipStream_t s);
void nh_launch_spatial_query(const nh_grid &G, const float *d_query, int nq, float range, int maxout,
                             int32_t *d_counts, uint32_t *d_ids, hipStream_t s);
void nh_launch_clearpath(int nq, const float *ent, const float *des_v, const float *dyn,
                         const int32_t *n_dyn, const float *stat, const int32_t *n_stat, float *out,
                         hipStream_t s);
