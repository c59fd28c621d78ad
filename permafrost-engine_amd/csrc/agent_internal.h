// agent_internal.h -- launch interface of agent_kernels.hip (used by navhip_api.hip).
#pragma once
#define NH_SCAN_T 256      /* threads (= cells) per block of the two-pass scans */
#include "navhip_internal.h"
#include "agent_types.h"

struct nh_spatial_scratch {
    int32_t *ent_cell, *ent_rank;            // [n]
    int32_t *cell_count;                     // [ncells]   zero between builds
    int32_t *cell_start;                     // [ncells+1]
    int32_t *tmp_id;                         // [n] cell-sorted uids, arrival order inside a cell
    int32_t *block_sum;                      // [ceil(ncells / NH_SCAN_T)] scan scratch
    int32_t *box;                            // [2][4] bounding box of the stepped slab (optional filter);
    int      box_parity;                     //        the two alternate between builds (no memset)
    float4  *recA;                           // [n] pool records
    float2  *recV;                           // [n]
    int32_t *pool_of;                        // [n]
    nh_pack_src src;
};

// fills G.cell_start / recA / recV / pool_of from S
void nh_launch_spatial_build(nh_grid &G, const float *d_pos_xz, nh_spatial_scratch &S,
                             int slab_begin, int slab_end, hipStream_t s);
void nh_launch_agent_nbr(const nh_step_params &P, const nh_nbr &NB, hipStream_t s);
size_t nh_cohesion_scratch_bytes(int n_flocks, int n_members);
hipError_t nh_cohesion_scratch_reset(int32_t *scratch, int n_flocks, int n_members, hipStream_t s);
// returns true when the lane regrouping for the next tick is still to be launched
// (nh_launch_cohesion_regroup, after the caller's "cohesion done" event)
bool nh_launch_cohesion(const nh_step_params &P, int32_t *scratch, float *d_coh, int *parity, hipStream_t s);
void nh_launch_cohesion_regroup(const nh_step_params &P, int32_t *scratch, int *parity, hipStream_t s);
// k_agent_mid -> k_cp_rows -> k_agent_full | k_cp_small -> retry -> k_cp_heavy; WL.count holds 2 * NH_WL_COUNTERS counters.
// side / ev (or null): a second stream for the second chain, two events
int nh_worklist_cap(int n_work);
bool nh_launch_agent_finish(const nh_step_params &P, const nh_nbr &NB, float *d_coh, nh_mid_rec *d_mid,
                            nh_worklists WL, int parity, const nh_step_outs &O, hipStream_t s,
                            hipStream_t side, navhip_ctx *ctx);
void nh_launch_state_update(const nh_step_params &P, const navhip_state_in &in, float4 *d_arrived, int32_t *d_arrived_n, uint8_t *d_state, uint8_t *d_flags,
                            hipStream_t s);
void nh_launch_region_lookup(const nh_step_params &P, int nq, const float *d_pos, const int32_t *d_rows,
                             const int32_t *d_centre_abs, const int32_t *d_radius, uint8_t *d_dir, uint8_t *d_at_slot,
                             hipStream_t s);
void nh_launch_spatial_query(const nh_grid &G, const float *d_query, int nq, float range, int maxout,
                             int32_t *d_counts, uint32_t *d_ids, hipStream_t s);
// rows = 1: one row of 16 lanes per problem (n_dyn + n_stat <= 16); 2: a team of waves per problem (a
// workgroup); 0: one wave per problem
void nh_launch_clearpath(int nq, const float *ent, const float *des_v, const float *dyn,
                         const int32_t *n_dyn, const float *stat, const int32_t *n_stat, float *out,
                         int rows, hipStream_t s);
