/* bindings/permafrost/field_hip.c -- the field.c half of the reference-side binding of libnavhip.so.
 *
 * Appended to src/navigation/field.c's translation unit (it needs that file's static frontier
 * extractors).  The padded-region builders of the reference -- field_update_enemies (field.c:1537),
 * field_update_entity (:1615), field_update_zone (:1822): what every job of the asynchronous field batch
 * runs (N_RequestAsyncEnemySeekField / SurroundField / GroupArrivalField, nav.c:3776,3831,3921) -- split
 * into two halves:
 *   - the GAME-SIDE half stays here on the host, unchanged: which tiles are the cost-zero frontier
 *     (positions, selection radii, factions, diplomacy, fog of war of the entities around the chunk;
 *     the open-tile flood of a zone) -- field_enemies_initial_frontier (:1209),
 *     field_entity_initial_frontier (:1313), field_zone_initial_frontier (:1708);
 *   - the GRID half -- 128 x 128 integration (field_build_integration_region :582) and the bake of the
 *     64 x 64 window (field_build_flow_region :763) -- becomes one navhip_region_req.
 * N_HIP_RegionRequest does the first half and describes the second.  In this repository the file is
 * compiled inside the test harness (oracle/ref/ref_field.c #includes it right after field.c).
 */
#include <navhip.h>

/* The request + seeds of N_FlowFieldUpdate(chunk, priv, faction_id, layer, target, ctx, inout) for an
 * ENEMIES / ENTITY / ZONE target.  out_seeds: (abs_r, abs_c) int16 pairs, at most max_seeds of them.
 * Returns false when the device form does not cover the case (a map one chunk high or wide, whose
 * padded region is not square): the caller runs the CPU builder. */
bool N_HIP_RegionRequest(struct coord chunk_coord, const struct nav_private *priv, enum nav_layer layer,
                         struct field_target target, struct nav_unit_query_ctx *ctx,
                         navhip_region_req *out_req, int16_t *out_seeds, size_t max_seeds, size_t *out_nseeds)
{
    if(target.type != TARGET_ENEMIES && target.type != TARGET_ENTITY && target.type != TARGET_ZONE)
        return false;

    /* geometry: field.c:1557-1572 == :1630-1645 == :1834-1849 */
    const int rdim = (priv->height > 1) ? FIELD_RES_R * 2 + (FIELD_RES_R % 2) : FIELD_RES_R;
    const int cdim = (priv->width  > 1) ? FIELD_RES_C * 2 + (FIELD_RES_C % 2) : FIELD_RES_C;
    if(rdim != cdim)
        return false;
    struct tile_desc base = (struct tile_desc){
        .chunk_r = (chunk_coord.r > 0) ? chunk_coord.r - 1 : chunk_coord.r,
        .chunk_c = (chunk_coord.c > 0) ? chunk_coord.c - 1 : chunk_coord.c,
        .tile_r  = (chunk_coord.r > 0) ? FIELD_RES_R / 2 + (FIELD_RES_R % 2) : 0,
        .tile_c  = (chunk_coord.c > 0) ? FIELD_RES_C / 2 + (FIELD_RES_C % 2) : 0,
    };
    const int roff = (chunk_coord.r > 0) ? FIELD_RES_R / 2 + (FIELD_RES_R % 2) : 0;
    const int coff = (chunk_coord.c > 0) ? FIELD_RES_C / 2 + (FIELD_RES_C % 2) : 0;

    /* the game-side half: the reference's own frontier extraction */
    STALLOC(struct tile_desc, init_frontier, rdim * cdim);
    size_t ninit = 0;
    switch(target.type) {
    case TARGET_ENEMIES:
        ninit = field_enemies_initial_frontier(&target.enemies, priv, base, rdim, cdim, layer, ctx,
            init_frontier, rdim * cdim);
        break;
    case TARGET_ENTITY:
        ninit = field_entity_initial_frontier(&target.ent, priv, base, rdim, cdim, layer, ctx,
            init_frontier, rdim * cdim);
        break;
    default: {
        /* field.c:1851-1856: a constant tile budget, the ideal disc's area */
        size_t budget = (size_t)(M_PI * target.zone.radius * target.zone.radius + 0.5);
        if(budget > (size_t)(rdim * cdim))
            budget = rdim * cdim;
        ninit = field_zone_initial_frontier(&target.zone, priv, base, rdim, cdim, layer,
            init_frontier, budget);
        break;
    }
    }
    bool ok = ninit <= max_seeds;
    for(size_t i = 0; ok && i < ninit; i++) {
        out_seeds[2 * i + 0] = (int16_t)(init_frontier[i].chunk_r * FIELD_RES_R + init_frontier[i].tile_r);
        out_seeds[2 * i + 1] = (int16_t)(init_frontier[i].chunk_c * FIELD_RES_C + init_frontier[i].tile_c);
    }
    STFREE(init_frontier);
    if(!ok)
        return false;

    memset(out_req, 0, sizeof(*out_req));
    out_req->layer = layer;
    out_req->out_mode = 1;              /* the 64 x 64 window, in place: field_build_flow_region */
    out_req->enemies = 0;               /* all three builders integrate with enemies = 0 (:1599,:1668,:1880) */
    out_req->base_abs_r = (int16_t)(base.chunk_r * FIELD_RES_R + base.tile_r);
    out_req->base_abs_c = (int16_t)(base.chunk_c * FIELD_RES_C + base.tile_c);
    out_req->rdim = rdim; out_req->cdim = cdim;
    out_req->roff = roff; out_req->coff = coff;
    out_req->seed_count = (uint32_t)ninit;
    *out_nseeds = ninit;
    return true;
}
