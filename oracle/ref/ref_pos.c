/* oracle/ref/ref_pos.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Pulls the reference's src/game/position.c into this translation unit (by #include from
 * /root/reference/src; nothing is copied) so that the harness can fill its static position table and
 * spatial index directly -- G_Pos_Set (position.c:125) drags in fog, regions, combat refs -- and gives
 * the handful of game-side queries that the reference's enemy / entity frontier extraction makes
 * (field.c:1209-1370: G_Pos_EntsInRect is position.c's own; faction, flags, selection radius, diplomacy,
 * fog and death come from game.c / combat.c / fog_of_war.c, which are not part of the hot path) a small
 * explicit game state to answer from.
 */
#include "game/position.c"

#include "pfref.h"

static struct{
    int             n;
    float          *radius;
    int32_t        *faction;
    uint32_t       *flags;
    bool            loaded;
}s_game;

/* every entity: uid == index.  Bounds = the map's, as G_Pos_Init computes them (position.c:276-283). */
void pfref_game_load(float xmin, float xmax, float zmin, float zmax, int n, const float *pos_xz,
                     const float *radius, const int32_t *faction, const uint32_t *flags)
{
    pfref_game_unload();
    s_postable = kh_init(pos);
    bg_ent_init(&s_postree, xmin, xmax, zmin, zmax, uids_equal);
    bg_ent_reserve(&s_postree, n > 0 ? n : 1);
    for(int i = 0; i < n; i++) {
        int ret;
        khiter_t k = kh_put(pos, s_postable, (uint32_t)i, &ret);
        kh_val(s_postable, k) = (vec3_t){pos_xz[2 * i], 0.0f, pos_xz[2 * i + 1]};
        bg_ent_insert(&s_postree, pos_xz[2 * i], pos_xz[2 * i + 1], (uint32_t)i);
    }
    bg_ent_cleanup(&s_postree);                       /* on_update_start, position.c:259 */
    s_game.n = n;
    s_game.radius = malloc(sizeof(float) * (n > 0 ? n : 1));
    s_game.faction = malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    s_game.flags = malloc(sizeof(uint32_t) * (n > 0 ? n : 1));
    memcpy(s_game.radius, radius, sizeof(float) * n);
    memcpy(s_game.faction, faction, sizeof(int32_t) * n);
    memcpy(s_game.flags, flags, sizeof(uint32_t) * n);
    s_game.loaded = true;
}

void pfref_game_unload(void)
{
    if(!s_game.loaded)
        return;
    kh_destroy(pos, s_postable);
    bg_ent_destroy(&s_postree);
    free(s_game.radius); free(s_game.faction); free(s_game.flags);
    memset(&s_game, 0, sizeof(s_game));
}

bool     G_EntityExists(uint32_t uid)        { return s_game.loaded && uid < (uint32_t)s_game.n; }
int      G_GetFactionID(uint32_t uid)        { return s_game.faction[uid]; }
uint32_t G_FlagsGet(uint32_t uid)            { return s_game.flags[uid]; }
float    G_GetSelectionRadius(uint32_t uid)  { return s_game.radius[uid]; }
bool     G_Combat_IsDying(uint32_t uid)      { (void)uid; return false; }
uint16_t G_GetPlayerControlledFactions(void) { return 0xffff; }
bool     G_Fog_ObjVisible(uint16_t fac_mask, const struct obb *obb) { (void)fac_mask; (void)obb; return true; }
void     Entity_CurrentOBB(uint32_t uid, struct obb *out, bool identity) { (void)uid; (void)identity; memset(out, 0, sizeof(*out)); }

/* game.c's diplomacy table, from the same explicit enemy masks as G_GetEnemyFactions (ref_support.c) */
uint16_t G_GetEnemyFactions(int faction_id);
bool G_GetDiplomacyState(int fac_id_a, int fac_id_b, enum diplomacy_state *out)
{
    if(fac_id_a < 0 || fac_id_a >= MAX_FACTIONS || fac_id_b < 0 || fac_id_b >= MAX_FACTIONS || fac_id_a == fac_id_b)
        return false;
    *out = (G_GetEnemyFactions(fac_id_a) & (1u << fac_id_b)) ? DIPLOMACY_STATE_WAR : DIPLOMACY_STATE_PEACE;
    return true;
}
