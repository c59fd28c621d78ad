/* oracle/ref/ref_internal.h -- TEST INFRASTRUCTURE ONLY: declarations shared by
 * the harness translation units (each of which #includes one reference .c). */
#ifndef PFREF_INTERNAL_H
#define PFREF_INTERNAL_H

#include "pfref.h"
#include "navigation/nav_private.h"
#include "navigation/field.h"

struct pfref_nav{
    struct nav_private priv;
    vec3_t             map_pos;
    unsigned           layer_mask;
};

bool pfref_make_target(const struct nav_private *priv, const pfref_field_req *req,
                       struct field_target *out);
void pfref_req_from_target(struct coord chunk, int faction_id, enum nav_layer layer,
                           const struct field_target *t, pfref_field_req *out);
void pfref_dirs_to_ff(const uint8_t *dirs, struct flow_field *ff);
void pfref_ff_to_dirs(const struct flow_field *ff, uint8_t *dirs);

#endif
