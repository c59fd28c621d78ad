// map_view.h -- device-side view of the resident nav planes (plain pointers; shared by every kernel
// file and by the host-compiled unit-test build of the per-thread agent code, tests/hostsim).
#pragma once
#include <stdint.h>
#include "navhip.h"

struct nh_layer_view {
    const uint8_t  *cost;
    const uint16_t *blockers;
    const uint16_t *local_islands;
    const uint8_t  *factions;
    const uint64_t *passmask;
    const uint8_t  *unit_cost;
    const uint8_t  *changed;
    const uint16_t *islands;
    const uint64_t *probemask;     // [chunks][64][2] derived row bits: {cost_base != COST_IMPASSABLE, blockers > 0}
};
struct nh_map_view {
    int w, h;
    nh_layer_view layers[NAVHIP_NAV_LAYER_MAX];
};
