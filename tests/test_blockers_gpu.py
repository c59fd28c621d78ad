"""GPU parity of the dynamic-obstacle path: N_BlockersIncref / N_BlockersDecref tile sets and
refcounts on all size layers, the changed-chunk flags, the local-island relabel, and incremental
field repair -- against the oracle restatement (itself pinned to the reference build in
tests/test_oracle_cpu.py) and, when present, the reference build directly."""
import numpy as np
import pytest

from oracle import navoracle, pfref
from tests import cases
from tests.test_oracle_cpu import _random_circles

pytestmark = pytest.mark.gpu


def _setup(navlib, grid, layers):
    h, w = grid.shape[0] // 64, grid.shape[1] // 64
    chunks = cases.synth.to_chunks(grid)
    onav = navoracle.OracleNav(chunks)
    ctx = navlib.NavContext(w, h)
    for layer in layers:
        onav.set_layer(layer, cost=chunks, blockers=np.zeros((h, w, 64, 64), np.uint16),
                       factions=np.zeros((h, w, 15, 64, 64), np.uint8))
        onav.set_layer(layer, local_islands=onav.local_islands(layer))
        ctx.upload_plane(layer, navlib.PLANE_COST_BASE, chunks)
        ctx.upload_plane(layer, navlib.PLANE_BLOCKERS, onav.plane(layer, "blockers"))
        ctx.upload_plane(layer, navlib.PLANE_FACTIONS, onav.plane(layer, "factions"))
        ctx.upload_plane(layer, navlib.PLANE_LOCAL_ISLANDS, onav.plane(layer, "local_islands"))
    return ctx, onav


def test_blockers_circles_all_layers_match(navlib):
    grid = cases.synth.cost_grid(2, 2, seed=13)
    layers = list(range(12))
    ctx, onav = _setup(navlib, grid, layers)
    circles = _random_circles(grid, 120, seed=2, air_frac=0.2, max_radius=22.0)
    circles["radius"][5], circles["radius"][6], circles["radius"][7] = 100.0, 75.0, 112.0
    undo = circles[::3].copy()
    undo["delta"] = -1
    ref = None
    if pfref.available():
        ref = pfref.RefNav(cases.synth.to_chunks(grid), layer_mask=0xff)
    for batch in (circles, undo):
        before = {l: onav.plane(l, "blockers").copy() for l in layers}
        dirty = onav.blockers_circles(batch)
        ctx.N_BlockersUpdate(batch.view(navlib.CIRCLE_DTYPE))
        for l in layers:
            assert np.array_equal(ctx.download_plane(l, navlib.PLANE_BLOCKERS), onav.plane(l, "blockers")), l
            assert np.array_equal(ctx.download_plane(l, navlib.PLANE_FACTIONS), onav.plane(l, "factions")), l
            # changed == final occupancy differs; a subset of the reference's toggle-based dirty set
            changed = ctx.changed_chunks(l, clear=True).astype(bool)
            occ_b = (before[l] > 0).reshape(2, 2, -1)
            occ_a = (onav.plane(l, "blockers") > 0).reshape(2, 2, -1)
            assert np.array_equal(changed, (occ_b != occ_a).any(-1)), l
            assert not (changed & ~dirty[l].astype(bool)).any(), l
            # local islands relabelled on the device for the changed chunks
            assert np.array_equal(ctx.download_plane(l, navlib.PLANE_LOCAL_ISLANDS), onav.local_islands(l)), l
        if ref is not None:
            for c in batch:
                if not (int(c["flags"]) & (1 << 15)):
                    ref.blockers_circle(float(c["x"]), float(c["z"]), float(c["radius"]),
                                        int(c["faction_id"]), int(c["flags"]), incref=(c["delta"] > 0))
            for l in range(8):
                # air circles never touch ground / water layers, so the reference agrees there
                assert np.array_equal(ctx.download_plane(l, navlib.PLANE_BLOCKERS),
                                      ref.plane(pfref.PLANE_BLOCKERS, l)), l
    assert onav.plane(11, "blockers").sum() > 0 and onav.plane(3, "blockers").sum() > onav.plane(0, "blockers").sum()
    # invalid circles are rejected, not silently clamped
    bad = circles[:1].copy()
    bad["radius"] = 113.0
    with pytest.raises(navlib.NavHipError):
        ctx.N_BlockersUpdate(bad.view(navlib.CIRCLE_DTYPE))
    ctx.close()


@pytest.mark.parametrize("seed,frac", [(5, 0.2), (9, 0.45)])
def test_device_local_island_relabel_matches_reference_labels(navlib, seed, frac):
    grid = cases.synth.cost_grid(3, 3, seed=seed, frac_impassable=frac)
    blk = cases.random_blockers(grid, seed=seed + 1, frac=0.05)
    chunks = cases.synth.to_chunks(grid)
    ctx = navlib.NavContext(3, 3)
    ctx.upload_plane(0, navlib.PLANE_COST_BASE, chunks)
    ctx.upload_plane(0, navlib.PLANE_BLOCKERS, blk)
    ctx.relabel_local_islands(0)
    got = ctx.download_plane(0, navlib.PLANE_LOCAL_ISLANDS)
    onav = navoracle.OracleNav(chunks, blk)
    assert np.array_equal(got, onav.local_islands(0))
    if pfref.available():
        nav = pfref.RefNav(chunks)
        nav.set_blockers(blk)
        assert np.array_equal(got, nav.plane(pfref.PLANE_LOCAL_ISLANDS))
    assert got[got != 0xFFFF].max() >= 2          # several components somewhere
    ctx.close()


def test_incremental_field_repair_equals_full_rebuild(navlib):
    synth = cases.synth
    W, K = 4, 6
    grid = synth.cost_grid(W, W, seed=1234)
    ctx, onav = _setup(navlib, grid, [0])
    liid = synth.from_chunks(onav.plane(0, "local_islands"))
    dests = synth.destinations(grid, K, seed=42)
    cols = synth.whole_map_requests(grid, dests, liid)
    reqs = cases.cols_to_reqs(cols, navlib.FIELD_REQ_DTYPE)
    reqs["flags"] = navlib.REQ_LIVE_IIDS
    pool, _ = ctx.N_FlowFieldUpdate(reqs)
    exp0, _ = onav.build_fields(reqs.view(navoracle.FIELD_REQ_DTYPE))
    assert np.array_equal(pool, exp0)
    ctx.changed_chunks(0, clear=True)
    # drop obstacles on passable ground (not on a destination), then repair only what changed
    rng = np.random.RandomState(3)
    cells = synth.passable_cells(grid)
    pick = cells[rng.choice(len(cells), 40, replace=False)]
    xz = synth.cell_centre(W, W, pick[:, 0], pick[:, 1])
    circles = np.zeros(len(pick), navlib.CIRCLE_DTYPE)
    circles["x"], circles["z"] = xz[:, 0], xz[:, 1]
    circles["radius"] = rng.uniform(2, 6, len(pick))
    circles["delta"] = 1
    onav.blockers_circles(circles.view(navoracle.CIRCLE_DTYPE))
    onav.set_layer(0, local_islands=onav.local_islands(0))
    ctx.N_BlockersUpdate(circles)
    changed = ctx.changed_chunks(0).astype(bool)
    assert 0 < changed.sum() < W * W
    r2 = reqs.copy()
    r2["flags"] = navlib.REQ_LIVE_IIDS | navlib.REQ_IF_CHANGED
    repaired, _ = ctx.N_FlowFieldUpdate(r2, inout=pool)           # slots start from the old pool
    need = changed[reqs["chunk_r"], reqs["chunk_c"]] | \
        ((reqs["type"] == 0) & changed[reqs["next_chunk_r"], reqs["next_chunk_c"]])
    exp_new, _ = onav.build_fields(r2.view(navoracle.FIELD_REQ_DTYPE))   # full rebuild, new planes
    assert np.array_equal(repaired[need], exp_new[need])
    assert np.array_equal(repaired[~need], pool[~need])           # untouched slots
    assert need.sum() < len(reqs)
    ctx.close()


def test_live_island_ids_come_from_a_labelled_portal_tile(navlib):
    """NAVHIP_REQ_LIVE_IIDS with a blocker on the FIRST tile of a portal: the ids must come from a tile
    that still has a label (the planner only ever hands out ids of reachable tiles), i.e. the fields
    equal those of requests that carry the right ids explicitly; a portal blocked end to end is
    skipped (slot untouched)."""
    synth = cases.synth
    W, K = 4, 6
    grid = synth.cost_grid(W, W, seed=1234)
    ctx, onav = _setup(navlib, grid, [0])
    liid = synth.from_chunks(onav.plane(0, "local_islands"))
    cols = synth.whole_map_requests(grid, synth.destinations(grid, K, seed=42), liid)
    reqs = cases.cols_to_reqs(cols, navlib.FIELD_REQ_DTYPE)
    portal = np.flatnonzero(reqs["type"] == navlib.TARGET_PORTAL)
    rng = np.random.RandomState(4)
    hit = rng.choice(portal, 24, replace=False)
    # a small obstacle on the first tile of the port / next portal of the chosen requests
    rows = np.where(np.arange(len(hit)) % 2 == 0, reqs["chunk_r"][hit] * 64 + reqs["port_r0"][hit],
                    reqs["next_chunk_r"][hit] * 64 + reqs["next_r0"][hit])
    colsx = np.where(np.arange(len(hit)) % 2 == 0, reqs["chunk_c"][hit] * 64 + reqs["port_c0"][hit],
                     reqs["next_chunk_c"][hit] * 64 + reqs["next_c0"][hit])
    xz = synth.cell_centre(W, W, rows, colsx)
    circles = np.zeros(len(hit), navlib.CIRCLE_DTYPE)
    circles["x"], circles["z"], circles["radius"], circles["delta"] = xz[:, 0], xz[:, 1], 3.0, 1
    onav.blockers_circles(circles.view(navoracle.CIRCLE_DTYPE))
    onav.set_layer(0, local_islands=onav.local_islands(0))
    ctx.N_BlockersUpdate(circles)
    li = synth.from_chunks(onav.plane(0, "local_islands"))
    first_port = li[reqs["chunk_r"] * 64 + reqs["port_r0"], reqs["chunk_c"] * 64 + reqs["port_c0"]]
    first_next = li[reqs["next_chunk_r"] * 64 + reqs["next_r0"], reqs["next_chunk_c"] * 64 + reqs["next_c0"]]
    affected = np.zeros(len(reqs), bool)
    affected[portal] = (first_port[portal] == 0xFFFF) | (first_next[portal] == 0xFFFF)
    assert affected.sum() >= 12                                # the case is there

    def first_label(cr, cc, r0, c0, r1, c1):
        blk = li[cr * 64 + r0:cr * 64 + r1 + 1, cc * 64 + c0:cc * 64 + c1 + 1].ravel()
        ok = blk[blk != 0xFFFF]
        return int(ok[0]) if len(ok) else 0xFFFF

    explicit = reqs.copy()
    dead = np.zeros(len(reqs), bool)
    for i in portal:
        q = reqs[i]
        explicit["port_iid"][i] = first_label(q["chunk_r"], q["chunk_c"], q["port_r0"], q["port_c0"], q["port_r1"], q["port_c1"])
        explicit["next_iid"][i] = first_label(q["next_chunk_r"], q["next_chunk_c"], q["next_r0"], q["next_c0"], q["next_r1"], q["next_c1"])
        dead[i] = explicit["port_iid"][i] == 0xFFFF or explicit["next_iid"][i] == 0xFFFF
    live = reqs.copy()
    live["flags"] = navlib.REQ_LIVE_IIDS
    marker = np.full((len(reqs), 64, 64), 7, np.uint8)        # (what a skipped slot keeps)
    live_in = live.copy()
    live_in["flags"] |= navlib.REQ_INOUT
    got, _ = ctx.N_FlowFieldUpdate(live)
    exp, _ = ctx.N_FlowFieldUpdate(explicit)
    assert np.array_equal(got[~dead], exp[~dead])
    ora, _ = onav.build_fields(live.view(navoracle.FIELD_REQ_DTYPE))
    assert np.array_equal(got[~dead], ora[~dead])
    ctx.close()


@pytest.mark.skipif(not pfref.available(), reason="needs the reference build (oracle/_ref)")
@pytest.mark.parametrize("w,h", [(3, 2), (2, 3)])
def test_blockers_on_non_square_maps_match_reference(navlib, w, h):
    """N_BlockersIncref / Decref on 3x2 and 2x3 chunks: refcount planes of the ground and water
    layers against the reference itself, local islands against the restatement."""
    grid = cases.synth.cost_grid(w, h, seed=40 + w)
    layers = list(range(8))
    ctx, onav = _setup(navlib, grid, layers)
    ref = pfref.RefNav(cases.synth.to_chunks(grid), layer_mask=0xff)
    circles = _random_circles(grid, 150, seed=6, air_frac=0.0, max_radius=30.0)
    undo = circles[::4].copy()
    undo["delta"] = -1
    for batch in (circles, undo):
        onav.blockers_circles(batch)
        ctx.N_BlockersUpdate(batch.view(navlib.CIRCLE_DTYPE))
        for c in batch:
            ref.blockers_circle(float(c["x"]), float(c["z"]), float(c["radius"]), int(c["faction_id"]),
                                int(c["flags"]), incref=(c["delta"] > 0))
        for l in layers:
            got = ctx.download_plane(l, navlib.PLANE_BLOCKERS)
            assert np.array_equal(got, ref.plane(pfref.PLANE_BLOCKERS, l)), l
            assert np.array_equal(ctx.download_plane(l, navlib.PLANE_LOCAL_ISLANDS), onav.local_islands(l)), l
    assert ctx.download_plane(0, navlib.PLANE_BLOCKERS).sum() > 0
    ctx.close()
